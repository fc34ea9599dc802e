#!/bin/bash
# 1x1x1 / UpConv weight gradient as a GEMM with K-contiguous operands: op tests, sweep on the
# shapes of the nets, then re-tune those wgrad problems in the four training workloads
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3nt}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pointwise_wgrad or upconv" > $O/ops.log 2>&1 || { tail -40 $O/ops.log; exit 1; }
tail -2 $O/ops.log
tools/retune.sh ${1:-r3nt} '^wgrad\|([0-9]+,[0-9]+,1,1,1,|5,)' lite183 full185 unet_lite140 unet132
