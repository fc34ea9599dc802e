#!/bin/bash
# pointwise GEMM with longer pipeline chunks and the UpConv scatter epilogue: op tests, then
# re-tune the 1x1x1 / UpConv problems of the four training workloads
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3pw}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pointwise or upconv" > $O/ops.log 2>&1 || { tail -30 $O/ops.log; exit 1; }
tail -2 $O/ops.log
tools/retune.sh ${1:-r3pw} '^igemm\|([0-2],[0-9]+,[0-9]+,1,1,1,|[34],)' lite183 full185 unet_lite140 unet132
