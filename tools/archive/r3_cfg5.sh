#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3e}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_unet_config5_gpu.py tests/test_native_size_gpu.py tests/test_bench_launcher.py tests/test_model_gpu.py tests/test_mfp_gpu.py -x -q -m gpu -s > $O/cfg5.log 2>&1; rc=$?
grep -v "^\[e2\]\|amdgpu.ids" $O/cfg5.log | tail -60
exit $rc
