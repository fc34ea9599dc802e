#!/bin/bash
# GPU test suite (run on the GPU box via gpurun): tools/gpu_tests.sh <tag> [pytest args]
set -o pipefail
T=${1:-tests}; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$T
if [ $# -eq 0 ]; then set -- tests -x; fi
timeout -k 10 1100 python -m pytest -m gpu -q --durations=15 "$@" > gpurun_out/$T/pytest_gpu.log 2>&1
rc=$?
tail -60 gpurun_out/$T/pytest_gpu.log
exit $rc
