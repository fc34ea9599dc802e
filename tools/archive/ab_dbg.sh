#!/bin/bash
# like tools/ab.sh but with a debug-environment build of the library (make DEBUG_ENV=1 on the box)
# the debug-switch build lives in a directory of its own (the product library is never overwritten)
cd $GRAFT_REPO_ROOT/elektronn2_amd/csrc && make -j16 BUILD=build/dbg OUT=build/dbg/libe2hip.so DEBUG_ENV=1 > /dev/null 2>&1 || { echo build failed; exit 1; }
export E2HIP_LIB=$GRAFT_REPO_ROOT/elektronn2_amd/csrc/build/dbg/libe2hip.so
cd $GRAFT_REPO_ROOT
W=$1; shift
R=2
for i in $(seq $R); do
  for v in "$@"; do
    ms=$(env $v python bench.py --workload $W --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['roofline']['device_ms_per_step'])")
    echo "$W [$v] $ms"
  done
done
