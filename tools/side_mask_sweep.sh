#!/bin/bash
# which weight gradients gain from the side stream in f32 mode?  tools/side_mask_sweep.sh <tag> <workload> <masks...>
set -o pipefail
T=$1; W=$2; shift; shift
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/$T
for i in 1 2; do for M in "$@"; do
  E2_SIDE_MASK=$M python bench.py --workload $W --steps 40 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W mask=$M %.4f ms' % d['ms_per_step'])"
done; done | tee gpurun_out/$T/sweep_$W.txt
