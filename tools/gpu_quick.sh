#!/bin/bash
# A quick GPU iteration (run on the GPU box via gpurun):
#   tools/gpu_quick.sh <tag> "<pytest -k expression or empty>" [bench workloads...]
# runs the selected GPU tests, then `bench.py --workload W --no-cpu-baseline --no-also` per workload
set -o pipefail
T=${1:-quick}; K=$2; shift; shift
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$T; mkdir -p $O
if [ -n "$K" ]; then
  timeout -k 10 900 python -m pytest -m gpu -q -x tests -k "$K" > $O/pytest.log 2>&1 || { tail -50 $O/pytest.log; exit 1; }
  tail -3 $O/pytest.log
fi
for W in "$@"; do
  timeout -k 10 300 python bench.py --workload $W --steps 40 --warmup 8 --no-cpu-baseline --no-also > $O/bench_$W.json 2> $O/bench_$W.err || { tail -20 $O/bench_$W.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_$W.json')); print('$W %.4f ms (dev %.4f) frac %.4f' % (d['ms_per_step'], d['roofline']['device_ms_per_step'], d['roofline']['frac']))"
done
