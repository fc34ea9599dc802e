#!/bin/bash
# round 3, first GPU call: op parity with the dz-clipped data gradient, per-layer GEMM tables,
# bench of both neuro3d nets
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3a}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q > $O/ops.log 2>&1 || { tail -30 $O/ops.log; exit 1; }
tail -3 $O/ops.log
for w in lite183 full185; do
  timeout -k 10 300 python tools/layer_table.py $w 20 > $O/layers_$w.md 2> $O/layers_$w.err || { tail -20 $O/layers_$w.err; exit 1; }
  tail -1 $O/layers_$w.md
  timeout -k 10 300 python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || { tail -20 $O/bench_$w.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done
