#!/bin/bash
# bf16 step with / without the operands made ahead (E2_BF16_AHEAD=1 / 0), on ONE box:
#   tools/ab_bf16_ahead.sh <tag> [workloads...]
set -o pipefail
T=${1:-ab_bf16}; shift
WL=${@:-lite183 full185}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$T; mkdir -p $O
for R in 1 2; do
  for V in 1 0; do
    for W in $WL; do
      E2_BF16_AHEAD=$V timeout -k 10 400 python bench.py --workload $W --mfma bf16 --steps 40 --warmup 8 --no-cpu-baseline --no-also > $O/bench_${V}_${W}_$R.json 2> $O/bench_${V}_${W}_$R.err || { tail -20 $O/bench_${V}_${W}_$R.err; exit 1; }
      python -c "import json; d=json.load(open('$O/bench_${V}_${W}_$R.json')); print('ahead=$V $W %.4f ms (dev %.4f) loss %.5f' % (d['ms_per_step'], d['roofline']['device_ms_per_step'], d['config']['final_loss']))"
    done
  done
done
