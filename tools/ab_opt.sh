#!/bin/bash
# A/B of ONE plan option through its environment default, interleaved on one box:
#   tools/ab_opt.sh <tag> <ENV_VAR> <rounds> <workloads...>      e.g.  r5f E2_ADAM_PACK 3 lite183 full185
# prints ms / step (wall, device) of bench.py per (round, value, workload)
set -o pipefail
T=$1; V=$2; R=$3; shift; shift; shift
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$T; mkdir -p $O
for i in $(seq $R); do for W in "$@"; do for X in 0 1; do
  env $V=$X timeout -k 10 300 python bench.py --workload $W --steps 40 --warmup 8 --no-cpu-baseline --no-also $E2_AB_ARGS > $O/ab_${W}_${X}_$i.json 2> $O/ab_${W}_${X}_$i.err || { tail -20 $O/ab_${W}_${X}_$i.err; exit 1; }
  python -c "import json; d=json.load(open('$O/ab_${W}_${X}_$i.json')); print('$W $V=$X  %.4f ms (dev %.4f) frac %.4f loss %.6f' % (d['ms_per_step'], d['roofline']['device_ms_per_step'], d['roofline']['frac'], d['config']['final_loss']))"
done; done; done
