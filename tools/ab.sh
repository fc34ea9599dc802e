#!/bin/bash
# A/B of an environment switch on ONE box, interleaved: tools/ab.sh <workload> "<envA>" "<envB>" [rounds]
cd $GRAFT_REPO_ROOT
W=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    ms=$(env $v python bench.py --workload $W --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f %.4f' % (d['ms_per_step'], d['roofline']['device_ms_per_step']))")
    echo "$W [$v] $ms"
  done
done
