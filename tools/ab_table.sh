#!/bin/bash
# bench.py with the shipped tuning table against another one, interleaved on one box:
#   tools/ab_table.sh <other tuned.json> [rounds] [workloads...]
cd $GRAFT_REPO_ROOT
T=$1; R=${2:-3}; shift; shift; WL=${@:-lite183 full185}
cp elektronn2_amd/tuned.json /tmp/tuned_shipped.json
for i in $(seq $R); do for w in $WL; do for v in shipped other; do
  if [ $v == other ]; then cp $T elektronn2_amd/tuned.json; else cp /tmp/tuned_shipped.json elektronn2_amd/tuned.json; fi
  python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$w $v %.4f (dev %.4f)' % (d['ms_per_step'], d['roofline']['device_ms_per_step']))"
done; done; done
cp /tmp/tuned_shipped.json elektronn2_amd/tuned.json
