#!/bin/bash
# bench.py with the shipped tuning table against another one, interleaved on one box:
#   tools/ab_table.sh <other tuned.json> [rounds] [workloads...]
# The shipped elektronn2_amd/tuned.json is NEVER touched: autotune._load layers
# $E2HIP_TUNE_CACHE over it, so the variant rides in a scratch copy of that cache.
set -o pipefail
cd $GRAFT_REPO_ROOT
T=$1; R=${2:-3}; shift; shift; WL=${@:-lite183 full185}
D=$(mktemp -d); trap 'rm -rf "$D"' EXIT INT TERM
echo '{}' > $D/shipped.json
cp "$T" $D/other.json
for i in $(seq $R); do for w in $WL; do for v in shipped other; do
  cp $D/$v.json $D/cache_$v.json            # (a run may add freshly tuned keys: to the copy)
  E2HIP_TUNE_CACHE=$D/cache_$v.json python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$w $v %.4f (dev %.4f)' % (d['ms_per_step'], d['roofline']['device_ms_per_step']))"
done; done; done
