#!/bin/bash
# A/B of shipped tilings inside the captured step, on ONE box (run via gpurun):
#   tools/ab_tuned.sh <tag> '<key>=<old tiling>' ... -- <workloads...>
# benches every workload with tuned.json as shipped ("new"), then with the listed keys set
# to the given tilings ("old"), twice each, interleaved.  The shipped table is never modified
# (ADVICE r4): the substitutions ride in $E2HIP_TUNE_CACHE, which autotune._load layers over it.
set -o pipefail
T=$1; shift
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$T; mkdir -p $O
SUBS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do SUBS+=("$1"); shift; done
shift
echo '{}' > $O/tuned_new.json
python - "$O" "${SUBS[@]}" <<'PY'
import json, sys
o = sys.argv[1]
d = json.load(open("elektronn2_amd/tuned.json"))
sub = {}
for s in sys.argv[2:]:
    k, v = s.split("=")
    assert k in d, k
    sub[k] = v
json.dump(sub, open(o + "/tuned_old.json", "w"), indent=0, sort_keys=True)
PY
for R in 1 2; do
  for V in new old; do
    for W in "$@"; do
      cp $O/tuned_$V.json $O/cache_$V.json
      E2HIP_TUNE_CACHE=$O/cache_$V.json timeout -k 10 300 python bench.py --workload $W --steps 40 --warmup 8 --no-cpu-baseline --no-also > $O/bench_${V}_${W}_$R.json 2> $O/bench_${V}_${W}_$R.err || { tail -20 $O/bench_${V}_${W}_$R.err; exit 1; }
      python -c "import json; d=json.load(open('$O/bench_${V}_${W}_$R.json')); print('$V $W %.4f ms (dev %.4f) frac %.4f' % (d['ms_per_step'], d['roofline']['device_ms_per_step'], d['roofline']['frac']))"
    done
  done
done
