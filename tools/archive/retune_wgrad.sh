#!/bin/bash
# re-tune the weight-gradient tilings only (isolated ranking + whole-step refinement)
set -o pipefail
T=${1:-wg}; shift
WL=${@:-lite183 full185}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/$T
python - <<'PY'
import json, re
p = "elektronn2_amd/tuned.json"
d = json.load(open(p))
keep = {k: v for k, v in d.items() if not re.search(r"^wgrad\|", k)}
print("dropping %d of %d shipped entries" % (len(d) - len(keep), len(d)))
json.dump(keep, open(p, "w"), indent=0, sort_keys=True)
PY
export E2HIP_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/$T/tune.json
for w in $WL; do
  timeout -k 10 900 python tools/tune_insitu.py $w ${E2_INSITU_TOPK:-5} ${E2_INSITU_REPLAYS:-40} > gpurun_out/$T/insitu_$w.log 2>&1 || { tail -20 gpurun_out/$T/insitu_$w.log; exit 1; }
  grep -v amdgpu gpurun_out/$T/insitu_$w.log | tail -12
done
for w in $WL; do
  timeout -k 10 300 python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/$T/bench_$w.json 2> gpurun_out/$T/bench_$w.err || { tail -20 gpurun_out/$T/bench_$w.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/$T/bench_$w.json')); print('$w', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done
