#!/bin/bash
# Re-tune the f32 conv fwd/dgrad tilings of the bench workloads with the current candidate
# lists (run on the GPU box): drops the shipped "igemm|" entries in the box's copy, lets the
# autotuner time every candidate, writes the new choices to gpurun_out/<tag>/tune.json.
#   tools/retune.sh <tag> [kinds-regex, default '^igemm\|'] [workloads...]
set -o pipefail
T=${1:-retune}; KR=${2:-'^igemm\|'}; shift; shift
WL=${@:-lite183 full185 unet_lite140 unet132}
# E2_RETUNE_FLAGS: extra bench.py flags (e.g. "--mfma bf16" to re-tune the *_bf16 entries)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$T
python - "$KR" <<'PY'
import json, re, sys
p = "elektronn2_amd/tuned.json"
d = json.load(open(p))
keep = {k: v for k, v in d.items() if not re.search(sys.argv[1], k)}
print("dropping %d of %d shipped entries" % (len(d) - len(keep), len(d)))
json.dump(keep, open(p, "w"), indent=0, sort_keys=True)
PY
export E2HIP_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/$T/tune.json
for w in $WL; do
  timeout -k 10 600 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline $E2_RETUNE_FLAGS > gpurun_out/$T/bench_$w.json 2> gpurun_out/$T/bench_$w.err || { tail -20 gpurun_out/$T/bench_$w.err; exit 1; }
  python -c "import json,sys; d=json.load(open('gpurun_out/$T/bench_$w.json')); print('$w', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done
