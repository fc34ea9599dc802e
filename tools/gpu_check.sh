#!/bin/bash
# The standard GPU check of a change (run on the GPU box via gpurun):
#   tools/gpu_check.sh <tag> [notests] [workloads...]      default workloads: lite183 full185
# 1. pytest -m gpu (unless `notests`)  2. the driver's bench line (python bench.py)
# 3. per workload: rocprofv3 kernel stats of `bench.py --workload W --steps 20 --warmup 5`
#    -> gpurun_out/<tag>/kernel_stats_<W>.csv, diffed against the profile of record
#    (tools/kstats_diff.py; a regression is REPORTED here, adopt_profile.py refuses it)
set -o pipefail
T=${1:-check}; shift
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$T; mkdir -p $O
if [ "$1" == "notests" ]; then shift; else
  timeout -k 10 900 python -m pytest -m gpu -q -x --durations=10 tests > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
  tail -3 $O/pytest_gpu.log
fi
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/bench.json"))
r = d["roofline"]
print("bench: lite183 %.4f ms (dev %.4f) frac %.4f" % (d["ms_per_step"], r["device_ms_per_step"], r["frac"]))
a = r.get("also", {}).get("full185")
if a: print("       full185 %.4f ms (dev %.4f) frac %.4f" % (a["ms_per_step"], a["device_ms_per_step"], a["frac"]))
PY
if [ $# -eq 0 ]; then set -- lite183 full185; fi
for W in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_prof_$W.json 2> $O/prof_$W.err || { tail -5 $O/prof_$W.err; exit 1; }
  cp $(find $O/prof_$W -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$W.csv && rm -rf $O/prof_$W
  python tools/kstats_diff.py $O/kernel_stats_$W.csv --workload $W > $O/kstats_diff_$W.txt; echo "kstats_diff $W rc=$?"; tail -25 $O/kstats_diff_$W.txt
done
