#!/bin/bash
# per-kernel durations of one bench workload: tools/kstats.sh <workload> <tag>  (GPU box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=$1; O=gpurun_out/$2; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$W.json 2> $O/prof_$W.err || { tail -5 $O/prof_$W.err; exit 1; }
cp $(find $O/prof_$W -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$W.csv && rm -rf $O/prof_$W
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/kernel_stats_$W.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$W", json.load(open("$O/bench_$W.json"))["ms_per_step"], "ms (under rocprof); kernels/step us:", round(tot / 25e3, 1))
for r in rows:
    if any(k in r["Name"] for k in ("pool", "act_bwd", "head", "first", "adam", "pack", "fill", "copy")):
        print("   %-70s %5.1f/step %7.1f us" % (r["Name"][:70].replace("(anonymous namespace)::", ""), int(r["Calls"]) / 25, float(r["AverageNs"]) / 1e3))
PY
