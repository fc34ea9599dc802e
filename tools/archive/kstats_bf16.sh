#!/bin/bash
# kernel statistics of a bench workload in bf16 mode (GPU box): tools/kstats_bf16.sh <workload> <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=$1; O=gpurun_out/$2; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python3 bench.py --workload $W --mfma bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_$W.json 2> $O/prof_$W.err || { tail -5 $O/prof_$W.err; exit 1; }
cp $(find $O/prof_$W -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bf16_$W.csv && rm -rf $O/prof_$W
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/kernel_stats_bf16_$W.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$W bf16", json.load(open("$O/bench_bf16_$W.json"))["ms_per_step"], "ms; kernels/step us:", round(tot / 25e3, 1))
for r in rows[:26]:
    print("   %-74s %5.1f/step %7.1f us %5.1f%%" % (r["Name"][:74].replace("(anonymous namespace)::", ""), int(r["Calls"]) / 25, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
