#!/bin/bash
# per-kernel durations of one bf16 bench workload: tools/kstats_bf16.sh <workload> <tag> [E2_BF16_AHEAD]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=$1; O=gpurun_out/$2; A=${3:-1}; mkdir -p $O
export E2_BF16_AHEAD=$A
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python3 bench.py --workload $W --mfma bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_${W}_a$A.json 2> $O/prof_$W.err || { tail -5 $O/prof_$W.err; exit 1; }
cp $(find $O/prof_$W -name '*kernel_stats.csv' | head -1) $O/kernel_stats_${W}_bf16_a$A.csv && rm -rf $O/prof_$W
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats_${W}_bf16_a$A.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$W bf16 ahead=$A kernels/step us:", round(tot / 25e3, 1))
for r in rows:
    print("   %-80s %5.1f/step %7.1f us  %6.1f us/step" % (r["Name"][:80].replace("(anonymous namespace)::", ""), int(r["Calls"]) / 25, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 25e3))
PY
