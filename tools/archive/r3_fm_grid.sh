#!/bin/bash
# first-layer kernels: sensitivity to the number of persistent work-groups per CU (debug build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export E2HIP_LIB=$GRAFT_REPO_ROOT/elektronn2_amd/csrc/build/dbg/libe2hip.so
for g in 1 2 3 4 6 8; do
  O=gpurun_out/r3s/g$g; mkdir -p $O
  E2_FM_GRID=$g timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload ${1:-full185} --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $O/bench.json 2> $O/err
  f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$g" <<'PY'
import csv, sys
rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1]))}
fw = [v for k, v in rows.items() if "firstm_fwd" in k]; bw = [v for k, v in rows.items() if "firstm_bwd" in k]
print("grid factor %s: fwd %.1f us  bwd %.1f us" % (sys.argv[2], fw[0] if fw else -1, bw[0] if bw else -1))
PY
  rm -rf $O/prof
done
