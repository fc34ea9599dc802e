#!/bin/bash
# kernel statistics of an arbitrary python command: tools/prof_cmd.sh <tag> <script> [args...]  (GPU box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=$1; shift
mkdir -p gpurun_out/$T
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof -- python3 "$@" > gpurun_out/$T/prof.log 2>&1 || { tail -5 gpurun_out/$T/prof.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$T/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-92s %6s %9.1f us" % (r["Name"][:92], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/$T/prof
