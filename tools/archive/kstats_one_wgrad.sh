#!/bin/bash
# kernel breakdown of one e2_conv3d_wgrad_bf16 configuration (GPU box):
#   tools/kstats_one_wgrad.sh <tag> cin cout kd kh kw D H W tile
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
T=${9}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$T -- python3 tools/one_wgrad.py "$@" > /dev/null 2>$O/p.err || { tail -5 $O/p.err; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$O/p_$T/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("$* |", r["Name"][:60].replace("(anonymous namespace)::", ""), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
rm -rf $O/p_$T
