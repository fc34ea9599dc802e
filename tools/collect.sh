#!/bin/bash
# every number DESIGN.md / README.md quote, in one go (GPU box): tools/collect.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-collect}
mkdir -p $O
run() { # name, args...
  n=$1; shift
  timeout -k 10 600 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n FAILED"; tail -5 $O/$n.err; return; }
  python - <<PY
import json
d = json.load(open("$O/$n.json"))
r = d.get("roofline", {})
print("%-22s %10.4f ms  value %.4g %s  frac %.4f" % ("$n", d["ms_per_step"], d["value"], d["unit"], r.get("frac") or 0))
PY
}
run lite183 --workload lite183
run full185 --workload full185 --no-cpu-baseline
run unet_lite140 --workload unet_lite140
run unet132 --workload unet132
run lite183_bf16 --workload lite183 --mfma bf16 --no-cpu-baseline
run full185_bf16 --workload full185 --mfma bf16 --no-cpu-baseline
run unet_lite140_bf16 --workload unet_lite140 --mfma bf16
run unet132_bf16 --workload unet132 --mfma bf16
run dense183 --workload dense183
run dense183mfp --workload dense183mfp
run dense512unet --workload dense512unet --steps 1
run dense512unet_bf16 --workload dense512unet --steps 1 --mfma bf16
run warp183 --workload warp183
timeout -k 10 300 python tools/soak.py 2000 > $O/soak_sync.txt 2>&1; tail -2 $O/soak_sync.txt
timeout -k 10 300 python tools/soak.py 2000 --async > $O/soak_async.txt 2>&1; tail -2 $O/soak_async.txt
timeout -k 10 300 python tools/dense512.py --net lite > $O/dense512_lite.txt 2>&1; tail -1 $O/dense512_lite.txt
