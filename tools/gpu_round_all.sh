#!/bin/bash
# profile collection for the record, all four training workloads of the driver's line in ONE
# gpurun call:  tools/gpu_round_all.sh <tag>   ->  gpurun_out/<tag>_<workload>/ (tools/gpu_round.sh)
set -o pipefail
T=${1:-round}
cd $GRAFT_REPO_ROOT
for W in lite183 full185 unet_lite140 unet132; do
  echo "== $W"
  tools/gpu_round.sh $W ${T}_$W > gpurun_out/${T}_$W.log 2>&1 || { tail -30 gpurun_out/${T}_$W.log; exit 1; }
  tail -4 gpurun_out/${T}_$W.log
done
