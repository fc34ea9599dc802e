#!/bin/bash
# per-kernel profile of one bench workload (run on the GPU box via gpurun)
#   tools/gpu_prof.sh <workload> <tag>   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv}
set -o pipefail
W=${1:-lite183}; T=${2:-prof}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$T
timeout -k 10 200 python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err || { tail -20 gpurun_out/$T/bench.err; exit 1; }
cat gpurun_out/$T/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$T/bench_prof.json 2> gpurun_out/$T/prof.err || { tail -20 gpurun_out/$T/prof.err; exit 1; }
cp $(find gpurun_out/$T/prof -name '*kernel_stats.csv' | head -1) gpurun_out/$T/kernel_stats.csv
rm -rf gpurun_out/$T/prof
