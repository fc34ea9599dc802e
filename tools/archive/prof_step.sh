#!/bin/bash
# rocprofv3 kernel trace of a short bench run; prints the last step's timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/ps
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ps -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ps_bench.json 2> gpurun_out/ps.err || { tail -20 gpurun_out/ps.err; exit 1; }
python3 tools/archive/timeline.py gpurun_out/ps
