#!/bin/bash
# full GPU verification + profile collection (run on the GPU box via gpurun)
set -o pipefail
mkdir -p gpurun_out/r
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/r/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r/pytest_gpu.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/r/smoke.log 2>&1 || { tail -20 gpurun_out/r/smoke.log; exit 1; }
tail -2 gpurun_out/r/smoke.log
timeout -k 10 200 python bench.py > gpurun_out/r/bench.json 2> gpurun_out/r/bench.err || { tail -20 gpurun_out/r/bench.err; exit 1; }
cat gpurun_out/r/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r/bench_prof.json 2> gpurun_out/r/prof.err || { tail -20 gpurun_out/r/prof.err; exit 1; }
cat gpurun_out/r/bench_prof.json
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/r/pmc_fetch.json 2> gpurun_out/r/pmc_fetch.err || { tail -20 gpurun_out/r/pmc_fetch.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/r/pmc_write.json 2> gpurun_out/r/pmc_write.err || { tail -20 gpurun_out/r/pmc_write.err; exit 1; }
find gpurun_out/r -name '*.csv' | xargs ls -la
