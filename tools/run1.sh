cd $GRAFT_REPO_ROOT
export E2HIP_TUNE_CACHE=$PWD/gpurun_out/tuned_v7.json
rm -f $E2HIP_TUNE_CACHE
timeout -k 10 300 python bench.py --no-cpu-baseline --workload lite183 > gpurun_out/bench_lite_v7.json 2> gpurun_out/bench_lite_v7.err || { tail -5 gpurun_out/bench_lite_v7.err; exit 1; }
cat gpurun_out/bench_lite_v7.json
timeout -k 10 300 python bench.py --no-cpu-baseline --workload full185 > gpurun_out/bench_full_v7.json 2> gpurun_out/bench_full_v7.err || { tail -5 gpurun_out/bench_full_v7.err; exit 1; }
cat gpurun_out/bench_full_v7.json
cp $E2HIP_TUNE_CACHE gpurun_out/tuned_v7_bench.json
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
