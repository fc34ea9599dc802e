cd $GRAFT_REPO_ROOT
export E2HIP_TUNE_CACHE=$PWD/gpurun_out/tuned_v8.json
rm -f $E2HIP_TUNE_CACHE
timeout -k 10 300 python bench.py --no-cpu-baseline --workload lite183 > gpurun_out/bench_lite_v8.json 2> gpurun_out/bench_lite_v8.err || { tail -5 gpurun_out/bench_lite_v8.err; exit 1; }
cut -c1-330 gpurun_out/bench_lite_v8.json
timeout -k 10 300 python bench.py --no-cpu-baseline --workload full185 > gpurun_out/bench_full_v8.json 2> gpurun_out/bench_full_v8.err || { tail -5 gpurun_out/bench_full_v8.err; exit 1; }
cut -c1-330 gpurun_out/bench_full_v8.json
cp $E2HIP_TUNE_CACHE gpurun_out/tuned_v8_bench.json
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
