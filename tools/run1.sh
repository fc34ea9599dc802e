cd $GRAFT_REPO_ROOT
export E2HIP_TUNE_CACHE=$PWD/gpurun_out/tuned_v9.json
rm -f $E2HIP_TUNE_CACHE
for w in lite183 full185 unet_lite140; do
timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/bench_${w}_v9.json 2> gpurun_out/bench_${w}_v9.err || { tail -5 gpurun_out/bench_${w}_v9.err; exit 1; }
cut -c1-260 gpurun_out/bench_${w}_v9.json
done
cp $E2HIP_TUNE_CACHE gpurun_out/tuned_v9_bench.json
