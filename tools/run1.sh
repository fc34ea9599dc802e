cd $GRAFT_REPO_ROOT
export E2HIP_TUNE_CACHE=$PWD/gpurun_out/tuned_v12.json
rm -f $E2HIP_TUNE_CACHE
for w in lite183 full185 unet_lite140; do
timeout -k 10 400 python bench.py --no-cpu-baseline --workload $w > gpurun_out/bench_${w}_v12.json 2> gpurun_out/bench_${w}_v12.err || { tail -5 gpurun_out/bench_${w}_v12.err; exit 1; }
cut -c1-200 gpurun_out/bench_${w}_v12.json
done
cp $E2HIP_TUNE_CACHE gpurun_out/tuned_v12_bench.json
