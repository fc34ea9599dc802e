#!/bin/bash
cd /root/repo
for wl in dense183 dense183mfp; do
timeout -k 10 900 python bench.py --workload $wl --steps 3 --warmup 2 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; echo "rc=$?"
tail -c 900 gpurun_out/bench_$wl.json; tail -3 gpurun_out/bench_$wl.err
done
