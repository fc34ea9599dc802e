#!/bin/bash
cd /root/repo
timeout -k 10 800 python -m pytest tests/test_bf16_gpu.py -q -x > gpurun_out/bf16_test.log 2>&1; echo "rc=$?"
tail -3 gpurun_out/bf16_test.log
E2HIP_TUNE_CACHE=/tmp/tune_bf16.json timeout -k 10 900 python bench.py --workload unet_lite140 --mfma bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_bf16_unet.json 2> gpurun_out/bench_bf16_unet.err; echo "bench rc=$?"
tail -c 150 gpurun_out/bench_bf16_unet.json; tail -3 gpurun_out/bench_bf16_unet.err
cp /tmp/tune_bf16.json gpurun_out/tuned_bf16_v5.json
