#!/bin/bash
cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/all8.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/all8.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_f32_now.json 2>gpurun_out/bench_f32_now.err; echo "bench rc=$?"; tail -c 700 gpurun_out/bench_f32_now.json
timeout -k 10 300 python bench.py --no-cpu-baseline --mfma bf16 > gpurun_out/bench_bf16_now.json 2>gpurun_out/bench_bf16_now.err; echo "bench rc=$?"; tail -c 900 gpurun_out/bench_bf16_now.json
