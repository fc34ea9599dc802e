#!/bin/bash
cd /root/repo
timeout -k 10 1100 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_mfp_gpu.py -q -x > gpurun_out/pool.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/pool.log
