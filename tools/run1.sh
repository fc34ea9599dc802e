#!/bin/bash
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_malis_nll_gpu.py -x -q > gpurun_out/malis_nll.log 2>&1; echo "rc=$?"
tail -40 gpurun_out/malis_nll.log
