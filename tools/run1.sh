#!/bin/bash
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_dp_gpu.py -x -q > gpurun_out/dp_test.log 2>&1; echo "test rc=$?"
tail -15 gpurun_out/dp_test.log
bash tools/dp_rehearse.sh
