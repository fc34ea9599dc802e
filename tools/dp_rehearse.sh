#!/bin/bash
# 2-rank data-parallel rehearsal on ONE GPU (gloo carries the exchange; RCCL needs one GPU per rank)
set -e
cd /root/repo
export E2_DIST_BACKEND=gloo
for ov in 1 0; do
  E2_DP_OVERLAP=$ov timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 30 --warmup 5 \
    > gpurun_out/dp_ov$ov.log 2>&1
  tail -1 gpurun_out/dp_ov$ov.log
done
