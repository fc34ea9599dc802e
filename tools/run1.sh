cd $GRAFT_REPO_ROOT
export E2_DIST_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 2>&1 | tail -4 | cut -c1-700
