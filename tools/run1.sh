cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline --workload lite183 2>&1 | tail -1 | cut -c1-260
timeout -k 10 300 python bench.py --no-cpu-baseline --workload unet_lite140 2>&1 | tail -1 | cut -c1-260
