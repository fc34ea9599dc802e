cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -3
timeout -k 10 300 python bench.py --workload unet_lite140 2>&1 | tail -1 | cut -c1-900
