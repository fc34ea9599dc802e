cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -8 || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --workload lite183 2>&1 | tail -1
