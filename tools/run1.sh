cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -6
