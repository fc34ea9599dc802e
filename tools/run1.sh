cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_warp.py -x -q -k sampler 2>&1 | tail -12
