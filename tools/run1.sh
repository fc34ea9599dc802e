cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
