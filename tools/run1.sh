cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload dense183 --steps 3 --warmup 1 2>&1 | tail -1
