cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py tests/test_autotune_gpu.py -x -q 2>&1 | tail -3 || exit 1
export E2_IGEMM_STAMPS=1
echo L5fwd 13,1,48; E2_IGEMM_FORCE=13,1,48,1 python tools/one_layer.py fwd 200 200 1 1 1 10 37 37 3 2>&1 | grep stamps | tail -1
echo L4fwd; E2_IGEMM_FORCE=7,2,64,1 python tools/one_layer.py fwd 200 200 1 3 3 10 39 39 3 2>&1 | grep stamps | tail -1
echo L1fwd; E2_IGEMM_FORCE=3,4,20,1 python tools/one_layer.py fwd 20 40 3 3 3 23 90 90 3 2>&1 | grep stamps | tail -1
unset E2_IGEMM_STAMPS
timeout -k 10 300 python bench.py --no-cpu-baseline --workload lite183 2>&1 | tail -1 | cut -c1-330
