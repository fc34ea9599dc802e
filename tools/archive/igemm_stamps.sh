#!/bin/bash
# in-kernel timeline of the 16x16 igemm kernel for the main layers of neuro3d_lite@183 (debug-env build)
# the debug-switch build lives in a directory of its own (the product library is never overwritten)
cd $GRAFT_REPO_ROOT/elektronn2_amd/csrc && make -j16 BUILD=build/dbg OUT=build/dbg/libe2hip.so DEBUG_ENV=1 > /dev/null 2>&1 || { echo build failed; exit 1; }
export E2HIP_LIB=$GRAFT_REPO_ROOT/elektronn2_amd/csrc/build/dbg/libe2hip.so
cd $GRAFT_REPO_ROOT
run() { # op cin cout kd kh kw D H W force
  E2_IGEMM_STAMPS=1 E2_VERBOSE=1 E2_IGEMM_FORCE="${10}" python tools/one_layer.py $1 $2 $3 $4 $5 $6 $7 $8 $9 5 2>&1 | grep -E "stamps|TF" | tail -2
}
run fwd 200 200 1 3 3 10 39 39 "7,2,52,1"
run dgrad 200 200 1 3 3 10 39 39 "7,2,52,1"
run fwd 150 200 1 3 3 10 41 41 "7,2,64,1"
run dgrad 150 200 1 3 3 10 41 41 "2,4,16,1"
run fwd 40 150 2 4 4 21 44 44 "2,4,16,1"
run dgrad 40 150 2 4 4 21 44 44 "3,4,16,3"
run fwd 20 40 3 3 3 23 90 90 "3,4,16,1"
run dgrad 20 40 3 3 3 23 90 90 "2,4,16,1"
run fwd 200 200 1 1 1 10 37 37 "13,1,64,1"
