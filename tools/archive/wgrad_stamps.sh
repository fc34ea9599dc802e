#!/bin/bash
# in-kernel timeline of the wgrad kernel for the main layers (debug-env build on the box)
# the debug-switch build lives in a directory of its own (the product library is never overwritten)
cd $GRAFT_REPO_ROOT/elektronn2_amd/csrc && make -j16 BUILD=build/dbg OUT=build/dbg/libe2hip.so DEBUG_ENV=1 > /dev/null 2>&1 || { echo build failed; exit 1; }
export E2HIP_LIB=$GRAFT_REPO_ROOT/elektronn2_amd/csrc/build/dbg/libe2hip.so
cd $GRAFT_REPO_ROOT
run() { # cin cout kd kh kw D H W force
  E2_WGRAD_STAMPS=1 E2_VERBOSE=1 E2_WGRAD_FORCE="$9" python tools/one_layer.py wgradp $1 $2 $3 $4 $5 $6 $7 $8 5 2>&1 | grep -E "stamps|wgradp" | tail -2
}
run 200 200 1 3 3 10 39 39 "7,2,1,128,8"
run 150 200 1 3 3 10 41 41 "7,2,1,128,11"
run 40 150 2 4 4 21 44 44 "2,2,1,256,10"
run 20 40 3 3 3 23 90 90 "3,2,14,256,30"
run 40 80 4 4 4 23 39 39 "5,2,1,256,12"
run 80 100 3 4 4 10 36 36 "7,2,1,256,8"
run 100 100 3 4 4 8 33 33 "7,2,1,256,6"
echo "---- three-set pipeline"
run 40 150 2 4 4 21 44 44 "2,2,21,256,10"
run 20 40 3 3 3 23 90 90 "3,2,24,256,30"
run 20 40 3 3 3 23 90 90 "3,2,21,256,30"
run 40 80 4 4 4 23 39 39 "4,2,21,256,12"
