#!/bin/bash
# in-kernel timelines of neuro3d's small late layers + lite's big ones (debug-switch build)
cd $GRAFT_REPO_ROOT
export E2HIP_LIB=$GRAFT_REPO_ROOT/elektronn2_amd/csrc/build/dbg/libe2hip.so
run() { # op cin cout kd kh kw D H W force
  E2_IGEMM_STAMPS=1 E2_VERBOSE=1 E2_IGEMM_FORCE="${10}" python tools/one_layer.py $1 $2 $3 $4 $5 $6 $7 $8 $9 5 2>&1 | grep -E "stamps\]|TF" | tail -3
}
run fwd 150 200 1 4 4 5 27 27 "7,1,28,8"
run fwd 150 200 1 4 4 5 27 27 "13,1,32,5"
run fwd 200 200 1 4 4 5 24 24 "13,1,32,8"
run dgrad 200 200 1 4 4 5 24 24 "7,1,40,6"
run fwd 100 100 3 4 4 8 33 33 "7,2,20,5"
run dgrad 100 100 3 4 4 8 33 33 "7,1,48,5"
run fwd 80 100 3 4 4 10 36 36 "7,2,12,4"
run fwd 40 150 2 4 4 21 44 44 "2,4,16,1"
run fwd 20 40 3 3 3 23 90 90 "3,4,12,1"
