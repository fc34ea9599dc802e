#!/bin/bash
cd $GRAFT_REPO_ROOT/elektronn2_amd/csrc && make -j16 BUILD=build/dbg OUT=build/dbg/libe2hip.so DEBUG_ENV=1 > /dev/null 2>&1 || { echo build failed; exit 1; }
export E2HIP_LIB=$GRAFT_REPO_ROOT/elektronn2_amd/csrc/build/dbg/libe2hip.so
cd $GRAFT_REPO_ROOT
for dbg in 0 1; do
for f in "2,2,1,256,10" "2,2,21,256,10" "5,2,1,256,10"; do
  echo -n "dbg=$dbg "
  E2_WGRAD_DBG=$dbg E2_WGRAD_STAMPS=1 E2_WGRAD_FORCE="$f" python tools/one_layer.py wgradp 40 150 2 4 4 21 44 44 5 2>&1 | grep -E "stamps" | tail -1 | cut -c1-260
done; done
