#!/bin/bash
# timing ablations of the 4x4x1 kernel (GPU box): conv_igemm4_k3 rebuilt with E2_G4_ABLATE=n into
# a SCRATCH copy of the library (build/ablate; the product objects and libe2hip.so are never
# touched -- ablation 6 computes garbage) and one problem timed with one tiling
#   tools/ablate_igemm4.sh "<one_layer args>" "<tiling>" ["0 1 2 ..."]
cd $GRAFT_REPO_ROOT/elektronn2_amd/csrc
make -j16 BUILD=build/ablate OUT=build/ablate/libe2hip.so > /dev/null 2>&1 || { echo build failed; exit 1; }
export E2HIP_LIB=$GRAFT_REPO_ROOT/elektronn2_amd/csrc/build/ablate/libe2hip.so
for ab in ${3:-0 1 2 3 4 5 6}; do
  rm -f build/ablate/conv_igemm4_k3.o
  make BUILD=build/ablate OUT=build/ablate/libe2hip.so EXTRA=-DE2_G4_ABLATE=$ab > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo -n "ablate=$ab  "
  (cd $GRAFT_REPO_ROOT && E2_IGEMM_FORCE="$2" python tools/one_layer.py $1 20 2>&1 | tail -1)
done
