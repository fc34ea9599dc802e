#!/bin/bash
# timing ablations of the 4x4x1 kernel (GPU box): rebuild conv_igemm4_k3 with E2_G4_ABLATE=n
# into a scratch copy of the library and time one problem with one tiling
#   tools/ablate_igemm4.sh "<one_layer args>" "<tiling>"
cd $GRAFT_REPO_ROOT/elektronn2_amd/csrc
for ab in ${3:-0 1 2 3 4 5 6}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DE2_G4_ABLATE=$ab -c conv_igemm4_k3.hip -o conv_igemm4_k3.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libe2hip.so *.o
  echo -n "ablate=$ab  "
  (cd $GRAFT_REPO_ROOT && E2_IGEMM_FORCE="$2" python tools/one_layer.py $1 20 2>&1 | tail -1)
done
