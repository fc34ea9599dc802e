#!/bin/bash
# FETCH_SIZE and time of the direct wgrad with / without the XCD-grouped block order
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/xcd
run() { # tag cin cout kd kh kw D H W force
  tag=$1; shift
  f=${9}
  E2_WGRAD_FORCE="$f" python tools/one_layer.py wgradp $1 $2 $3 $4 $5 $6 $7 $8 20 2>&1 | tail -1
  E2_WGRAD_FORCE="$f" timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/xcd/$tag -- python3 tools/one_layer.py wgradp $1 $2 $3 $4 $5 $6 $7 $8 5 > /dev/null 2>&1
  python tools/pmc_sum.py gpurun_out/xcd/$tag FETCH_SIZE 8 | grep wgrad | cut -c1-20,100-200
  rm -rf gpurun_out/xcd/$tag
}
run a 200 200 1 3 3 10 39 39 "7,2,1,128,8"
run b 200 200 1 3 3 10 39 39 "7,2,101,128,8"
run c 150 200 1 3 3 10 41 41 "7,2,1,128,11"
run d 150 200 1 3 3 10 41 41 "7,2,101,128,12"
run e 40 150 2 4 4 21 44 44 "2,2,1,256,10"
run f 40 150 2 4 4 21 44 44 "2,2,101,256,8"
run g 20 40 3 3 3 23 90 90 "3,2,14,256,30"
run h 20 40 3 3 3 23 90 90 "3,2,114,256,32"
