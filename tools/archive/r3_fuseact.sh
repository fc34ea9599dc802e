#!/bin/bash
# relu backward inside the consumer's data-gradient epilogue (E2_FUSE_ACTBWD=1, off by default):
# does it pay on the U-Nets, whose activation tensors are 10-100x neuro3d's?
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3fa}
mkdir -p $O
for rep in 1 2; do for e in 0 1; do for w in unet_lite140 unet132; do
  E2_FUSE_ACTBWD=$e timeout -k 10 400 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/b_${w}_$e.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
  python -c "import json; d=json.load(open('$O/b_${w}_$e.json')); print('fuse_actbwd=$e $w', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done; done; done
