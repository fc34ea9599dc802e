#!/bin/bash
# U-Net launch-count cleanups (UpConv images in the one repack launch, accumulate form of its
# backward, concat slices as outputs / gradients of sole-consumer parents): tests, then A/B
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3ul}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "upconv or pack" > $O/ops.log 2>&1 || { tail -30 $O/ops.log; exit 1; }
tail -2 $O/ops.log
timeout -k 10 1100 python -m pytest tests/test_unet_config5_gpu.py tests/test_native_size_gpu.py tests/test_model_gpu.py tests/test_checkpoint.py tests/test_malis_nll_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do for e in 0 1; do for w in unet_lite140 unet132; do
  E2_UPCONV_PACKED=$e E2_CONCAT_ALIAS=$e timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 6 --no-cpu-baseline > $O/b_${w}_$e.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
  python -c "import json; d=json.load(open('$O/b_${w}_$e.json')); print('new=$e $w', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done; done; done
