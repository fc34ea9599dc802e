#!/bin/bash
# bf16: kept forward copy read by the weight gradient -- op tests, step test, A/B of the steps
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3bf16a}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_unet_config5_gpu.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do for e in 0 1; do for w in lite183 full185 unet132; do
  E2_BF16_XKEEP=$e timeout -k 10 300 python bench.py --workload $w --mfma bf16 --steps 40 --warmup 8 --no-cpu-baseline > $O/b_${w}_$e.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
  python -c "import json; d=json.load(open('$O/b_${w}_$e.json')); print('xkeep=$e $w bf16', round(d['ms_per_step'],4), 'ms')"
done; done; done
