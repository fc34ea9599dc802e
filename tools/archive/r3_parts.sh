#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3g}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "split_k or dgrad_skips or fwd_dgrad_wgrad or forced" > $O/ops.log 2>&1 || { tail -40 $O/ops.log; exit 1; }
tail -2 $O/ops.log
timeout -k 10 900 python -m pytest tests/test_native_size_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_unet_config5_gpu.py -x -q -m gpu > $O/model.log 2>&1 || { tail -40 $O/model.log; exit 1; }
tail -2 $O/model.log
for w in lite183 full185; do
  timeout -k 10 300 python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || { tail -20 $O/bench_$w.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done
