#!/bin/bash
# kernel-time breakdown of the f32 U-Net steps
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3unet}
mkdir -p $O
for w in unet_lite140 unet132; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/err_$w.txt || { tail -20 $O/err_$w.txt; exit 1; }
cp $(find $O/prof_$w -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$w.csv
python tools/trace_step.py $O/prof_$w > $O/step_$w.txt
rm -rf $O/prof_$w
python -c "import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],4), 'ms')"
done
