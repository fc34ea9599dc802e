#!/bin/bash
# one bf16 step's kernel sequence (rocprofv3 --kernel-trace)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3bf16}
mkdir -p $O
for w in ${2:-lite183}; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$w -- python3 bench.py --workload $w --mfma bf16 --steps 10 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> $O/err_$w.txt || { tail -20 $O/err_$w.txt; exit 1; }
python tools/trace_step.py $O/tr_$w > $O/step_${w}_bf16.txt
rm -rf $O/tr_$w
done
