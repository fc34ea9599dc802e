#!/bin/bash
# inter-kernel gaps of the graph-replayed step from a rocprofv3 kernel trace (tools/trace_gaps.py)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r3gaps}
mkdir -p $O
for w in lite183 full185; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$w -- python3 bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$w.json 2> $O/err_$w.txt || { tail -20 $O/err_$w.txt; exit 1; }
python -c "import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],4), 'ms under the trace')"
python tools/trace_gaps.py $O/tr_$w | tee $O/gaps_$w.txt
rm -rf $O/tr_$w
done
