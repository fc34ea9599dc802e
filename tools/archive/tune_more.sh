#!/bin/bash
# tune whatever the listed workloads still miss in tuned.json (keeps the shipped entries)
cd $GRAFT_REPO_ROOT
T=$1; shift
mkdir -p gpurun_out/$T
export E2HIP_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/$T/tune.json
for w in "$@"; do
  timeout -k 10 900 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$T/bench_$w.json 2> gpurun_out/$T/bench_$w.err || { tail -20 gpurun_out/$T/bench_$w.err; exit 1; }
  python -c "import json,sys; d=json.load(open('gpurun_out/$T/bench_$w.json')); print('$w', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done
