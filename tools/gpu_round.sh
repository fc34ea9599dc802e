#!/bin/bash
# Profile collection of one workload for the record (run on the GPU box via gpurun):
#   tools/gpu_round.sh <workload> <tag>  ->  gpurun_out/<tag>/  bench.json, kernel_stats.csv,
#   pmc_mfma.csv, pmc_fetch.csv, pmc_write.csv   (copy the ones to keep into profiles/)
set -o pipefail
W=${1:-lite183}; T=${2:-round}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$T
mkdir -p $O
python bench.py --sources-sha > $O/sources_sha16.txt
timeout -k 10 400 python bench.py --workload $W > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_prof.json 2> $O/prof.err || { tail -20 $O/prof.err; exit 1; }
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv && rm -rf $O/prof
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-graph > $O/pmc_mfma.json 2> $O/pmc_mfma.err || { tail -20 $O/pmc_mfma.err; exit 1; }
python tools/pmc_mfma.py $O/pmc_mfma 7 > $O/pmc_mfma.csv && rm -rf $O/pmc_mfma
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-graph > $O/pmc_fetch.json 2> $O/pmc_fetch.err || { tail -20 $O/pmc_fetch.err; exit 1; }
python tools/pmc_sum.py $O/pmc_fetch FETCH_SIZE 7 > $O/pmc_fetch.csv && rm -rf $O/pmc_fetch
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-graph > $O/pmc_write.json 2> $O/pmc_write.err || { tail -20 $O/pmc_write.err; exit 1; }
python tools/pmc_sum.py $O/pmc_write WRITE_SIZE 7 > $O/pmc_write.csv && rm -rf $O/pmc_write
tail -3 $O/pmc_mfma.csv; tail -1 $O/pmc_fetch.csv; tail -1 $O/pmc_write.csv
