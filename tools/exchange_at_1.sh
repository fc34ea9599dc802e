#!/bin/bash
# what the data-parallel machinery itself costs (GPU box): the plain step against the segmented
# step + collectives over a ONE-rank RCCL group, interleaved:  tools/exchange_at_1.sh <tag>
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-ex1}; mkdir -p $O
for r in 1 2; do for w in lite183 full185; do for f in "" "--exchange-at-1"; do
  python bench.py --workload $w $f --no-cpu-baseline --no-also --steps 40 --warmup 8 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "import json; d=json.load(open('$O/b.json')); print('$w [$f] %.4f ms (dev %.4f)' % (d['ms_per_step'], d['roofline']['device_ms_per_step']))"
done; done; done
