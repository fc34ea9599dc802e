"""Whole-step refinement of the shipped tilings (GPU box).

The tuner ranks the tilings of a conv launch by timing that launch alone, back to back,
with its operands warm in the caches.  Inside the captured training step the ranking is
not always the same (the operands come from the kernel before, the next kernel's prologue
overlaps the tail) -- round 3 measured re-tuned tables that were faster launch by launch
and slower as a step.  This tool settles it where it counts: for every conv problem of a
workload, the best few tilings of the isolated ranking are tried IN the step (one eager
run + capture + replays each), and a tiling is kept only if the step gets faster.

usage: python tools/tune_insitu.py <workload> [top_k] [replays]
       E2HIP_TUNE_CACHE=<file> receives the refined table (tools/retune.sh convention).
The problems' isolated rankings come from a fresh tuning of the launches that the
shipped table does not hold; to refine ALL problems drop them first (tools/retune.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from elektronn2_amd import autotune, nets


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "lite183"
    top_k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    replays = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    builder, sp, _ = bench.WORKLOADS[wl]
    np.random.seed(1)
    model = getattr(nets, builder)((None, 1) + sp)
    osp = tuple(model.prediction_node.shape.spatial_shape)
    model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    opt = model.optimisers['Adam']
    opt.step.compile()
    plan = opt.step.func
    rng = np.random.RandomState(0)
    x = torch.tensor(rng.rand(1, 1, *sp).astype(np.float32), device="cuda")
    t = torch.tensor(rng.randint(0, 2, (1, 1) + osp).astype(np.float32), device="cuda")
    plan.set_inputs([x, t])
    ctx = plan.ctx

    def step():
        with torch.cuda.stream(plan.stream):
            opt._ensure_state(plan)
            opt._sync_hyper(plan)
        plan.run()

    def measure():
        """ms per replayed step with the tilings the cache holds now (re-plans, re-captures)"""
        for g in (plan._graphs or []):
            ctx.graph_destroy(g)
        plan._graphs = None
        plan._calls = 0
        step(); step()                       # eager (notes the zero fills), then capture
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            e0, e1 = ctx.event(), ctx.event()
            old = ctx.stream
            ctx.set_stream(plan.stream); ctx.record(e0); ctx.set_stream(old)
            for _ in range(replays):
                step()
            ctx.set_stream(plan.stream); ctx.record(e1); ctx.set_stream(old)
            torch.cuda.synchronize()
            best = min(best, ctx.elapsed_ms(e0, e1) / replays)
        return best

    step()                                   # tunes whatever the table does not hold
    torch.cuda.synchronize()
    cache = autotune._load()
    keys = [k for k in autotune.last_ranking if len(autotune.last_ranking[k]) > 1]
    base = measure()
    print("%s: %d problems tuned now, step %.4f ms with the isolated winners" % (wl, len(keys), base), flush=True)
    # the problems whose launches take longest first
    keys.sort(key=lambda k: -autotune.last_ranking[k][0][0])
    for k in keys:
        rank = autotune.last_ranking[k]
        cur = cache[k]
        for tm, cand in rank[:top_k]:
            if cand == cur or not cand:
                continue
            if cand.startswith("13,2,"):      # (igemm_kernel<13,2>: operand registers spill)
                continue
            cache[k] = cand
            try:
                ms = measure()
            except Exception as err:          # a tiling the step cannot use (fused epilogue ...)
                print("   %s %s: %s" % (k, cand, str(err)[:80]))
                ms = float("inf")
            if ms < base - 0.002:
                print("   %s: %s -> %s  step %.4f -> %.4f ms" % (k, cur, cand, base, ms), flush=True)
                base, cur = ms, cand
            cache[k] = cur
    cache.update({})
    final = measure()
    autotune._dirty = True
    autotune.save()
    print("%s refined: %.4f ms per step" % (wl, final))


if __name__ == "__main__":
    main()
