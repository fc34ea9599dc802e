"""A/B of tilings INSIDE the captured training step, interleaved on one box (GPU box):

    python tools/ab_tiling.py <workload> "<cache key>" "<tiling A>" "<tiling B>" [...more] [--rounds 3]

e.g.  python tools/ab_tiling.py lite183 "igemm|2,200,150,1,3,3,10,39,39,41" 7,2,40,1 7,2,52,1
Each tiling is put into the tuning table, the step is re-planned / re-captured and 40 replays
are timed (best of 3), `rounds` times round robin.  (a re-tune that was
never A/B-ed in the step cost 13 us.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from elektronn2_amd import autotune, nets


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = 3
    for i, a in enumerate(sys.argv):
        if a == "--rounds":
            rounds = int(sys.argv[i + 1]); args.remove(sys.argv[i + 1])
    wl, key, cands = args[0], args[1], args[2:]
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

    def measure(replays=40):
        for g in (plan._graphs or []):
            ctx.graph_destroy(g)
        plan._graphs = None
        plan._calls = 0
        step(); step()
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

    step()
    torch.cuda.synchronize()
    cache = autotune._load()
    if key not in cache:
        sys.exit("no such key in the tuning table: %s\n(keys of this kind: %s)"
                 % (key, [k for k in cache if k.split('|')[0] == key.split('|')[0]][:8]))
    shipped = cache[key]
    res = {c: [] for c in cands}
    for r in range(rounds):
        for c in cands:
            cache[key] = c
            res[c].append(measure())
    cache[key] = shipped
    for c in cands:
        print("%s %s = %-14s %s  min %.4f ms%s" % (wl, key, c, " ".join("%.4f" % v for v in res[c]), min(res[c]),
                                                    "  (shipped)" if c == shipped else ""), flush=True)


if __name__ == "__main__":
    main()
