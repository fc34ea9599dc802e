"""Try the position-split GEMM forms of the weight gradients ("MT,NT,9,0,S" / "MT,NT,8,0,S")
INSIDE the captured training step of a workload, problem by problem, against the shipped tiling
(tools/tune_insitu.py does this for a full re-tune; this keeps every other entry as shipped).
usage: python tools/ks_insitu.py <workload> [replays] [min gain in ms]
prints the entries to put into elektronn2_amd/tuned.json."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from elektronn2_amd import autotune, nets
from elektronn2_amd.neuromancer import bf16_ahead


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "lite183"
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    min_gain = float(sys.argv[3]) if len(sys.argv) > 3 else 0.003
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
    base = measure()
    print("%s: step %.4f ms with the shipped table" % (wl, base), flush=True)
    changed = {}
    for node in bf16_ahead.conv_nodes(plan):
        if (node, 'dy') not in plan.scratch or node._fused_first(plan) or node._fused_head(plan) is not None:
            continue
        sig = node._sig_wgrad(plan)
        key = "wgrad|" + ",".join(str(int(v)) for v in sig)
        cur = cache.get(key)
        if cur is None:
            continue
        k = tuple(node._k3)
        cin = node.parent.shape['f']
        cands = autotune.position_split_wgrad_candidates(node.n_f, cin, k, sig[5:8])
        cands += [c for c in autotune.pointwise_wgrad_candidates(node.n_f, cin, k, sig[5:8]) if ",8,0," in c]
        best_c, best_ms = cur, base
        for cand in cands:
            if cand == cur:
                continue
            cache[key] = cand
            try:
                ms = measure()
            except Exception as err:
                print("   %s %s: %s" % (key, cand, str(err)[:80]))
                ms = float("inf")
            if ms < best_ms - min_gain:
                best_c, best_ms = cand, ms
        cache[key] = best_c
        if best_c != cur:
            print('   "%s": "%s",   (was %s; step %.4f -> %.4f ms)' % (key, best_c, cur, base, best_ms), flush=True)
            changed[key] = best_c
            base = best_ms
    # UpConv nodes: their weight gradient is the 1x1x1 GEMM over the space-to-depth image
    from elektronn2_amd.neuromancer.neural import UpConv
    for node in plan.nodes:
        if type(node) is not UpConv or not plan.training:
            continue
        sig, cw = node._tune_sigs(plan)['wgrad']
        key = "wgrad|" + ",".join(str(int(v)) for v in sig)
        cur = cache.get(key)
        if cur is None:
            continue
        best_c, best_ms = cur, base
        for cand in [c for c in cw if ",8,0," in c]:
            if cand == cur:
                continue
            cache[key] = cand
            try:
                ms = measure()
            except Exception as err:
                print("   %s %s: %s" % (key, cand, str(err)[:80]))
                ms = float("inf")
            if ms < best_ms - min_gain:
                best_c, best_ms = cand, ms
        cache[key] = best_c
        if best_c != cur:
            print('   "%s": "%s",   (was %s; step %.4f -> %.4f ms)' % (key, best_c, cur, base, best_ms), flush=True)
            changed[key] = best_c
            base = best_ms
    final = measure()
    print("%s with %d entries changed: %.4f ms per step" % (wl, len(changed), final))
    import json
    print(json.dumps(changed, indent=0, sort_keys=True))


if __name__ == "__main__":
    main()
