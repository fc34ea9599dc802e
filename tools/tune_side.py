"""f32 mode: which weight gradients of a workload should run on the plan's side stream?

The f32 step keeps ONE stream (two GEMMs whose work-groups each take a CU's whole register file
thrash when they share the chip: DESIGN findings 7, 54) -- but for some layers a weight gradient
beside the data-gradient chain does pay (neuro3d_lite's 200-channel layers: 220 work-groups of
~100 us each on 256 CUs).  Like a tiling, that is decided by measurement inside the captured
step and kept per PROBLEM in the tuning table ("side|<wgrad signature>": "1").

usage: python tools/tune_side.py <workload> [max_run=4] [steps=40]
       E2_TUNE_SIDE_WINDOW=lo,hi  every subset of the layers lo .. hi - 1 (a net too deep for all subsets)
       E2_TUNE_SIDE_EXHAUSTIVE=n  nets of up to n layers are searched exhaustively (default 9)
       E2_TUNE_SIDE_ROUNDS=n      greedy rounds for deeper nets (default 6)
       greedy: tries every run of consecutive conv layers (length 1 .. max_run) as a set whose weight
       gradients go to the side stream, keeps the best if it gains > 0.2 %, tries to add another run
       on top (up to six rounds), re-measures the result three times against none, interleaved, and
       writes its flags to $E2HIP_TUNE_CACHE if it wins every time by > 0.3 %."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
from elektronn2_amd import autotune, nets, neuromancer as nm

wl = sys.argv[1] if len(sys.argv) > 1 else "lite183"
max_run = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
builder, sp, _ = B.WORKLOADS[wl]


def measure(mask):
    """ms per step of the captured training step with the weight gradients of `mask` on the side stream"""
    nm.model_manager.reset()
    np.random.seed(1)
    with nm.plan_options(side_mask=mask, side_table=False):
        m = getattr(nets, builder)((None, 1) + sp)
        m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
        osp = tuple(m.prediction_node.shape.spatial_shape)
        rng = np.random.RandomState(0)
        x = rng.rand(1, 1, *sp).astype(np.float32)
        t = rng.randint(0, 2, (1, 1) + osp).astype(np.float32)
        for _ in range(3):
            m.trainingstep(x, t, optimiser='Adam')
    plan = m.optimisers['Adam'].step.func
    ctx = plan.ctx
    for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    old = ctx.stream
    ctx.set_stream(plan.stream); ctx.record(e0); ctx.set_stream(old)
    for _ in range(steps):
        plan.run()
    ctx.set_stream(plan.stream); ctx.record(e1); ctx.set_stream(old)
    torch.cuda.synchronize()
    ms = ctx.elapsed_ms(e0, e1) / steps
    order = plan._side_order if plan._side_order is not None else []
    if not order:
        plan.side_rank(None)
        order = plan._side_order
    sigs = [n._sig_wgrad(plan) for n in order]
    names = [n.name for n in order]
    del m, plan
    torch.cuda.empty_cache()
    return ms, sigs, names


base, sigs, names = measure(0)
n = len(sigs)
print("%s: %d conv layers with a weight-gradient launch (%s); no side stream: %.4f ms" % (wl, n, " ".join(names), base), flush=True)
def finals(cands):
    """re-measure the best few masks against none, interleaved; -> (mask, wins) of the best average"""
    res = {}
    for mask in cands:
        pairs = []
        for _ in range(3):
            a, _, _ = measure(0)
            b, _, _ = measure(mask)
            pairs.append((a, b))
        res[mask] = pairs
        print("  final: mask %6d  %s" % (mask, "  ".join("%.4f / %.4f" % p for p in pairs)), flush=True)
    best = min(res, key=lambda m: sum(b - a for a, b in res[m]))
    return best, res[best]


WIN = os.environ.get("E2_TUNE_SIDE_WINDOW")            # "lo,hi": every subset of the layers lo .. hi - 1 only
if WIN:
    lo_w, hi_w = (int(v) for v in WIN.split(","))
    table = []
    for sub in range(1, 1 << (hi_w - lo_w)):
        mask = sub << lo_w
        ms, _, _ = measure(mask)
        table.append((ms, mask))
        print("  mask %6d  %.4f ms (%+.1f us)" % (mask, ms, (ms - base) * 1e3), flush=True)
    table.sort()
    cur, wins = finals([m for _, m in table[:5]])
elif n <= int(os.environ.get("E2_TUNE_SIDE_EXHAUSTIVE", "9")):
    # few layers: every subset (the gains are not additive -- a chain on the side stream shares one
    # join, a lone launch pays it alone -- so greedy steps miss sets like neuro3d's {1, 5, 7, 8, 9})
    table = []
    for mask in range(1, 1 << n):
        ms, _, _ = measure(mask)
        table.append((ms, mask))
        print("  mask %6d  %.4f ms (%+.1f us)" % (mask, ms, (ms - base) * 1e3), flush=True)
    table.sort()
    cur, wins = finals([m for _, m in table[:5]])
else:
    # greedy over runs of consecutive layers: add the run that helps most, measure again on top of it
    cur, cur_ms = 0, base
    for it in range(int(os.environ.get("E2_TUNE_SIDE_ROUNDS", "6"))):
        best = (cur_ms, cur)
        for length in range(1, max_run + 1):
            for lo in range(0, n - length + 1):
                run = ((1 << length) - 1) << lo
                if run & cur:
                    continue
                ms, _, _ = measure(cur | run)
                print("  %d: + %-36s mask %6d  %.4f ms (%+.1f us)" % (it, " ".join(names[lo:lo + length]), cur | run, ms,
                                                                    (ms - cur_ms) * 1e3), flush=True)
                if ms < best[0]:
                    best = (ms, cur | run)
        if best[1] == cur or best[0] > cur_ms * 0.998:
            break
        # (a step is kept only if it shows again)
        a2, _, _ = measure(cur)
        b2, _, _ = measure(best[1])
        if b2 > a2 * 0.9985:
            print("  -> mask %d did not repeat (%.4f against %.4f)" % (best[1], b2, a2), flush=True)
            break
        cur_ms, cur = min(best[0], b2), best[1]
        print("  -> mask %d: %.4f ms" % (cur, cur_ms), flush=True)
    wins = []
    if cur:
        cur, wins = finals([cur])
if cur == 0:
    print("nothing beats the single stream")
    sys.exit(0)
if all(b < a * 0.997 for a, b in wins):
    for r in range(n):
        if (cur >> r) & 1:
            autotune.set_side_flag(sigs[r], True)
            print("  side|%s = 1   (%s)" % (",".join(str(int(v)) for v in sigs[r]), names[r]))
    autotune.save()
    print("written to", os.environ.get("E2HIP_TUNE_CACHE", "(the default cache)"))
else:
    print("mask %d does not win every round by > 0.3 %%: nothing written" % cur)
