"""Layer <-> kernel <-> ideal us <-> measured us of one bench workload's training step.

usage: python tools/layer_table.py <workload> [iters]      (on the GPU box)

Builds the workload as bench.py does, runs one eager step (tilings come from tuned.json /
the tuner), then a second eager step in which every conv GEMM launch of the plan
(forward, data gradient, weight gradient; `Plan.tuned`) is re-run `iters` times between
two HIP events with the tiling the plan uses.  Prints a markdown table: layer, GEMM,
tiling, algorithmic GFLOP, ideal us at 157.3 TFLOP/s, measured us, fraction of peak.
The kernels that are not GEMMs are in the rocprofv3 kernel stats of the same workload
(tools/gpu_round.sh).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from elektronn2_amd import autotune, nets

PEAK = bench.PEAK_FP32_MFMA_TFLOPS


def gflop(kind, sig):
    try:
        if kind == 'wgrad':
            nf, cin, kd, kh, kw, d, h, w = sig[:8]
            return 2.0 * nf * cin * kd * kh * kw * d * h * w / 1e9
        mode, cout, cin, kd, kh, kw, d, h, w = sig[:9]
        if mode == 1:      # data gradient: (1, cin_of_layer, n_f, k, dx dims)
            d, h, w = d - kd + 1, h - kh + 1, w - kw + 1
        return 2.0 * cout * cin * kd * kh * kw * d * h * w / 1e9
    except Exception:
        return float('nan')


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "lite183"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    builder, sp, _ = bench.WORKLOADS[wl]
    np.random.seed(1)
    model = getattr(nets, builder)((None, 1) + sp)
    osp = tuple(model.prediction_node.shape.spatial_shape)
    model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    opt = model.optimisers['Adam']
    opt.step.compile()
    plan = opt.step.func
    plan.use_graph = False
    rng = np.random.RandomState(0)
    x = torch.tensor(rng.rand(1, 1, *sp).astype(np.float32), device="cuda")
    t = torch.tensor(rng.randint(0, 2, (1, 1) + osp).astype(np.float32), device="cuda")
    plan.set_inputs([x, t])

    def step():
        with torch.cuda.stream(plan.stream):
            opt._ensure_state(plan)
            opt._sync_hyper(plan)
        plan.run()
        torch.cuda.synchronize()

    step()                                     # tunes what tuned.json does not hold
    cur, rows, rec = [None], [], [False]
    for n in plan.nodes:
        for ph in ('fwd', 'bwd'):
            f = getattr(n, '_plan_' + ph, None)
            if f is None:
                continue

            def w(plan_, f=f, n=n, ph=ph):
                cur[0] = (n.name, ph)
                return f(plan_)
            setattr(n, '_plan_' + ph, w)
    orig = autotune.tuned_call

    def timed(ctx, kind, sig, cands, fn, allow_tune=True, fn_tune=None, fn_once=None):
        best = orig(ctx, kind, sig, cands, fn, allow_tune=allow_tune, fn_tune=fn_tune,
                    fn_once=fn_once)
        if rec[0]:
            ft = fn_tune if fn_tune is not None else fn
            if best:
                ctx.set_tiling(kind, best)
            try:
                tm = autotune._time(ctx, ft, iters=iters) * 1e3
            finally:
                ctx.set_tiling(kind, None)
            rows.append((cur[0], kind, tuple(int(v) for v in sig), best, tm))
        return best
    autotune.tuned_call = timed
    rec[0] = True
    step()
    rec[0] = False
    autotune.tuned_call = orig

    print("| layer | GEMM | problem (sig) | tiling | GF | ideal us | us | of peak |")
    print("|---|---|---|---|---|---|---|---|")
    tot_gf = tot_t = 0.0
    for (name, ph), kind, sig, best, tm in rows:
        gf = gflop(kind, sig)
        what = 'wgrad' if kind == 'wgrad' else {0: 'fwd', 2: 'fwd+act', 1: 'dgrad'}.get(sig[0], 'igemm')
        ideal = gf / PEAK * 1e3
        print("| %s | %s | %s | %s | %.2f | %.1f | %.1f | %.3f |" % (
            name, what, ",".join(map(str, sig)), best or "(cost model)", gf, ideal, tm,
            ideal / tm if tm > 0 else 0))
        if gf == gf:
            tot_gf += gf
            tot_t += tm
    print("| **all GEMMs** | | | | %.2f | %.1f | %.1f | %.3f |" % (
        tot_gf, tot_gf / PEAK * 1e3, tot_t, tot_gf / PEAK * 1e3 / tot_t))


if __name__ == "__main__":
    main()
