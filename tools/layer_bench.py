"""Per-layer kernel timing of the neuro3d nets on one MI355X (HIP events).
usage: python tools/layer_bench.py [lite|full] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from elektronn2_amd import backend

LITE = [(20, (1, 4, 4), (1, 2, 2)), (40, (3, 3, 3), (1, 2, 2)), (150, (2, 4, 4), (2, 1, 1)),
        (200, (1, 3, 3), (1, 1, 1)), (200, (1, 3, 3), (1, 1, 1)), (200, (1, 1, 1), (1, 1, 1)),
        (2, (1, 1, 1), (1, 1, 1))]
FULL = [(20, (1, 6, 6), (1, 2, 2)), (30, (1, 5, 5), (1, 2, 2)), (40, (1, 5, 5), (1, 1, 1)),
        (80, (4, 4, 4), (2, 1, 1)), (100, (3, 4, 4), (1, 1, 1)), (100, (3, 4, 4), (1, 1, 1)),
        (150, (2, 4, 4), (1, 1, 1)), (200, (1, 4, 4), (1, 1, 1)), (200, (1, 4, 4), (1, 1, 1)),
        (200, (1, 1, 1), (1, 1, 1)), (2, (1, 1, 1), (1, 1, 1))]
PEAK = 157.3


def timeit(ctx, fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(iters):
        fn()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / iters * 1e3   # us


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "lite"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    spec, sp = (LITE, (23, 183, 183)) if which == "lite" else (FULL, (23, 185, 185))
    ctx = backend.Context(0)
    cin = 1
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "pw": 0.0}
    totgf = 0.0
    print("%-3s %-22s %9s | %8s %6s | %8s %6s | %8s %6s | %7s %7s" % (
        "L", "geom", "GF", "fwd us", "TF", "dgrad us", "TF", "wgrad us", "TF", "pool_f", "pool_b"))
    for li, (nf, k, p) in enumerate(spec):
        osp = tuple(sp[i] - k[i] + 1 for i in range(3))
        psp = tuple(osp[i] // p[i] for i in range(3))
        x = torch.rand(1, cin, *sp, device="cuda")
        w = torch.randn(nf, cin, *k, device="cuda") * 0.05
        b = torch.rand(nf, device="cuda")
        y = torch.empty(1, nf, *osp, device="cuda")
        out = torch.empty(1, nf, *psp, device="cuda")
        dout = torch.randn(1, nf, *psp, device="cuda")
        dyp = torch.zeros(1, nf, *[osp[i] + 2 * (k[i] - 1) for i in range(3)], device="cuda")
        dy = dyp[:, :, k[0] - 1:k[0] - 1 + osp[0], k[1] - 1:k[1] - 1 + osp[1], k[2] - 1:k[2] - 1 + osp[2]]
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.zeros(nf, device="cuda")
        ws = torch.empty(ctx.conv_ws_bytes(nf, cin, k) // 4 + 64, device="cuda")
        ws2 = torch.empty_like(ws)
        ctx.conv3d_pack(w, 0, ws)
        ctx.conv3d_pack(w, 1, ws2)
        gf = 2.0 * nf * cin * np.prod(k) * np.prod(osp) / 1e9
        t_f = timeit(ctx, lambda: ctx.conv3d_fwd_packed(x, ws, nf, k, y), iters)
        t_p = timeit(ctx, lambda: ctx.pool_bias_act_fwd(y, b, p, 'relu', out), iters)
        t_pb = timeit(ctx, lambda: ctx.pool_bias_act_bwd(dout, y, b, p, 'relu', dy, db), iters)
        t_d = timeit(ctx, lambda: ctx.conv3d_dgrad_packed(dyp, ws2, cin, k, dx), iters) if li > 0 else 0.0
        t_w = timeit(ctx, lambda: ctx.conv3d_wgrad(x, dy, dw), iters)
        tf = lambda t: gf / t * 1e3 / 1e3 if t > 0 else 0.0   # GF/us*1e3 = TF
        print("%-3d %-22s %9.3f | %8.1f %6.1f | %8.1f %6.1f | %8.1f %6.1f | %7.1f %7.1f" % (
            li, "%d>%d k%s p%s" % (cin, nf, "".join(map(str, k)), "".join(map(str, p))), gf,
            t_f, tf(t_f), t_d, tf(t_d), t_w, tf(t_w), t_p, t_pb))
        tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w; tot["pw"] += t_p + t_pb
        totgf += gf * (3 if li > 0 else 2)
        cin, sp = nf, psp
    tall = sum(tot.values())
    print("total us: fwd %.1f dgrad %.1f wgrad %.1f pointwise %.1f  = %.1f us ; %.2f GF -> %.1f TF/s = %.1f%% of %.1f"
          % (tot["fwd"], tot["dgrad"], tot["wgrad"], tot["pw"], tall, totgf, totgf / tall * 1e3 / 1e3,
             totgf / tall / PEAK * 100, PEAK))


if __name__ == "__main__":
    main()
