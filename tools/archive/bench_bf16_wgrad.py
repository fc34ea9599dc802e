"""time e2_conv3d_wgrad_bf16 (conversion pass included) on one layer next to the f32 and the
operand-rounding bf16 wgrad:  python tools/bench_bf16_wgrad.py cin cout kd kh kw D H W"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from elektronn2_amd import backend, autotune

cin, cout, kd, kh, kw, D, H, W = map(int, sys.argv[1:9])
ctx = backend.Context(0)
k = (kd, kh, kw)
osp = (D - kd + 1, H - kh + 1, W - kw + 1)
x = torch.rand(1, cin, D, H, W, device="cuda")
pshape = (1, cout) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
dyp = flat[:int(np.prod(pshape))].view(pshape)
dyv = dyp[:, :, kd - 1:kd - 1 + osp[0], kh - 1:kh - 1 + osp[1], kw - 1:kw - 1 + osp[2]]
dyv.copy_(torch.randn(1, cout, *osp, device="cuda"))
dw = torch.zeros(cout, cin, *k, device="cuda")
gf = 2.0 * cout * cin * kd * kh * kw * osp[0] * osp[1] * osp[2] / 1e9
print("wgrad %s: %.2f GF; f32 MFMA ideal %.1f us, bf16 ideal %.1f us" % (sys.argv[1:9], gf, gf / 157.3 * 1e3, gf / 2500 * 1e3))
ctx.set_mfma_dtype('bf16')
t = autotune._time(ctx, lambda: ctx.conv3d_wgrad_pad(x, dyp, dw, accumulate=False), iters=10)
print("   bf16 operand rounding (own tiling)        %8.1f us" % (t * 1e3))
ctx.set_mfma_dtype('f32')
nbs = sorted({min(kw, n) for n in (1, 2, 3, 4)})
forms = ([0] if cin >= 48 else []) + ([1] if cin <= 112 else [])
for mb, nb, r in [(mb, nb, r) for r in forms for mb in (1, 2) for nb in nbs]:
    if True:
        for s in (0, 8, 16):
            tile = "32,%d,%d,%d,%d" % (mb, nb, r, s)
            ctx.set_tiling("wgrad", tile)
            t = autotune._time(ctx, lambda: ctx.conv3d_wgrad_bf16(x, dyv, dw, accumulate=True), iters=10)
            print("   bf16 in memory %-12s %8.1f us (%5.1f%% of 2.5 PF)" % (tile, t * 1e3, gf / t / 2500 * 100))
ctx.set_tiling("wgrad", None)
