"""time e2_conv3d_fwd_bf16 / dgrad_bf16 (conversion passes included) on one layer, every wave
tile, next to the f32 and the operand-rounding bf16 form:
  python tools/bench_bf16_conv.py cin cout kd kh kw D H W"""
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
w = torch.randn(cout, cin, *k, device="cuda") * 0.05
y = torch.empty(1, cout, *osp, device="cuda")
pshape = (1, cout) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
dyp = torch.zeros(pshape, device="cuda")
dx = torch.empty_like(x)
gf = 2.0 * cout * cin * kd * kh * kw * osp[0] * osp[1] * osp[2] / 1e9
print("layer %s: %.2f GF; f32 MFMA ideal %.1f us, bf16 ideal %.1f us" % (sys.argv[1:9], gf, gf / 157.3 * 1e3, gf / 2500 * 1e3))
ws = torch.empty(ctx.conv_ws_bytes(cout, cin, k) // 4 + 64, device="cuda")
ctx.conv3d_pack(w, 0, ws)
t = autotune._time(ctx, lambda: ctx.conv3d_fwd_packed(x, ws, cout, k, y), iters=10)
print("   f32 igemm (library's own tiling)      %8.1f us" % (t * 1e3))
ctx.set_mfma_dtype('bf16')
t = autotune._time(ctx, lambda: ctx.conv3d_fwd_packed(x, ws, cout, k, y), iters=10)
print("   bf16 operand rounding (own tiling)    %8.1f us" % (t * 1e3))
ctx.set_mfma_dtype('f32')
for tile in ("32,1,1", "32,1,2", "32,2,1", "32,2,2", "32,4,1", "32,1,4", "32,4,2", "32,2,4"):
    ctx.set_tiling("igemm", tile)
    try:
        tf = autotune._time(ctx, lambda: ctx.conv3d_fwd_bf16(x, w, y), iters=10)
        td = autotune._time(ctx, lambda: ctx.conv3d_dgrad_bf16(dyp, w, dx), iters=10)
    except backend.E2Error as e:
        print("   bf16 in memory %-8s refused: %s" % (tile, str(e)[-70:]))
        continue
    print("   bf16 in memory %-8s fwd %8.1f us (%5.1f%% of 2.5 PF)   dgrad %8.1f us" % (tile, tf * 1e3, gf / tf / 2500 * 100, td * 1e3))
ctx.set_tiling("igemm", None)
