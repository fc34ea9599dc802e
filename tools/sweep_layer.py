"""Time every tiling candidate of one conv fwd / dgrad problem (both kernels):
usage: python tools/sweep_layer.py <fwd|dgrad> cin cout kd kh kw D H W [top]
(D H W = INPUT extent of the forward conv)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from elektronn2_amd import backend, autotune

op = sys.argv[1]
cin, cout, kd, kh, kw, D, H, W = map(int, sys.argv[2:10])
top = int(sys.argv[10]) if len(sys.argv) > 10 else 6
ctx = backend.Context(0)
k = (kd, kh, kw)
osp = (D - kd + 1, H - kh + 1, W - kw + 1)
x = torch.rand(1, cin, D, H, W, device="cuda")
w = torch.randn(cout, cin, *k, device="cuda") * 0.05
y = torch.empty(1, cout, *osp, device="cuda")
pshape = (1, cout) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
dyp = flat[:int(np.prod(pshape))].view(pshape)
dyp[:, :, kd - 1:kd - 1 + osp[0], kh - 1:kh - 1 + osp[1], kw - 1:kw - 1 + osp[2]] = torch.randn(1, cout, *osp, device="cuda")
dx = torch.empty_like(x)
ws = torch.empty(ctx.conv_ws_bytes(cout, cin, k) // 4 + 64, device="cuda")
ctx.conv3d_pack(w, 0 if op == "fwd" else 1, ws)
if op == "fwd":
    fn = lambda: ctx.conv3d_fwd_packed(x, ws, cout, k, y)
    cands = autotune.igemm_candidates(cout, cin, k, osp)
else:
    fn = lambda: ctx.conv3d_dgrad_packed(dyp, ws, cin, k, dx)
    cands = autotune.igemm_candidates(cin, cout, k, (D, H, W))
gf = 2.0 * cout * cin * kd * kh * kw * osp[0] * osp[1] * osp[2] / 1e9
res = []
for c in cands:
    ctx.set_tiling("igemm", c)
    try:
        t = autotune._time(ctx, fn, iters=6)
    except backend.E2Error:
        continue
    res.append((t * 1e3, c))
ctx.set_tiling("igemm", None)
for kind, sel in (("16x16x4", [r for r in res if r[1].count(",") in (3, 4)]),
                  ("4x4x1", [r for r in res if r[1].count(",") >= 5])):
    sel.sort()
    print("%s %s %s: %.2f GF, ideal %.1f us" % (op, sys.argv[2:10], kind, gf, gf / 157.3 * 1e3))
    for t, c in sel[:top]:
        print("   %-18s %8.1f us  %5.1f%% of peak" % (c, t, gf / t * 1e3 / 157.3 * 100))
