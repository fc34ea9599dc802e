"""Time every weight-gradient tiling candidate of a 1x1x1 problem (both kernels):
usage: python tools/sweep_pw_wgrad.py cin cout D H W [top]     (UpConv: cout = n_f * prod(pool))"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from elektronn2_amd import backend, autotune

cin, cout, D, H, W = map(int, sys.argv[1:6])
top = int(sys.argv[6]) if len(sys.argv) > 6 else 6
ctx = backend.Context(0)
x = torch.rand(1, cin, D, H, W, device="cuda")
flat = torch.zeros(cout * D * H * W + 32, device="cuda")
dy = flat[:cout * D * H * W].view(1, cout, D, H, W)
dy.copy_(torch.randn(1, cout, D, H, W, device="cuda"))
dw = torch.zeros(cout, cin, 1, 1, 1, device="cuda")
fn = lambda: ctx.conv3d_wgrad_pad(x, dy, dw, accumulate=False)
gf = 2.0 * cout * cin * D * H * W / 1e9
res = []
for c in autotune.wgrad_candidates(cout, cin, (1, 1, 1), (D, H, W)):
    ctx.set_tiling("wgrad", c)
    try:
        t = autotune._time(ctx, fn, iters=6)
    except backend.E2Error:
        continue
    res.append((t * 1e3, c))
ctx.set_tiling("wgrad", None)
for kind, sel in (("direct / staged", [r for r in res if r[1].split(",")[2] not in "78"]),
                  ("K-contiguous GEMM", [r for r in res if r[1].split(",")[2] == "7"]),
                  ("K-contiguous GEMM, waves split the positions", [r for r in res if r[1].split(",")[2] == "8"])):
    sel.sort()
    print("wgrad 1x1x1 %s %s: %.2f GF, ideal %.1f us (incl. the zero fill of dw)" % (sys.argv[1:6], kind, gf, gf / 157.3 * 1e3))
    for t, c in sel[:top]:
        print("   %-18s %8.1f us  %5.1f%% of peak" % (c, t, gf / t * 1e3 / 157.3 * 100))
