"""Time the position-split GEMM form of a conv weight gradient ("MT,NT,9,0,S") against the
tiling the shipped table / cost model picks, on the plan's padded-gradient layout:
usage: python tools/sweep_wgrad_ks.py cin cout kd kh kw Do Ho Wo  [more tilings ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from elektronn2_amd import backend, autotune

cin, cout, kd, kh, kw, Do, Ho, Wo = map(int, sys.argv[1:9])
extra = sys.argv[9:]
k = (kd, kh, kw)
ctx = backend.Context(0)
D, H, W = Do + kd - 1, Ho + kh - 1, Wo + kw - 1
xf = torch.zeros(cin * D * H * W + 32, device="cuda")
x = xf[:cin * D * H * W].view(1, cin, D, H, W)
x.copy_(torch.rand(1, cin, D, H, W, device="cuda"))
ctx.set_input_slack(128)
pad = [kk - 1 for kk in k]
pshape = (1, cout, Do + 2 * pad[0], Ho + 2 * pad[1], Wo + 2 * pad[2])
pitch = Wo + pad[2]
plane = pshape[3] * pitch
flat = torch.zeros(cout * pshape[2] * plane + pad[2] + 32, device="cuda")
dyp = flat.as_strided(pshape, (cout * pshape[2] * plane, pshape[2] * plane, plane, pitch, 1))
dyp[:, :, pad[0]:pad[0] + Do, pad[1]:pad[1] + Ho, pad[2]:pad[2] + Wo] = torch.randn(1, cout, Do, Ho, Wo, device="cuda")
dw = torch.zeros(cout, cin, *k, device="cuda")
fn = lambda: ctx.conv3d_wgrad_pad(x, dyp, dw, accumulate=True)
gf = 2.0 * cout * cin * kd * kh * kw * Do * Ho * Wo / 1e9
sig = (cout, cin) + k + (Do, Ho, Wo) + (x.stride(3), dyp.stride(3))
shipped = autotune.known(ctx, 'wgrad', sig)
res = []
for c in [shipped] + autotune.position_split_wgrad_candidates(cout, cin, k, (Do, Ho, Wo)) + extra:
    ctx.set_tiling("wgrad", c)
    try:
        t = min(autotune._time(ctx, fn, iters=10) for _ in range(3))
    except backend.E2Error as e:
        print("   %-18s refused: %s" % (c, str(e)[:90]))
        continue
    res.append((t * 1e3, c))
ctx.set_tiling("wgrad", None)
print("wgrad %s: %.2f GF, ideal %.1f us (accumulate: no fill); shipped tiling %s" % (sys.argv[1:9], gf, gf / 157.3 * 1e3, shipped))
for t, c in sorted(res):
    print("   %-18s %8.1f us  %5.1f%% of peak" % (c, t, gf / t * 1e3 / 157.3 * 100))
