"""time the 1x1x1 conv forward on one layer for a list of forced tilings:
python tools/bench_pw.py cin cout D H W tiling [tiling ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from elektronn2_amd import backend, autotune

cin, cout, D, H, W = map(int, sys.argv[1:6])
ctx = backend.Context(0)
k = (1, 1, 1)
x = torch.rand(1, cin, D, H, W, device="cuda")
w = torch.randn(cout, cin, 1, 1, 1, device="cuda") * 0.05
b = torch.randn(cout, device="cuda") * 0.1
y = torch.empty(1, cout, D, H, W, device="cuda")
ws = torch.empty(ctx.conv_ws_bytes(cout, cin, k) // 4 + 64, device="cuda")
ctx.conv3d_pack(w, 0, ws)
gf = 2.0 * cin * cout * D * H * W / 1e9
print("1x1x1 %d -> %d on %dx%dx%d: %.2f GF, ideal %.1f us" % (cin, cout, D, H, W, gf, gf / 157.3 * 1e3))
for t in sys.argv[6:]:
    ctx.set_tiling("igemm", t)
    try:
        a = autotune._time(ctx, lambda: ctx.conv3d_fwd_packed(x, ws, cout, k, y), iters=20)
        c = autotune._time(ctx, lambda: ctx.conv3d_fwd_packed_act(x, ws, cout, k, b, 'relu', y), iters=20)
        print("   %-12s plain %7.1f us   bias+relu %7.1f us" % (t, a * 1e3, c * 1e3))
    except backend.E2Error as e:
        print("   %-12s refused: %s" % (t, str(e)[:80]))
ctx.set_tiling("igemm", None)
