import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import backend
ctx = backend.Context(0)
torch.manual_seed(0)
cin, cout, pool = 128, 128, (1, 2, 2)
x = torch.rand(1, cin, 14, 28, 28, device="cuda")
w = torch.randn(cout, cin, *pool, device="cuda") * 0.05
b = torch.randn(cout, device="cuda") * 0.1
y = torch.empty(1, cout, 14, 56, 56, device="cuda")
nb = ctx.upconv_ws_bytes(cout, cin, pool, x.shape)
ws = torch.empty(nb // 4 + 64, device="cuda")
ctx.upconv3d_fwd(x, w, b, pool, 'relu', y, ws)
dout = torch.randn_like(y)
ref = None
for it in range(8):
    dx = torch.full_like(x, float('nan')); dw = torch.zeros_like(w); db = torch.zeros_like(b)
    os.environ['E2_VERBOSE'] = '1' if it == 0 else '0'
    if it > 0: os.environ.pop('E2_VERBOSE')
    ctx.upconv3d_bwd(x, w, y, dout, pool, 'relu', dx, dw, db, ws)
    torch.cuda.synchronize()
    cur = (dx.clone(), dw.clone(), db.clone())
    if ref is None:
        ref = cur
        print("nan in dx:", bool(torch.isnan(dx).any()))
        continue
    print(it, [float((a - r).abs().max() / r.abs().max()) for a, r in zip(cur, ref)])
