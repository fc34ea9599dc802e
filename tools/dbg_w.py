import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from elektronn2_amd import backend
from oracle import e2_oracle as O
ctx = backend.Context(0)
cin, cout, k, sp = 150, 200, (1,3,3), (2,11,12)
rng = np.random.RandomState(0)
x = rng.rand(1, cin, *sp).astype(np.float32)
y_shape = (1, cout, 2, 9, 10)
dy = rng.randn(*y_shape).astype(np.float32)
dw_ref = O.conv3d_wgrad(dy, x, (cout, cin) + k)
xd = torch.tensor(x).cuda(); dyd = torch.tensor(dy).cuda()
for cfg in ["1,1,1,64,1", "1,1,1,64,2", "7,2,1,64,4", "1,1,1,64,1"]:
    os.environ["E2_WGRAD_FORCE"] = cfg
    for rep in range(3):
        dw = torch.full((cout, cin) + k, float("nan"), device="cuda")
        ctx.conv3d_wgrad(xd, dyd, dw)
        d = np.abs(dw.cpu().numpy() - dw_ref) / np.abs(dw_ref).max()
        badmask = ~(d < 1e-4)
        print(cfg, rep, "bad elems", badmask.sum(), "of", d.size, "max", np.nanmax(d) if not np.isnan(d).all() else "nan")
        if badmask.sum():
            idx = np.argwhere(badmask)
            print("   co range", idx[:,0].min(), idx[:,0].max(), "ci range", idx[:,1].min(), idx[:,1].max(), "first", idx[:5].tolist())
