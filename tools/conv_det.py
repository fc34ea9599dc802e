import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import backend, autotune
ctx = backend.Context(0)
torch.manual_seed(0)
cin, cout, k, sp = 128, 128, (3, 3, 3), (16, 30, 30)
x = torch.rand(1, cin, *sp, device="cuda")
w = torch.randn(cout, cin, *k, device="cuda") * 0.03
b = torch.randn(cout, device="cuda") * 0.1
osp = tuple(sp[i] - k[i] + 1 for i in range(3))
ws = torch.empty(ctx.conv_ws_bytes(cout, cin, k) // 4 + 64, device="cuda")
ctx.conv3d_pack(w, 0, ws)
cands = autotune.igemm_candidates(cout, cin, k, osp, split_k=False)
print(len(cands), "candidates")
bad = []
for c in [None] + cands:
    if c: os.environ["E2_IGEMM_FORCE"] = c
    else: os.environ.pop("E2_IGEMM_FORCE", None)
    ref = None; worst = 0.0
    try:
        for it in range(5):
            out = torch.full((1, cout) + osp, float("nan"), device="cuda")
            ctx.conv3d_fwd_packed_act(x, ws, cout, k, b, 'relu', out)
            torch.cuda.synchronize()
            if ref is None: ref = out.clone()
            else: worst = max(worst, float((out - ref).abs().max()))
    except backend.E2Error as e:
        continue
    if worst > 0: bad.append((c, worst))
print("nondeterministic configs:", bad)
