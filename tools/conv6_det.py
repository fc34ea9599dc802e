import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import backend
from oracle import e2_oracle as O
ctx = backend.Context(0)
torch.manual_seed(0)
cin, cout, k, sp = 128, 256, (3, 3, 3), (22, 14, 14)
x = torch.rand(1, cin, *sp, device="cuda")
w = torch.randn(cout, cin, *k, device="cuda") * 0.03
osp = tuple(sp[i] - k[i] + 1 for i in range(3))
ws = torch.empty(ctx.conv_ws_bytes(cout, cin, k) // 4 + 64, device="cuda")
ctx.conv3d_pack(w, 0, ws)
ref = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu().flip(2, 3, 4)).float()
s = torch.cuda.Stream()
ctx.set_stream(s)
for force in ["8,1,32,4", "8,1,32,1", "8,1,32,2", "4,2,32,6"]:
    os.environ["E2_IGEMM_FORCE"] = force
    y = torch.full((1, cout) + osp, float("nan"), device="cuda")
    with torch.cuda.stream(s):
        ctx.conv3d_fwd_packed(x, ws, cout, k, y)     # eager
        s.synchronize()
        e0 = float((y.cpu() - ref).abs().max() / ref.abs().max())
        ctx.graph_begin()
        ctx.conv3d_fwd_packed(x, ws, cout, k, y)
        g = ctx.graph_end()
        errs = []
        for it in range(40):
            y.fill_(float(it))                       # garbage the graph's memset must clear
            ctx.graph_launch(g)
            s.synchronize()
            errs.append(float((y.cpu() - ref).abs().max() / ref.abs().max()))
    print(force, "eager err %.2e" % e0, "graph replays: max err %.2e, #bad %d" % (max(errs), sum(e > 1e-4 for e in errs)))
