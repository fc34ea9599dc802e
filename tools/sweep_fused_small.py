"""neuro3d's small late layers (2,205-8,712 positions x 100-200 channels): the plan runs their
forward GEMM with split-K into slabs + a bias / activation launch that adds the slabs up
(`Conv._fused_act` refuses the fused epilogue below 160 tiles of 112 x 128).  Is a NO-split launch
of SMALL tiles (32-64 rows x 64 positions: >= 200 work-groups from the output alone) with the fused
bias + relu epilogue faster than split-K + consumer?  Both forms timed back to back per layer.
usage: python tools/sweep_fused_small.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from elektronn2_amd import backend, autotune

LAYERS = [  # name, cin, cout, k, input spatial
    ("conv2", 30, 40, (1, 5, 5), (23, 43, 43)),
    ("conv4", 80, 100, (3, 4, 4), (10, 36, 36)),
    ("conv5", 100, 100, (3, 4, 4), (8, 33, 33)),
    ("conv6", 100, 150, (2, 4, 4), (6, 30, 30)),
    ("conv7", 150, 200, (1, 4, 4), (5, 27, 27)),
    ("conv8", 200, 200, (1, 4, 4), (5, 24, 24)),
]
ctx = backend.Context(0)
for name, cin, cout, k, sp in LAYERS:
    osp = tuple(sp[i] - k[i] + 1 for i in range(3))
    x = torch.rand(1, cin, *sp, device="cuda")
    w = torch.randn(cout, cin, *k, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda") * 0.1
    ws = torch.empty(ctx.conv_ws_bytes(cout, cin, k) // 4 + 64, device="cuda")
    ctx.conv3d_pack(w, 0, ws)
    yp = torch.empty((8, 1, cout) + osp, device="cuda")
    out = torch.empty((1, cout) + osp, device="cuda")
    gf = 2.0 * cout * cin * np.prod(k) * np.prod(osp) / 1e9
    got = [1]

    def split_form():
        got[0] = ctx.conv3d_fwd_packed_parts(x, ws, cout, k, yp)
        if got[0] > 1:
            ctx.pool_bias_act_fwd_parts(yp, got[0], b, (1, 1, 1), 'relu', out)
        else:
            ctx.pool_bias_act_fwd(yp[0], b, (1, 1, 1), 'relu', out)

    def fused_form():
        ctx.conv3d_fwd_packed_act(x, ws, cout, k, b, 'relu', out)

    res = {"split": [], "fused": []}
    for c in autotune.igemm_candidates(cout, cin, k, osp):
        if c.count(",") != 3:
            continue
        ctx.set_tiling("igemm", c)
        try:
            res["split"].append((autotune._time(ctx, split_form, iters=8) * 1e3, c))
        except backend.E2Error:
            pass
        if c.endswith(",1"):
            try:
                res["fused"].append((autotune._time(ctx, fused_form, iters=8) * 1e3, c))
            except backend.E2Error:
                pass
    # small tiles the shipped candidate list does not hold (it keeps the 4 heights with the
    # least padding): every height, no split
    mblocks = -(-cout // 16)
    for mt in (1, 2, 3, 4, 5, 6, 7):
        if mt > mblocks:
            continue
        for nt in (1, 2):
            for cc in (16, 32, 48, 64):
                c = "%d,%d,%d,1" % (mt, nt, cc)
                if any(c == cc_ for _, cc_ in res["fused"]):
                    continue
                ctx.set_tiling("igemm", c)
                try:
                    res["fused"].append((autotune._time(ctx, fused_form, iters=8) * 1e3, c))
                except backend.E2Error as e:
                    if cc == 32:
                        print("   [%s: %s]" % (c, str(e)[:110]))
    ctx.set_tiling("igemm", None)
    print("%s %d->%d %s on %s: %.2f GF, ideal %.1f us" % (name, cin, cout, k, osp, gf, gf / 157.3 * 1e3))
    for form in ("split", "fused"):
        r = sorted(res[form])[:6]
        print("   %-6s " % form + "   ".join("%s %.1f" % (c, t) for t, c in r))
