import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import neuromancer as nm, nets
nm.model_manager.reset()
np.random.seed(5)
rng = np.random.RandomState(6)
model = nets.unet3d_lite()
x = rng.rand(1, 1, 22, 140, 140).astype(np.float32)
t = rng.randint(0, 2, (1, 1, 10, 52, 52)).astype(np.float32)
P0 = None
good = {}
for it in range(7):
    g = model.gradients(x, t)
    L = float(model.loss(x, t))
    plan = model.loss_node.func if hasattr(model.loss_node, 'func') else None
    lp = None
    for cand in (getattr(model.loss_node, '_output_func', None),):
        if cand is not None:
            lp = getattr(cand, 'func', cand)
    P = model.P.detach().cpu().numpy().copy()
    if P0 is None: P0 = P
    print(it, "loss %.5f" % L, "P changed:", int((P != P0).sum()))
    if lp is None:
        print("no plan handle", [k for k in vars(model.loss_node).keys() if 'func' in k]); break
    stats = {}
    for n in lp.nodes:
        b = lp.out.get(n)
        if b is not None:
            a = b.detach().float()
            stats[n.name] = (float(a.abs().max()), bool(torch.isnan(a).any()))
    if abs(L - 0.71144) < 1e-3:
        good = stats
    else:
        for n in lp.nodes:
            if n.name in stats and n.name in good:
                m0, m1 = good[n.name][0], stats[n.name][0]
                flag = "  <-- BAD" if (stats[n.name][1] or m1 > 10 * m0 + 1) else ""
                print("   %-10s max|out| good %.3e now %.3e nan=%s%s" % (n.name, m0, m1, stats[n.name][1], flag))
        break
