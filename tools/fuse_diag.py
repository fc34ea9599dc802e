"""Why does plan option fuse_actbwd give other gradients than the default plan under SOME tuned
tilings?  (DESIGN finding 56's open point.)  usage: E2HIP_TUNE_CACHE=<cache of a failing run>
python tools/fuse_diag.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import e2_oracle as O          # (a diagnostic tool, not the product)
from elektronn2_amd import autotune
from elektronn2_amd.neuromancer import plan_options
from test_model_gpu import build, rel

spec, sp = O.NEURO3D, (17, 109, 109)
params = O.init_net(spec, 1, seed=1)
SEED = int(sys.argv[sys.argv.index('--seed') + 1]) if '--seed' in sys.argv else 3
rng = np.random.RandomState(SEED)
x = rng.rand(1, 1, *sp).astype(np.float32)
t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
loss_ref, grads_ref, _ = O.net_loss_and_grads(spec, params, x, t)
flat_ref = []
for gw, gb in grads_ref:
    flat_ref += [gw, gb]
res = {}
for fuse in (0, 1):
    autotune.launch_log = []
    with plan_options(fuse_actbwd=fuse):
        m = build('full', sp, params)
        if "--sync" in sys.argv or "--nosync" in sys.argv:
            # the first evaluation in three steps: build the plan, (wait for the device), run
            import torch
            f = m._grad_func
            f.compile()
            f.func.set_inputs([x, t])
            if "--sync" in sys.argv:
                torch.cuda.synchronize()
            f.func.run()
            g = f.func.fetch()
        else:
            g = m.gradients(x, t)
        if "--tensors" in sys.argv and fuse:
            # the plan's tensors after the FIRST evaluation against the same after the second
            import torch
            f = m._grad_func
            plan = f.func
            torch.cuda.synchronize()
            snap = {}
            for n in plan.nodes:
                for kind, d in (("out", plan.out), ("grad", plan.grad)):
                    tns = d.get(n)
                    if tns is not None:
                        snap[n.name, kind] = tns.clone()
                for key in ("y", "dy", "dy_pad"):
                    tns = plan.scratch.get((n, key))
                    if tns is not None:
                        snap[n.name, key] = tns.clone()
            g2 = m.gradients(x, t)
            torch.cuda.synchronize()
            for n in plan.nodes:
                for kind in ("out", "y", "grad", "dy", "dy_pad"):
                    if (n.name, kind) in snap:
                        cur = plan.out.get(n) if kind == "out" else plan.grad.get(n) if kind == "grad" else plan.scratch.get((n, kind))
                        a, b = snap[n.name, kind], cur
                        d = float((a - b).abs().max()); mx = float(b.abs().max())
                        if d > 1e-5 * max(mx, 1e-30):
                            bad = (a - b).abs() > 1e-5 * max(mx, 1e-30)
                            idx = bad.nonzero()
                            print("   DIFFERS %-8s %-7s max |d| %.3e of %.3e, %d elements, first %s last %s" % (
                                n.name, kind, d, mx, int(bad.sum()), idx[0].tolist(), idx[-1].tolist()))
        if "--twice" in sys.argv:          # a second evaluation of the same plan
            g2 = m.gradients(x, t)
            print("   second evaluation against the first, conv3_w:", rel(g2[6], g[6]), " conv_w:", rel(g2[0], g[0]))
            g = g2
        if "--steps" in sys.argv:          # (the test's flow: three Adam steps of this model before the next is built)
            print("   losses", [float(m.trainingstep(x, t, optimiser='Adam')[0]) for _ in range(3)])
            print("   params object shared with the caller's dict?", any(
                np.shares_memory(p.get_value(borrow=True) if hasattr(p, 'get_value') else 0, v)
                for p in [] for v in []))
    res[fuse] = g
    # did the model write into the caller's parameter arrays?
    chk = O.init_net(spec, 1, seed=1)
    same = all(np.array_equal(a, b) for la, lb in zip(params, chk) for a, b in zip(la, lb))
    print("   caller's initial parameters unchanged after this model:", same)
    names = list(m.loss_node.all_trainable_params.keys())
    print("fuse_actbwd=%d: gradient against the f64 oracle, per tensor" % fuse)
    for n, a, r in zip(names, g, flat_ref):
        print("   %-10s %.2e" % (n, rel(a, r)))
    if fuse:
        for key, tiling, ll in autotune.launch_log:
            if key.startswith("igemm|1,") or key.startswith("igemm|2,") or key.startswith("igemm|0,"):
                print("   launch %-44s tiling %-22s ran %s" % (key, tiling, ll))
for n, a, b in zip(names, res[0], res[1]):
    print("0 vs 1  %-10s %.2e" % (n, rel(a, b)))
