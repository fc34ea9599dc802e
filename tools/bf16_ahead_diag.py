"""why did `test_step_with_operands_made_ahead_equals_the_converting_step
[False-neuro3d-sp1]` differ in its 4th loss by 8.75e-5 on the driver's box?

ONE diagnostic run, three questions:
  1. A/A: the converting bf16 step against ITSELF (two fresh models, same seeds), per step --
     the spread the weight gradients' atomic summation order alone produces;
  2. ahead vs converting, per step and -- after ONE gradient evaluation -- per tensor (forward
     outputs, output gradients: bit compare; parameter gradients: relative distance), with the
     first diverging tensor named;
  3. both forms' step-1 loss and gradients against oracle/e2_oracle.py fed with bf16-rounded GEMM
     operands (rnd=), so that "which of the two is wrong" has an answer.
usage: python tools/bf16_ahead_diag.py [neuro3d|neuro3d_lite] [graph]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import e2_oracle as O


def bf16_round(a):
    return torch.tensor(np.asarray(a, np.float32)).bfloat16().double().numpy()


def build(net, sp, ahead):
    from elektronn2_amd import nets, neuromancer as nm
    nm.model_manager.reset()
    nm.set_plan_options(bf16_ahead=ahead)
    spec = O.NEURO3D_LITE if net == "neuro3d_lite" else O.NEURO3D
    params = O.init_net(spec, 1, seed=3)
    rng = np.random.RandomState(5)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    m = getattr(nets, net)((None, 1) + sp, params=params)
    m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    return m, spec, params, x, t


def steps(net, sp, ahead, use_graph, n):
    from elektronn2_amd import autotune
    m, spec, params, x, t = build(net, sp, ahead)
    autotune.force('igemm', "32,1,2")
    autotune.force('wgrad', "32,1,2,0,0")
    try:
        opt = m.optimisers['Adam']
        opt.step.compile()
        opt.step.func.use_graph = use_graph
        losses, ps = [], []
        for _ in range(n):
            losses.append(float(m.trainingstep(x, t, optimiser='Adam')[0]))
            ps.append(opt.step.func.model.P.detach().cpu().numpy().copy())
        kinds = sorted(set(k for (_, k) in opt.step.func.bf16a))
    finally:
        autotune.force('igemm', None)
        autotune.force('wgrad', None)
    return np.array(losses), ps, kinds


def grads_once(net, sp, ahead):
    """one gradient evaluation (eager): loss, every node's output / output gradient, parameter
    gradients by name"""
    from elektronn2_amd import autotune
    m, spec, params, x, t = build(net, sp, ahead)
    autotune.force('igemm', "32,1,2")
    autotune.force('wgrad', "32,1,2,0,0")
    try:
        g = m.gradients(x, t)
        plan = m._grad_func.func
        torch.cuda.synchronize()
        outs = {n.name: plan.out[n].detach().cpu().numpy().copy() for n in plan.nodes
                if plan.out.get(n) is not None}
        douts = {n.name: plan.grad[n].detach().cpu().numpy().copy() for n in plan.nodes
                 if plan.grad.get(n) is not None}
        names = list(m.loss_node.all_trainable_params.keys())
        kinds = sorted(set(k for (_, k) in plan.bf16a))
    finally:
        autotune.force('igemm', None)
        autotune.force('wgrad', None)
    return dict(outs=outs, douts=douts, g=dict(zip(names, g)), kinds=kinds,
                spec=spec, params=params, x=x, t=t)


def sensitivity(net, sp, n_steps=3, trials=24):
    """does parameter noise of the size two runs of ONE form differ by (A/A above: 7e-9 ... 4e-8 of
    max |P| after three steps) reach the 4th loss?  Three steps of the converting form, then the
    loss of the SAME batch with P perturbed by relative Gaussian noise of that size"""
    from elektronn2_amd import autotune
    m, spec, params, x, t = build(net, sp, False)
    autotune.force('igemm', "32,1,2")
    autotune.force('wgrad', "32,1,2,0,0")
    try:
        for _ in range(n_steps):
            m.trainingstep(x, t, optimiser='Adam')
        P = m.optimisers['Adam'].step.func.model.P
        P0 = P.clone()
        base = float(m.loss(x, t))
        rng = torch.Generator(device='cpu').manual_seed(1)
        print("\nsensitivity of the loss after %d steps (%.7f) to parameter noise:" % (n_steps, base))
        for eps in (1e-8, 4e-8, 1e-7, 1e-6):
            d = []
            for _ in range(trials):
                noise = torch.randn(P0.numel(), generator=rng).to(P0.device)
                P.copy_(P0 * (1.0 + eps * noise))
                torch.cuda.synchronize()
                d.append(abs(float(m.loss(x, t)) - base) / abs(base))
            d = np.sort(np.array(d))
            print("  relative noise %.0e: |dloss|/loss median %.2e, 90%% %.2e, max %.2e, share above 2e-5: %.2f"
                  % (eps, d[len(d) // 2], d[int(0.9 * len(d))], d[-1], float((d > 2e-5).mean())))
        P.copy_(P0)
    finally:
        autotune.force('igemm', None)
        autotune.force('wgrad', None)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    import elektronn2_amd
    net = sys.argv[1] if len(sys.argv) > 1 else "neuro3d"
    use_graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
    sp = (23, 121, 121) if net == "neuro3d" else (9, 71, 71)
    elektronn2_amd.set_mfma_dtype('bf16')
    N = 4
    runs = {}
    for tag, ahead in (("conv#1", False), ("conv#2", False), ("conv#3", False),
                       ("ahead#1", True), ("ahead#2", True), ("ahead#3", True)):
        runs[tag] = steps(net, sp, ahead, use_graph, N)
        print("%-8s losses %s  launches %s" % (tag, np.array2string(runs[tag][0], precision=7), runs[tag][2]),
              flush=True)
    print("\npairwise: per step |dloss| / loss and max |dP| / max |P|")
    pairs = [("conv#1", "conv#2"), ("conv#1", "conv#3"), ("conv#2", "conv#3"),
             ("ahead#1", "ahead#2"), ("ahead#1", "ahead#3"),
             ("conv#1", "ahead#1"), ("conv#2", "ahead#2"), ("conv#3", "ahead#3"), ("conv#1", "ahead#3")]
    for a, b in pairs:
        la, pa, _ = runs[a]
        lb, pb, _ = runs[b]
        print("%-8s vs %-8s  loss %s   P %s" % (
            a, b, " ".join("%.2e" % (abs(u - v) / abs(v)) for u, v in zip(la, lb)),
            " ".join("%.2e" % rel(u, v) for u, v in zip(pa, pb))))

    sensitivity(net, sp)
    print("\none gradient evaluation, per tensor (bit compare for outputs / output gradients)")
    A = grads_once(net, sp, False)
    A2 = grads_once(net, sp, False)
    B = grads_once(net, sp, True)
    print("launch kinds: converting %s, ahead %s" % (A['kinds'], B['kinds']))
    for what in ("outs", "douts"):
        for name in A[what]:
            if name not in B[what]:
                print("  %-6s %-10s only in the converting form" % (what, name))
                continue
            a, a2, b = A[what][name], A2[what][name], B[what][name]
            nb = int((a.view(np.int32) != b.view(np.int32)).sum())
            na = int((a.view(np.int32) != a2.view(np.int32)).sum())
            print("  %-6s %-10s differing words: conv/conv %d, ahead/conv %d of %d  (rel %.2e)"
                  % (what, name, na, nb, a.size, rel(b, a)))
    # the oracle with bf16-rounded operands: every layer but the fused first one and the head
    spec, params, x, t = A['spec'], A['params'], A['x'], A['t']
    layers = set(range(1, len(spec) - 1))
    loss_ref, grads_ref, _ = O.net_loss_and_grads(spec, params, x, t, rnd=bf16_round, rnd_layers=layers)
    ref = {}
    names = list(A['g'].keys())
    flat = []
    for gw, gb in grads_ref:
        flat += [gw, gb]
    # parameter order of the model = node order, w before b (checked by the shapes)
    for nme in names:
        g = A['g'][nme]
        cands = [r for r in flat if r.shape == g.shape]
        ref[nme] = min(cands, key=lambda r: np.abs(g - r).max())
    print("\nparameter gradients: conv/conv, ahead/conv, conv/oracle(bf16 operands), ahead/oracle")
    for nme in names:
        print("  %-12s %.2e  %.2e  %.2e  %.2e" % (nme, rel(A2['g'][nme], A['g'][nme]),
                                                 rel(B['g'][nme], A['g'][nme]),
                                                 rel(A['g'][nme], ref[nme]), rel(B['g'][nme], ref[nme])))
    print("oracle loss (bf16 operands) %.7f" % loss_ref)


if __name__ == "__main__":
    main()
