"""BASELINE config 1 (examples/mnist.py, the reference's own CPU-runnable case) through the
HIP path: 2-D convs (unit z axis) with train-mode batch normalisation, Perceptrons, softmax /
NLL on 'b,f' tensors -- loss, probabilities, every parameter gradient (through the batch
statistics), Adam steps with gamma's 3x weight decay and the running-statistics updates,
against oracle/mnist_oracle.py (torch float64 autograd of the restated forward pass)."""
import numpy as np
import pytest
import torch

from oracle import e2_oracle as O
from oracle import mnist_oracle as MO

pytestmark = pytest.mark.gpu
TOL = 1e-4
HYP = dict(lr=2e-4, mom=0.9, beta2=0.99, wd=0.5e-3)        # examples/mnist.py:18-23


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def all_params(m):
    """name -> value of every parameter (trainable and running statistics)"""
    out = {}
    for node in m.nodes.values():
        for k, p in node.params.items():
            if k in ('w', 'b', 'gamma', 'mean', 'std'):
                out['%s_%s' % (node.name, k)] = p.get_value()
    return out


def dev(a):
    return torch.tensor(np.ascontiguousarray(a, np.float32), device="cuda")


def test_dense_and_batchnorm_ops(ctx):
    rng = np.random.RandomState(5)
    x = rng.randn(20, 57).astype(np.float32)
    w = rng.randn(57, 33).astype(np.float32)
    dy = rng.randn(20, 33).astype(np.float32)
    y = torch.empty(20, 33, device="cuda")
    ctx.dense_fwd(dev(x), dev(w), y)
    assert rel(y.cpu().numpy(), x.astype(np.float64) @ w) < 2e-6
    dx = torch.full((20, 57), float("nan"), device="cuda")
    ctx.dense_dgrad(dev(dy), dev(w), dx)
    assert rel(dx.cpu().numpy(), dy.astype(np.float64) @ w.T) < 2e-6
    ctx.dense_dgrad(dev(dy), dev(w), dx, accumulate=True)
    assert rel(dx.cpu().numpy(), 2 * (dy.astype(np.float64) @ w.T)) < 2e-6
    dw = torch.zeros(57, 33, device="cuda")
    ctx.dense_wgrad(dev(x), dev(dy), dw, accumulate=True)
    ctx.dense_wgrad(dev(x), dev(dy), dw, accumulate=True)
    assert rel(dw.cpu().numpy(), 2 * (x.astype(np.float64).T @ dy)) < 2e-6
    # batch norm + relu, train mode, gradient through the statistics, strided dx view
    for shape in [(20, 12, 1, 12, 12), (3, 5, 2, 4, 7), (20, 200, 1, 1, 1)]:
        xb = rng.randn(*shape).astype(np.float32) * 2 + 0.5
        g = (rng.rand(shape[1]) + 0.5).astype(np.float32)
        b = rng.randn(shape[1]).astype(np.float32) * 0.3
        do = rng.randn(*shape).astype(np.float32)
        X = torch.tensor(xb, dtype=torch.float64, requires_grad=True)
        G = torch.tensor(g, dtype=torch.float64, requires_grad=True)
        B = torch.tensor(b, dtype=torch.float64, requires_grad=True)
        yb, mean, std = MO.batchnorm(X, G, B)
        out_ref = MO.relu(yb)
        (out_ref * torch.tensor(do, dtype=torch.float64)).sum().backward()
        rm = torch.zeros(shape[1], device="cuda"); rs = torch.ones(shape[1], device="cuda")
        out = torch.full(shape, float("nan"), device="cuda")
        save = torch.zeros(2 * shape[1], device="cuda")
        ctx.batchnorm_act_fwd(dev(xb), dev(g), dev(b), rm, rs, True, True, 'relu', out, save)
        assert rel(out.cpu().numpy(), out_ref.detach().numpy()) < 1e-5
        assert rel(save[:shape[1]].cpu().numpy(), mean.detach().numpy()) < 1e-5
        assert rel(save[shape[1]:].cpu().numpy(), std.detach().numpy()) < 1e-5
        assert rel(rm.cpu().numpy(), 0.0005 * mean.detach().numpy()) < 1e-5
        assert rel(rs.cpu().numpy(), 0.9995 + 0.0005 * std.detach().numpy()) < 1e-6
        big = torch.zeros((shape[0], shape[1], shape[2] + 2, shape[3] + 2, shape[4] + 3), device="cuda")
        dxv = big[:, :, 1:-1, 1:-1, 2:-1]
        dg = torch.zeros(shape[1], device="cuda"); db = torch.zeros(shape[1], device="cuda")
        ctx.batchnorm_act_bwd(dev(do), dev(xb), dev(g), dev(b), save, True, 'relu', dxv, dg, db)
        assert rel(dxv.cpu().numpy(), X.grad.numpy()) < 2e-5
        assert rel(dg.cpu().numpy(), G.grad.numpy()) < 2e-5
        assert rel(db.cpu().numpy(), B.grad.numpy()) < 2e-5
        border = big.clone()
        border[:, :, 1:-1, 1:-1, 2:-1] = 0
        assert not border.any()                    # nothing written outside the view
        # predict mode: stored statistics, constants of the gradient
        rm2 = dev(rng.randn(shape[1])); rs2 = dev(rng.rand(shape[1]) + 0.5)
        ctx.batchnorm_act_fwd(dev(xb), dev(g), dev(b), rm2, rs2, False, False, 'lin', out, save)
        ref = (g / rs2.cpu().numpy()).reshape(1, -1, 1, 1, 1) * xb + \
            (b - g * rm2.cpu().numpy() / rs2.cpu().numpy()).reshape(1, -1, 1, 1, 1)
        assert rel(out.cpu().numpy(), ref) < 1e-5


def test_mnist_config_parity():
    from elektronn2_amd import nets, neuromancer as nm
    nm.model_manager.reset()
    np.random.seed(11)
    m = nets.mnist()
    m.set_opt_meta_params('Adam', HYP)
    rng = np.random.RandomState(2)
    x = rng.rand(20, 1, 26, 26).astype(np.float32)
    t = rng.randint(0, 10, (20, 1)).astype(np.float32)
    P = all_params(m)
    assert set(P) == {'conv_w', 'conv_b', 'conv_gamma', 'conv_mean', 'conv_std',
                      'conv1_w', 'conv1_b', 'conv1_gamma', 'conv1_mean', 'conv1_std',
                      'conv2_w', 'conv2_b', 'conv2_gamma', 'conv2_mean', 'conv2_std',
                      'dot_w', 'dot_b', 'dot1_w', 'dot1_b'}
    loss_ref, grads_ref, probs_ref, stats = MO.loss_and_grads(P, x, t)

    assert abs(float(m.loss(x, t)) - loss_ref) / abs(loss_ref) < TOL
    pr = m.predict(x)
    assert pr.shape == (20, 10) and rel(pr, probs_ref) < TOL
    l2, err, pr2 = m.predict_ext(x, t)
    assert abs(float(l2) - loss_ref) / abs(loss_ref) < TOL
    assert abs(float(err) - MO.classification_errors(probs_ref, t)) < 1e-6
    names = list(m.loss_node.all_trainable_params.keys())
    g = m.gradients(x, t)
    assert len(g) == len(names) == 13
    for nme, gv in zip(names, g):
        assert rel(gv, grads_ref[nme]) < TOL, nme
    # loss / prediction / gradient functions leave the running statistics alone
    P1 = all_params(m)
    for k in P:
        assert np.array_equal(P[k], P1[k]), k

    # three Adam steps: gamma with three times the weight decay, running statistics
    reg = {k: p.apply_reg for k, p in m.loss_node.all_trainable_params.items()}
    assert reg['conv_gamma'] == 3.0 and reg['conv_w'] is True and reg['conv_b'] is False
    Q = {k: np.asarray(v, np.float64) for k, v in P.items()}
    M = {k: np.zeros_like(Q[k]) for k in names}
    S = {k: np.zeros_like(Q[k]) for k in names}
    for step in range(1, 4):
        xs = rng.rand(20, 1, 26, 26).astype(np.float32)
        ts = rng.randint(0, 10, (20, 1)).astype(np.float32)
        lref, gref, _, st = MO.loss_and_grads(Q, xs, ts)
        for k in names:
            Q[k], M[k], S[k] = O.adam_step(Q[k], gref[k], M[k], S[k], step, HYP['lr'],
                                           HYP['mom'], HYP['beta2'], HYP['wd'], reg[k])
        for nme, (mean, std) in zip(('conv', 'conv1', 'conv2'), st):
            Q[nme + '_mean'] = 0.9995 * Q[nme + '_mean'] + 0.0005 * mean
            Q[nme + '_std'] = 0.9995 * Q[nme + '_std'] + 0.0005 * std
        loss = float(m.trainingstep(xs, ts, optimiser='Adam')[0])
        assert abs(loss - lref) / abs(lref) < TOL, step
    P3 = all_params(m)
    for k in P3:
        tol = 1e-5 if k.endswith(('_mean', '_std')) else 5e-4
        assert rel(P3[k], Q[k]) < tol, k
    assert not np.array_equal(P3['conv_mean'], P['conv_mean'])     # the statistics moved
