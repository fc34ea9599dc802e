"""Generates the golden fixtures under tests/golden/ from the CPU oracle
(oracle/e2_oracle.py, float64).  The reference itself cannot be imported here
(Theano absent -- see oracle header: "parity unpinned"), so these vectors pin
the BUILD's oracle, cross-checked against torch-CPU autograd at generation time.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import e2_oracle as O            # noqa: E402
from oracle import torch_step as TS          # noqa: E402


def ops_fixture():
    rng = np.random.RandomState(1234)
    d = {}
    # conv: anisotropic taps, Cin not a multiple of 4, batch 2
    x = rng.rand(2, 3, 4, 7, 9).astype(np.float32)
    w = (rng.randn(5, 3, 2, 3, 2) / 4).astype(np.float32)
    b = rng.randn(5).astype(np.float32)
    y = O.conv3d_fwd(x, w)
    dy = rng.randn(*y.shape).astype(np.float32)
    d.update(conv_x=x, conv_w=w, conv_y=y, conv_dy=dy,
             conv_dx=O.conv3d_dgrad(dy, w, x.shape), conv_dw=O.conv3d_wgrad(dy, x, w.shape))
    # second opinion
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    yt = torch.nn.functional.conv3d(xt, wt.flip(2, 3, 4))
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    assert np.abs(yt.detach().numpy() - y).max() < 1e-12
    assert np.abs(xt.grad.numpy() - d['conv_dx']).max() < 1e-12
    assert np.abs(wt.grad.numpy() - d['conv_dw']).max() < 1e-12
    # conv node with pool (ties + exact zero pre-activation)
    yy = np.round(rng.randn(1, 2, 4, 6, 8) * 2) / 2
    bb = np.array([0.5, -0.25], np.float32)
    p = O.maxpool3d_fwd(yy, (2, 2, 2))
    out = O.bias_act_fwd(p, bb, 'relu')
    dout = rng.randn(*out.shape).astype(np.float32)
    dp, db = O.bias_act_bwd(dout, p, bb, 'relu')
    d.update(pool_y=yy.astype(np.float32), pool_b=bb, pool_out=out, pool_dout=dout,
             pool_dy=O.maxpool3d_bwd(dp, yy, (2, 2, 2)), pool_db=db)
    # upconv
    ux = rng.randn(1, 4, 2, 3, 3).astype(np.float32)
    uw = rng.randn(3, 4, 2, 1, 2).astype(np.float32)
    uy = O.upconv3d_fwd(ux, uw, (2, 1, 2))
    assert np.abs(uy - O.upconv3d_fwd_literal(ux, uw, (2, 1, 2))).max() < 1e-12
    udy = rng.randn(*uy.shape).astype(np.float32)
    d.update(up_x=ux, up_w=uw, up_y=uy, up_dy=udy, up_dx=O.upconv3d_dgrad(udy, uw, (2, 1, 2)),
             up_dw=O.upconv3d_wgrad(udy, ux, (2, 1, 2)))
    # loss
    lg = (rng.randn(1, 3, 2, 3, 4) * 2).astype(np.float32)
    tg = rng.randint(0, 3, (1, 1, 2, 3, 4)).astype(np.float32)
    tg[0, 0, 0, 0, 0] = -1
    loss, dl, pr = O.nll_loss_and_grad(lg, tg)
    d.update(nll_logits=lg, nll_target=tg, nll_loss=np.array(loss), nll_dlogits=dl, nll_probs=pr)
    # adam, 3 steps
    p0 = rng.randn(50); g = rng.randn(3, 50)
    pp, m, s = p0.copy(), np.zeros(50), np.zeros(50)
    for t in range(1, 4):
        pp, m, s = O.adam_step(pp, g[t - 1], m, s, t, 5e-4, 0.9, 0.999, 0.5e-4, True)
    d.update(adam_p0=p0, adam_g=g, adam_p3=pp, adam_m3=m, adam_s3=s)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **d)


def step_fixture(name, spec, sp):
    params = O.init_net(spec, 1, seed=1)
    rng = np.random.RandomState(0)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    osp = O.net_out_shape(spec, sp)
    t = rng.randint(0, 2, (1, 1) + osp).astype(np.float32)
    t.flat[::17] = -1
    loss, grads, probs = O.net_loss_and_grads(spec, params, x, t)
    # torch-CPU fp32 second opinion on loss + gradients
    net = TS.TorchNet(spec, params, dtype=torch.float64)
    l2, _ = net.loss_and_grads(torch.tensor(x, dtype=torch.float64), torch.tensor(t, dtype=torch.float64))
    assert abs(float(l2) - loss) < 1e-10
    for i in range(len(spec)):
        assert np.abs(net.w[i].grad.numpy() - grads[i][0]).max() < 1e-10
    losses, P = O.net_train_steps(spec, params, x, t, 3)
    d = dict(in_spatial=np.array(sp), loss=np.array(loss), losses3=np.array(losses),
             probs=probs.astype(np.float32))
    for i in range(len(spec)):
        d["gw_l2_%d" % i] = np.array(np.sqrt((grads[i][0] ** 2).sum()))
        d["gw_sum_%d" % i] = np.array(grads[i][0].sum())
        d["gb_%d" % i] = grads[i][1]
        d["b3_%d" % i] = P[i][1]
        d["w3_l2_%d" % i] = np.array(np.sqrt((P[i][0] ** 2).sum()))
    d["gw_first"] = grads[0][0]
    d["gw_last"] = grads[-1][0]
    d["w3_first"] = P[0][0]
    d["w3_last"] = P[-1][0]
    np.savez_compressed(os.path.join(HERE, "step_%s.npz" % name), **d)


if __name__ == "__main__":
    ops_fixture()
    step_fixture("lite", O.NEURO3D_LITE, (7, 47, 47))
    step_fixture("full", O.NEURO3D, (17, 109, 109))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
