"""CPU: the oracle against (a) the one known-answer relation the reference
states, (b) independent implementations (torch-CPU autograd, scipy), (c) the
committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import e2_oracle as O
from oracle import torch_step as TS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_flip_convention_matches_np_convolve():
    """reference tests/test_conv.py:89-104: conv(x, w) == np.convolve(x, w, 'valid')."""
    rng = np.random.RandomState(0)
    x, w = rng.rand(300), rng.rand(11)
    y = O.conv3d_fwd(x.reshape(1, 1, 1, 1, -1), w.reshape(1, 1, 1, 1, -1))[0, 0, 0, 0]
    assert np.abs(y - np.convolve(x, w, 'valid')).max() < 1e-12
    from scipy import signal
    x3, w3 = rng.rand(5, 6, 7), rng.rand(2, 3, 2)
    y3 = O.conv3d_fwd(x3[None, None], w3[None, None])[0, 0]
    assert np.abs(y3 - signal.convolve(x3, w3, mode='valid')).max() < 1e-12


def test_conv_node_against_torch_autograd():
    rng = np.random.RandomState(1)
    x = rng.rand(2, 3, 6, 9, 10); w = rng.randn(4, 3, 2, 3, 4); b = rng.randn(4)
    for pool in [(1, 2, 1), (2, 1, 1), (1, 1, 1)]:
        xt = torch.tensor(x, requires_grad=True); wt = torch.tensor(w, requires_grad=True)
        bt = torch.tensor(b, requires_grad=True)
        yt = TS.conv_node(xt, wt, bt, pool, 'relu')
        yo, cache = O.conv_node_fwd(x, w, b, pool, 'relu')
        assert np.abs(yt.detach().numpy() - yo).max() < 1e-12
        dout = rng.randn(*yo.shape)
        yt.backward(torch.tensor(dout))
        dx, dw, db = O.conv_node_bwd(dout, x, w, b, cache, pool, 'relu')
        assert np.abs(xt.grad.numpy() - dx).max() < 1e-11
        assert np.abs(wt.grad.numpy() - dw).max() < 1e-11
        assert np.abs(bt.grad.numpy() - db).max() < 1e-11


def test_pool_tie_rule_every_max_gets_gradient():
    y = np.zeros((1, 1, 2, 2, 2)); y[0, 0, 0, 0, 0] = y[0, 0, 1, 1, 1] = 3.0
    dy = O.maxpool3d_bwd(np.full((1, 1, 1, 1, 1), 2.0), y, (2, 2, 2))
    assert dy[0, 0, 0, 0, 0] == 2.0 and dy[0, 0, 1, 1, 1] == 2.0 and dy.sum() == 4.0
    # relu'(0) = 0.5 (Theano grad of 0.5*(x+|x|))
    d, db = O.bias_act_bwd(np.ones((1, 1, 1, 1, 2)), np.array([[[[[0.0, -1.0]]]]]), [0.0], 'relu')
    assert d[0, 0, 0, 0, 0] == 0.5 and d[0, 0, 0, 0, 1] == 0.0


def test_upconv_closed_form_equals_literal_cpu_path():
    rng = np.random.RandomState(2)
    for pool in [(2, 2, 2), (1, 2, 2), (3, 1, 2)]:
        x = rng.rand(1, 3, 3, 4, 5); w = rng.randn(6, 3, *pool)
        a, l = O.upconv3d_fwd(x, w, pool), O.upconv3d_fwd_literal(x, w, pool)
        assert a.shape == l.shape and np.abs(a - l).max() < 1e-12
        t = torch.nn.functional.conv_transpose3d(torch.tensor(x), torch.tensor(w).permute(1, 0, 2, 3, 4),
                                                 stride=pool)
        assert np.abs(t.numpy() - a).max() < 1e-12


def test_loss_matches_elementwise_reference_formula():
    rng = np.random.RandomState(3)
    lg = rng.randn(2, 3, 3, 4, 5); tg = rng.randint(-1, 3, (2, 1, 3, 4, 5)).astype(np.float32)
    loss, dl, p = O.nll_loss_and_grad(lg, tg)
    assert abs(loss - O.aggregate_loss(O.multinoulli_nll(O.softmax(lg), tg))) < 1e-12
    lt = torch.tensor(lg, requires_grad=True)
    L, _ = TS.nll_loss(lt, torch.tensor(tg, dtype=torch.float64))
    L.backward()
    assert abs(float(L.detach()) - loss) < 1e-12 and np.abs(lt.grad.numpy() - dl).max() < 1e-12


def test_shape_rules_f9():
    assert O.net_out_shape(O.NEURO3D_LITE, (23, 183, 183)) == (10, 37, 37)
    assert O.net_out_shape(O.NEURO3D, (23, 185, 185)) == (5, 21, 21)
    with pytest.raises(ValueError):
        O.net_out_shape(O.NEURO3D, (23, 183, 183))


def test_ops_golden():
    g = np.load(os.path.join(GOLD, "ops.npz"))
    assert np.abs(O.conv3d_fwd(g['conv_x'], g['conv_w']) - g['conv_y']).max() < 1e-12
    assert np.abs(O.conv3d_dgrad(g['conv_dy'], g['conv_w'], g['conv_x'].shape) - g['conv_dx']).max() < 1e-12
    assert np.abs(O.conv3d_wgrad(g['conv_dy'], g['conv_x'], g['conv_w'].shape) - g['conv_dw']).max() < 1e-12
    p = O.maxpool3d_fwd(g['pool_y'], (2, 2, 2))
    assert np.abs(O.bias_act_fwd(p, g['pool_b'], 'relu') - g['pool_out']).max() < 1e-12
    dp, db = O.bias_act_bwd(g['pool_dout'], p, g['pool_b'], 'relu')
    assert np.abs(O.maxpool3d_bwd(dp, g['pool_y'], (2, 2, 2)) - g['pool_dy']).max() < 1e-12
    assert np.abs(db - g['pool_db']).max() < 1e-12
    assert np.abs(O.upconv3d_fwd(g['up_x'], g['up_w'], (2, 1, 2)) - g['up_y']).max() < 1e-12
    loss, dl, pr = O.nll_loss_and_grad(g['nll_logits'], g['nll_target'])
    assert abs(loss - float(g['nll_loss'])) < 1e-12 and np.abs(dl - g['nll_dlogits']).max() < 1e-12
    pp, m, s = g['adam_p0'].copy(), np.zeros(50), np.zeros(50)
    for t in range(1, 4):
        pp, m, s = O.adam_step(pp, g['adam_g'][t - 1], m, s, t, 5e-4, 0.9, 0.999, 0.5e-4, True)
    assert np.abs(pp - g['adam_p3']).max() < 1e-14


def test_step_golden_lite_and_fp32_port():
    """oracle reproduces the committed step fixture; the fp32 torch port (the
    cpu_baseline implementation) agrees with it to fp32 accuracy."""
    g = np.load(os.path.join(GOLD, "step_lite.npz"))
    spec, sp = O.NEURO3D_LITE, tuple(int(v) for v in g['in_spatial'])
    params = O.init_net(spec, 1, seed=1)
    rng = np.random.RandomState(0)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    t.flat[::17] = -1
    loss, grads, probs = O.net_loss_and_grads(spec, params, x, t)
    assert abs(loss - float(g['loss'])) < 1e-12
    assert np.abs(grads[0][0] - g['gw_first']).max() < 1e-12
    assert np.abs(grads[-1][0] - g['gw_last']).max() < 1e-12
    net = TS.TorchNet(spec, params)
    losses = [net.trainingstep(torch.tensor(x), torch.tensor(t)) for _ in range(3)]
    assert np.abs(np.array(losses) - g['losses3']).max() / abs(g['losses3']).max() < 1e-5
    w_last = net.w[-1].detach().numpy()
    assert np.abs(w_last - g['w3_last']).max() / np.abs(g['w3_last']).max() < 5e-4


DENSE_SPEC = [(4, (1, 3, 3), (1, 2, 2), 'relu'), (6, (3, 3, 3), (2, 1, 1), 'relu'),
              (2, (1, 1, 1), (1, 1, 1), 'lin')]


def test_predict_dense_tiling_equals_per_voxel_evaluation():
    """oracle self-check for the dense-prediction row (node_basic.py:805-1012): the
    tiled, stride-offset-interleaved prediction equals the net evaluated on the
    field-of-view patch of every voxel (independent of tiling, strides, end blocks)."""
    spec = DENSE_SPEC
    params = O.init_net(spec, 1, seed=3)
    strides, fov, offset = O.net_geometry(spec)
    assert tuple(strides) == (2, 2, 2) and tuple(fov) == (4, 8, 8)
    rng = np.random.RandomState(0)
    raw = rng.rand(1, 11, 23, 25).astype(np.float32)
    dense = O.predict_dense(spec, params, raw, (6, 16, 16))
    assert dense.shape == (2, 11 - 2 * offset[0], 23 - 2 * offset[1], 25 - 2 * offset[2])
    for z, x, y in [(0, 0, 0), (1, 2, 3), (6, 14, 16), (3, 14, 0), (5, 7, 9), (6, 0, 16)]:
        ref = O.predict_voxel(spec, params, raw, (z, x, y))
        assert np.abs(dense[:, z, x, y] - ref).max() < 1e-6, (z, x, y)
    # integer input is scaled by 1/255; uint8 output is floor(255 p); mirror padding
    raw8 = (raw * 255).astype(np.uint8)
    d8 = O.predict_dense(spec, params, raw8, (6, 16, 16), as_uint8=True)
    assert d8.dtype == np.uint8 and np.abs(d8.astype(np.float32) - 255 * dense).max() < 3.0
    dp = O.predict_dense(spec, params, raw, (6, 16, 16), pad_raw=True)
    assert dp.shape == (2, 11, 23, 25)
    assert np.abs(dp[:, offset[0]:-offset[0], offset[1]:-offset[1], offset[2]:-offset[2]]
                  - dense).max() < 1e-6
