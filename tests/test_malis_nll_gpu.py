"""MalisNLL end to end on the GPU (loss.py:560-690): a small affinity
net -> Softmax(n_indep = 3) -> MalisNLL -> AggregateLoss.

The MALIS counts depend on the ORDER of the predicted affinities, so the expectation is
built from the product's own float32 predictions: oracle counts (pure-Python Kruskal,
pinned by the reference's known-answer vectors) on those affinities, then the loss and
its gradient through a float64 torch restatement of the net with the counts as
constants -- the reference's semantics (malisop.py:114-120: zero gradient)."""
import os

import numpy as np
import pytest
import torch

from oracle import malis_oracle as MO
from oracle import torch_step as TS

pytestmark = pytest.mark.gpu
SP = (5, 14, 14)
OSP = (5, 12, 12)
NHOOD = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, -1]], np.int32)     # data/image.py:30-77


def _seg():
    """three slabs along x (every id one connected blob) and a background plane"""
    seg = np.zeros(OSP, np.int32)
    for x in range(OSP[1]):
        seg[:, x, :] = 1 + x // 4
    seg[0] = 0
    seg[3, 5:9, 2:6] = 7                      # an island inside slab 2
    return seg


def _build(params=None):
    from elektronn2_amd import neuromancer as nm, malis
    nm.model_manager.reset()
    inp = nm.Input((1, 1) + SP, 'b,f,z,x,y', name='raw')
    kw = [{}, {}] if params is None else [dict(w=params[0][0], b=params[0][1]),
                                          dict(w=params[1][0], b=params[1][1])]
    c = nm.Conv(inp, 8, (1, 3, 3), **kw[0])
    out = nm.Conv(c, 6, (1, 1, 1), activation_func='lin', **kw[1])
    probs = nm.Softmax(out, n_indep=3)
    aff_gt = nm.Input((1, 3) + OSP, 'b,f,z,x,y', name='aff_gt', dtype='int16')
    seg_gt = nm.Input((1, 1) + OSP, 'b,f,z,x,y', name='seg_gt', dtype='int16')
    nll = nm.MalisNLL(probs, aff_gt, seg_gt, NHOOD, unrestrict_neg=True)
    loss = nm.AggregateLoss(nll)
    m = nm.model_manager.getmodel()
    m.designate_nodes(input_node=inp, target_node=aff_gt, loss_node=loss,
                      prediction_node=probs)
    m.set_opt_meta_params('Adam', dict(lr=2e-3, mom=0.9, beta2=0.999, wd=0.5e-4))
    return m, nll


def _params(seed):
    rng = np.random.RandomState(seed)
    return [(rng.randn(8, 1, 1, 3, 3).astype(np.float32) * 0.5,
             rng.randn(8).astype(np.float32) * 0.1),
            (rng.randn(6, 8, 1, 1, 1).astype(np.float32) * 0.5,
             rng.randn(6).astype(np.float32) * 0.1)]


def _inputs(seed):
    from elektronn2_amd import malis
    rng = np.random.RandomState(seed)
    x = rng.rand(1, 1, *SP).astype(np.float32)
    seg = _seg()
    aff = malis.seg_to_affgraph(seg, NHOOD)
    return x, aff[None].astype(np.int16), seg[None, None].astype(np.int16)


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def test_loss_counts_and_gradients_match_the_oracle():
    from elektronn2_amd import malis
    P = _params(1)
    m, nll = _build(P)
    x, aff_gt, seg_gt = _inputs(2)
    probs = m.predict(x)                                   # (1, 6, z, x, y) float32
    assert np.allclose(probs[0, 0::2] + probs[0, 1::2], 1.0, atol=1e-6)
    loss = float(m.loss(x, aff_gt, seg_gt))
    pos, neg = MO.malis_weights(probs[0, 1::2], aff_gt[0], seg_gt[0, 0], NHOOD,
                                unrestrict_neg=True)
    assert np.array_equal(nll.pos_count, pos) and np.array_equal(nll.neg_count, neg)
    assert pos.sum() > 0 and neg.sum() > 0
    want_loss, _ = MO.malis_nll(probs[0], pos, neg)
    assert abs(loss - want_loss) <= 1e-4 * abs(want_loss)
    a = probs[0, 1::2]
    fs, fm = int(pos[a < 0.5].sum()), int(neg[a > 0.5].sum())
    assert nll.false_splits == fs and nll.false_merges == fm
    assert abs(float(nll.rand_index) - (fs + fm) / float(pos.sum() + neg.sum())) < 1e-6

    # gradients: float64 torch restatement of the net, counts as constants
    tp = [(torch.tensor(w, dtype=torch.float64, requires_grad=True),
           torch.tensor(b, dtype=torch.float64, requires_grad=True)) for w, b in P]
    h = TS.conv_node(torch.tensor(x, dtype=torch.float64), tp[0][0], tp[0][1], (1, 1, 1), 'relu')
    lg = TS.conv_node(h, tp[1][0], tp[1][1], (1, 1, 1), 'lin')[0]
    pr = torch.softmax(lg.view(3, 2, *OSP), dim=1).view(6, *OSP)
    Pt, Nt = torch.tensor(pos.astype(np.float64)), torch.tensor(neg.astype(np.float64))
    n_tot = Pt.sum() + Nt.sum()
    L = -(Pt * torch.log(pr[1::2] + 1e-5) + Nt * torch.log(pr[0::2] + 1e-5)).sum() / (n_tot + 1e-5)
    L.backward()
    assert abs(loss - float(L.detach())) <= 1e-4 * abs(float(L.detach()))
    got = m.gradients(x, aff_gt, seg_gt)
    names = list(m.loss_node.all_trainable_params.keys())
    want = {}
    for i, (w, b) in enumerate(tp):
        want[('w', i)] = w.grad.numpy(); want[('b', i)] = b.grad.numpy()
    assert len(got) == 4
    for g, nme in zip(got, names):
        cands = [v for v in want.values() if v.shape == g.shape]
        assert any(rel(g, c) < 1e-4 for c in cands), (nme, [rel(g, c) for c in cands])


def test_training_steps_graph_replay_matches_eager_and_learns():
    x, aff_gt, seg_gt = _inputs(3)
    finals, losses = [], []
    for no_graph in ("0", "1"):
        from elektronn2_amd.neuromancer import plan_options
        with plan_options(graph=no_graph == "0"):
            m, nll = _build(_params(4))
            ls = [float(m.trainingstep(x, aff_gt, seg_gt, optimiser='Adam')[0])
                  for _ in range(40)]
        finals.append(m.P.cpu().numpy().copy())
        losses.append(ls)
        assert np.isfinite(ls).all()
        assert nll.pos_count is not None and nll.pos_count.dtype == np.uint64
    # same arithmetic either way; the MALIS counts could only differ if two affinities
    # swapped order through last-bit noise of the atomically accumulated gradients
    assert np.abs(finals[0] - finals[1]).max() <= 1e-3 * np.abs(finals[1]).max()
    assert abs(losses[0][0] - losses[1][0]) <= 1e-5 * abs(losses[1][0])
    assert np.mean(losses[0][-5:]) < 0.8 * np.mean(losses[0][:5])


def test_sampler_feeds_malis_training():
    """cnndata.py:377-382: getbatch(affinities='malis') -> (images, aff, seg) device
    tensors that a MalisNLL step consumes as they are"""
    from elektronn2_amd.data import PatchSampler, make_affinities
    m, nll = _build(_params(6))
    rng = np.random.RandomState(0)
    vol = rng.rand(1, 12, 40, 40).astype(np.float32)
    ids = np.zeros((1, 12, 40, 40), np.float32)
    for x in range(40):
        ids[0, :, x, :] = 1 + x // 5
    ids[0, :, :, 18:21] = 0
    tn = m.nodes['aff_gt']
    offs = tuple((a - b) // 2 for a, b in zip(SP, OSP))
    smp = PatchSampler([vol], [ids], SP, (1, 1, 1), offs, seed=1, target_discrete_ix=[0])
    d, aff, seg = smp.getbatch(1, 'train', affinities='malis', nhood=NHOOD)
    assert d.is_cuda and tuple(aff.shape) == (1, 3) + OSP and tuple(seg.shape) == (1, 1) + OSP
    a2, s2 = make_affinities(np.rint(seg[:, 0].cpu().numpy()).astype(np.int32), NHOOD)
    assert np.array_equal(a2, aff.cpu().numpy().astype(np.int16))     # seg is self-consistent
    for _ in range(3):
        d, aff, seg = smp.getbatch(1, 'train', warp=0.5, affinities='malis', nhood=NHOOD)
        loss = float(m.trainingstep(d, aff, seg, optimiser='Adam')[0])
        assert np.isfinite(loss) and loss > 0
    assert tuple(tn.shape.spatial_shape) == OSP
