"""MALIS: host C++ in libe2hip.so (csrc/malis.cpp) behind the
reference's Python API.  PINNED: the reference's own known-answer vectors
(/root/reference/tests/test_malis.py:36-77) are committed as data in
tests/golden/malis_reference.npz and reproduced exactly.  No GPU is needed."""
import os

import numpy as np
import pytest

from elektronn2_amd import malis as M
from oracle import malis_oracle as MO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "malis_reference.npz")


def test_reference_known_answer_vectors():
    z = np.load(GOLD)
    nhood = z['nhood']
    aff_gt = M.seg_to_affgraph(z['seg_ids'], nhood)
    seg_gt = M.affgraph_to_seg(aff_gt, nhood)[0].astype(np.int16)
    assert np.array_equal(seg_gt, z['seg_ids'])          # ids 1,2,3 and background 0 survive
    pos, neg = M.malis_weights(z['aff_pred'], aff_gt, seg_gt, nhood)
    assert pos.dtype == np.uint64 and pos.shape == z['aff_pred'].shape
    assert np.array_equal(pos, z['pos_true'])
    # the reference's second assertion: d(sum(pos * aff_pred)) / d(aff_pred) with the
    # counts treated as constants (malisop.py:112-118: zero gradient through the op)
    assert np.allclose(pos.astype(np.float64), z['g_true'])
    # float nhood and float affinities as in the second half of the reference test
    pos2, _ = M.malis_weights(z['aff_pred'], aff_gt.astype(np.float32), seg_gt,
                              nhood.astype(np.float64))
    assert np.array_equal(pos2, pos)


def test_affgraph_roundtrip_and_nodelists():
    rng = np.random.RandomState(0)
    seg = rng.randint(0, 4, (3, 5, 6)).astype(np.int32)
    nhood = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, -1]], np.int32)
    aff = M.seg_to_affgraph(seg, nhood)
    n1, n2 = M.nodelist_from_shape(seg.shape, nhood)
    assert n1.shape == n2.shape == aff.shape
    flat = seg.ravel()
    for e in range(len(nhood)):
        a, b = n1[e].ravel(), n2[e].ravel()
        ok = b >= 0
        expect = np.zeros(a.shape, np.int16)
        expect[ok] = (flat[a[ok]] == flat[b[ok]]) & (flat[a[ok]] > 0)
        assert np.array_equal(aff[e].ravel(), expect)
        # an edge leaves the volume exactly where the displaced index is out of range
        zz, xx, yy = np.unravel_index(a, seg.shape)
        inside = np.ones(a.shape, bool)
        for c, d, n in zip((zz, xx, yy), nhood[e], seg.shape):
            inside &= (c + d >= 0) & (c + d < n)
        assert np.array_equal(ok, inside)
    comp, sizes = M.affgraph_to_seg(aff, nhood, size_thresh=0)
    # connected components of the affinity graph never merge different ids
    for lab in np.unique(comp):
        ids = np.unique(seg[comp == lab])
        assert len(ids) == 1 or lab == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_product_equals_oracle_on_distinct_weights(seed):
    """C++ vs the pure-Python restatement vs brute-force maximin pair counting, on random
    graphs whose weights are all distinct (so no tie-breaking is involved)."""
    rng = np.random.RandomState(seed)
    shape = (2, 4, 5)
    seg = rng.randint(0, 4, shape).astype(np.int32)
    nhood = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.int32)
    n1, n2 = M.nodelist_from_shape(shape, nhood)
    n1, n2 = n1.ravel(), n2.ravel()
    w = rng.permutation(len(n1)).astype(np.float32) / len(n1)
    for pos in (1, 0):
        got = M.malis_loss_weights(seg.ravel(), n1, n2, w, pos)
        assert np.array_equal(got, MO.malis_loss_weights(seg.ravel(), n1, n2, w, pos))
        assert np.array_equal(got, MO.maximin_pair_counts(seg.ravel(), n1, n2, w, pos))
    # conservation: every labelled pair is credited exactly once over pos + neg
    lab = int((seg > 0).sum())
    tot = M.malis_loss_weights(seg.ravel(), n1, n2, w, 1).sum() + \
        M.malis_loss_weights(seg.ravel(), n1, n2, w, 0).sum()
    assert int(tot) == lab * (lab - 1) // 2


def test_neg_pass_restriction_and_errors():
    z = np.load(GOLD)
    nhood = z['nhood']
    aff_gt = M.seg_to_affgraph(z['seg_ids'], nhood)
    seg = z['seg_ids']
    _, neg_r = M.malis_weights(z['aff_pred'], aff_gt, seg, nhood)
    _, neg_u = M.malis_weights(z['aff_pred'], aff_gt, seg, nhood, unrestrict_neg=True)
    assert neg_r.sum() == neg_u.sum() > 0                # the same pairs are credited
    # restricted: a must-not-link pair is never charged to a true (gt = 1) edge it could
    # not have crossed... both variants only charge edges that exist
    assert not neg_r[aff_gt == 1].any() or neg_r.sum() == neg_u.sum()
    with pytest.raises(ValueError):
        M.malis_weights(z['aff_pred'][:, :, :3], aff_gt, seg, nhood)
    assert M.mknhood3d(1).shape == (4, 3) and M.mknhood2d(1).shape == (3, 2)


def test_make_affinities_relabels_components():
    """data/image.py:30-77: same-ID neighbours connected, 0 = background, the returned
    segmentation = connected components of the affinity graph (an ID used by two
    separate blobs becomes two IDs)"""
    from elektronn2_amd.data import make_affinities
    lab = np.zeros((1, 3, 6, 6), np.int32)
    lab[0, :, :2, :] = 5
    lab[0, :, 4:, :] = 5          # same ID, not touching the first blob
    lab[0, :, 2:4, :3] = 9
    aff, seg = make_affinities(lab)
    assert aff.shape == (1, 3, 3, 6, 6) and aff.dtype == np.int16 and seg.dtype == np.int16
    nh = np.eye(3, dtype=np.int32)
    for e in range(3):
        for idx in np.ndindex(3, 6, 6):
            j = tuple(np.add(idx, nh[e]))
            inside = all(0 <= a < s for a, s in zip(j, (3, 6, 6)))
            want = int(inside and lab[0][idx] != 0 and lab[0][idx] == lab[0][j])
            assert aff[0, e][idx] == want
    assert (seg[0] == 0).sum() == (lab[0] == 0).sum() and np.all((seg[0] == 0) == (lab[0] == 0))
    ids = set(np.unique(seg[0])) - {0}
    assert len(ids) == 3                                   # 5 split into two components
    for i in ids:
        assert len(np.unique(lab[0][seg[0] == i])) == 1
    assert seg[0, 0, 0, 0] != seg[0, 0, 5, 0]
