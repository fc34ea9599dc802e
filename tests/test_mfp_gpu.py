"""Max-fragment pooling (neural.py:666-675, 862-903; computations.py:652-701):
``Conv(mfp=True)`` pools at every offset of the pooling window and stacks the fragments
on the batch axis, ``FragmentsToDense`` interleaves them -- ONE forward pass predicts a
dense block that the plain net needs prod(strides) shifted passes for.

Checked against (a) the plain net evaluated at every stride offset (the product's own
predict_dense, itself held to the oracle in test_model_gpu.py) and (b) the f64 oracle's
per-voxel field-of-view evaluation."""
import numpy as np
import pytest

from oracle import e2_oracle as O

pytestmark = pytest.mark.gpu


def test_mfp_bookkeeping_and_errors():
    from elektronn2_amd import nets, neuromancer as nm
    nm.model_manager.reset()
    # training extents do not satisfy the MFP rule: (s - f + 1 - p + 1) % p == 0
    with pytest.raises(ValueError, match="using MFP"):
        nets.neuro3d_lite((1, 1, 7, 47, 47), mfp=True)
    nm.model_manager.reset()
    with pytest.raises(ValueError, match="batchsize of the raw image input must be 1"):
        nets.neuro3d_lite((2, 1, 8, 54, 54), mfp=True)
    nm.model_manager.reset()
    m = nets.neuro3d_lite((1, 1, 8, 54, 54), mfp=True)
    sm = m.nodes['softmax']
    assert sm.shape['b'] == 32 and list(sm.shape.strides) == [2, 4, 4]
    offs = np.asarray(sm.shape.mfp_offsets)
    assert offs.shape == (32, 3)
    assert len({tuple(o) for o in offs}) == 32                       # every offset once
    assert offs.min() == 0 and list(offs.max(0)) == [1, 3, 3]
    pn = m.prediction_node
    assert isinstance(pn, nm.FragmentsToDense)
    assert pn.shape['b'] == 1 and list(pn.shape.strides) == [1, 1, 1]
    assert pn.shape.fov == [5, 39, 39]
    sp = sm.shape.spatial_shape
    assert pn.shape.spatial_shape == [sp[0] * 2, sp[1] * 4, sp[2] * 4]
    assert pn.shape.spatial_shape == [8 - 5 + 1, 54 - 39 + 1, 54 - 39 + 1]


def test_mfp_prediction_equals_offset_interleave_and_oracle():
    from elektronn2_amd import nets, neuromancer as nm
    spec = O.NEURO3D_LITE
    params = O.init_net(spec, 1, seed=4)
    rng = np.random.RandomState(1)
    raw = rng.rand(1, 10, 62, 58).astype(np.float32)
    nm.model_manager.reset()
    plain = nets.neuro3d_lite((None, 1, 7, 47, 47), params=params)
    want = plain.predict_dense(raw)                  # 32 shifted passes per block
    nm.model_manager.reset()
    mfp = nets.neuro3d_lite((1, 1, 8, 54, 54), params=params, mfp=True)
    # one pass: the dense block is the whole valid extent of the patch
    x = raw[None, :, :8, :54, :54]
    dense = mfp.predict(x)
    assert dense.shape == (1, 2, 4, 16, 16)
    assert np.abs(dense[0] - want[:, :4, :16, :16]).max() < 1e-5
    # ... and the tiled form over the whole image
    got = mfp.predict_dense(raw)
    assert got.shape == want.shape == (2, 10 - 4, 62 - 38, 58 - 38)
    assert np.abs(got - want).max() < 1e-5
    # oracle: a few voxels by their own field of view
    P = [(np.asarray(w, np.float64), np.asarray(b, np.float64)) for w, b in params]
    for (z, xx, y) in [(0, 0, 0), (5, 23, 19), (2, 7, 11)]:
        ref = O.predict_voxel(spec, P, raw.astype(np.float64), (z, xx, y))
        assert np.abs(got[:, z, xx, y] - ref).max() < 1e-4
