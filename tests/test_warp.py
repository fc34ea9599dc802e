"""Patch / warp augmentation.  CPU: the oracle's warp_slice is pinned by
exact identities (no reference run is possible here: parity unpinned, see
oracle/warp_oracle.py).  GPU: e2_warp_slice / e2_grey_augment and the host logic on top
(elektronn2_amd/data) against that oracle."""
import numpy as np
import pytest

from oracle import warp_oracle as W


def test_oracle_pure_translation_is_a_crop():
    rng = np.random.RandomState(0)
    img = rng.rand(2, 12, 30, 31).astype(np.float32)
    tgt = rng.randint(0, 5, (1, 12, 30, 31)).astype(np.float32)
    ps, tps = (4, 10, 12), (2, 4, 6)
    M = W.translate(-3, -7, -5)                 # dest = src - (3,7,5)
    d, t = W.warp_slice(img, ps, M, target=tgt, target_ps=tps)
    assert np.array_equal(d, img[:, 3:7, 7:17, 5:17])
    assert np.array_equal(t, tgt[:, 4:6, 10:14, 8:14])       # centred sub-block


def test_oracle_flip_swap_and_integer_scale():
    rng = np.random.RandomState(1)
    img = rng.rand(1, 9, 20, 20).astype(np.float32)
    ps = (3, 8, 8)
    # flip x about the patch: dest x -> src (12 - x)
    M = W.chain_matrices([W.translate(0, 0, 0), np.linalg.inv(
        W.chain_matrices([W.translate(2, 12, 4), np.diag([1, -1, 1, 1]).astype(np.float32)]))])
    d, _ = W.warp_slice(img, ps, M)
    assert np.array_equal(d[0], img[0, 2:5, 12:4:-1, 4:12])
    # swap x and y
    S = np.eye(4, dtype=np.float32)[[0, 2, 1, 3]]
    M = np.linalg.inv(W.chain_matrices([W.translate(1, 3, 5), S])).astype(np.float32)
    d, _ = W.warp_slice(img, ps, M)
    assert np.array_equal(d[0], img[0, 1:4, 3:11, 5:13].transpose(0, 2, 1))
    # zoom 2x in y: dest y -> src y/2: odd destination voxels are midpoints
    M = W.chain_matrices([W.scale(1, 1, 2), W.translate(-2, -3, -4)])
    d, _ = W.warp_slice(img, ps, M)
    src = img[0, 2:5, 3:11, 4:9].astype(np.float64)
    assert np.allclose(d[0][:, :, 0::2], src[:, :, :4], atol=1e-7)
    assert np.allclose(d[0][:, :, 1::2], 0.5 * (src[:, :, :4] + src[:, :, 1:5]), atol=1e-7)


def test_oracle_out_of_bounds_and_centering_errors():
    img = np.zeros((1, 8, 16, 16), np.float32)
    with pytest.raises(W.WarpingOOBError):
        W.warp_slice(img, (4, 8, 8), W.translate(1, 0, 0))          # src z = -1
    with pytest.raises(W.WarpingOOBError):
        W.warp_slice(img, (4, 8, 8), W.translate(-4, -8, -8))       # + 1 for interpolation
    with pytest.raises(ValueError, match="centered"):
        W.warp_slice(img, (4, 8, 8), W.translate(-1, -1, -1), target=img[:, :7],
                     target_ps=(2, 4, 4))


def _rand_case(seed, perspective, lock_z=True):
    rng = np.random.RandomState(seed)
    np.random.seed(seed)
    img = rng.rand(2, 40, 120, 110).astype(np.float32)
    tgt = np.concatenate([rng.randint(0, 7, (1, 36, 116, 106)).astype(np.float32),
                          rng.rand(1, 36, 116, 106).astype(np.float32)])
    ps, tps = (12, 48, 44), (8, 20, 24)
    for _ in range(50):
        M = W.random_warp_matrix(img.shape[1:], ps, 2, True, 1.0, lock_z, False, perspective,
                                 tgt.shape[1:], tps, rng)
        try:
            ref = W.warp_slice(img, ps, M, target=tgt, target_ps=tps, target_discrete_ix=[0])
            return img, tgt, ps, tps, M, ref
        except W.WarpingOOBError:
            continue
    raise RuntimeError("no in-bounds warp found")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,perspective,lock_z", [(0, False, True), (1, True, True),
                                                     (2, True, False), (3, False, False)])
def test_warp_slice_matches_oracle(ctx, seed, perspective, lock_z):
    """random rotation + warp (+ perspective) matrices drawn like get_warped_slice does:
    trilinear image channels within 1e-4 (fp32 coordinates and weights against the
    reference's float32 tensordot / float64 weights), the discrete target channel exact
    except where a coordinate sits within 1e-3 of a rounding boundary."""
    from elektronn2_amd.data import transformations as T
    img, tgt, ps, tps, M, (d_ref, t_ref) = _rand_case(seed, perspective, lock_z)
    d, t = T.warp_slice(img, ps, M, target=tgt, target_ps=tps, target_discrete_ix=[0])
    d, t = d.cpu().numpy(), t.cpu().numpy()
    assert d.shape == d_ref.shape and t.shape == t_ref.shape
    assert np.abs(d - d_ref).max() < 1e-4
    assert np.abs(t[1] - t_ref[1]).max() < 1e-4                 # continuous target channel
    mism = t[0] != t_ref[0]
    if mism.any():
        coords, _ = W.source_coords(ps, M)
        off_ps = np.subtract(ps, tps) // 2
        ct = coords[off_ps[0]:off_ps[0] + tps[0], off_ps[1]:off_ps[1] + tps[1],
                    off_ps[2]:off_ps[2] + tps[2]]
        frac = np.abs(ct - np.floor(ct) - 0.5).min(-1)
        assert mism.mean() < 1e-3 and frac[mism].max() < 1e-3


@pytest.mark.gpu
def test_warp_slice_errors_and_identities(ctx):
    from elektronn2_amd.data import transformations as T
    rng = np.random.RandomState(5)
    img = rng.rand(1, 12, 30, 31).astype(np.float32)
    d, _ = T.warp_slice(img, (4, 10, 12), T.translate(-3, -7, -5))
    assert np.array_equal(d.cpu().numpy(), img[:, 3:7, 7:17, 5:17])
    with pytest.raises(T.WarpingOOBError):
        T.warp_slice(img, (4, 10, 12), T.translate(1, 0, 0))
    with pytest.raises(ValueError, match="centered"):
        T.warp_slice(img, (4, 10, 12), T.translate(-3, -7, -5), target=img[:, :11],
                     target_ps=(2, 4, 4))
    with pytest.raises(NotImplementedError):
        T.warp_slice(img, (4, 10, 12), T.identity(), target_vec_ix=[(0, 1, 2)])


@pytest.mark.gpu
def test_patch_sampler_batches(ctx):
    """PatchSampler.getbatch = draw cube, warp-cut image + centred target, grey augment,
    stride the target (cnndata.py:214-402): shapes, determinism under a seed, and equality
    with the oracle driven by the same generator."""
    from elektronn2_amd.data import PatchSampler
    rng = np.random.RandomState(7)
    data = [rng.rand(1, 30, 100, 100).astype(np.float32) for _ in range(2)]
    tgts = [rng.randint(0, 2, (1, 30, 100, 100)).astype(np.float32) for _ in range(2)]
    ps, strides, offsets = (13, 47, 47), (2, 4, 4), (2, 19, 19)

    def run(seed):
        np.random.seed(seed)
        s = PatchSampler(data, tgts, ps, strides, offsets, seed=seed)
        return s, s.getbatch(3, 'train', grey_augment_channels=[0], warp=0.5,
                             warp_args={'sample_aniso': True, 'perspective': True})
    s1, (d1, t1) = run(11)
    s2, (d2, t2) = run(11)
    assert tuple(d1.shape) == (3, 1, 13, 47, 47) and tuple(t1.shape) == (3, 1, 5, 3, 3)
    assert torch_eq(d1, d2) and torch_eq(t1, t2)
    assert s1.n_successful_warp == 3
    # the same three patches from the oracle
    np.random.seed(11)
    g = np.random.RandomState(11)
    w = np.hstack((0, np.cumsum([0.5, 0.5])))
    k = 0
    while k < 3:
        i = int(np.flatnonzero(w <= g.rand())[-1])
        do_warp = g.rand() < 0.5
        M = W.random_warp_matrix(data[i].shape[1:], ps, 2, True, 1.0 if do_warp else 0.0, True,
                                 False, True, tgts[i].shape[1:], (9, 9, 9), g)
        try:
            d_ref, t_ref = W.warp_slice(data[i], ps, M, target=tgts[i], target_ps=(9, 9, 9))
        except W.WarpingOOBError:
            continue
        d_ref = W.grey_augment(d_ref, [0], g)
        assert np.abs(d1[k].cpu().numpy() - d_ref).max() < 2e-4
        tt = t_ref[:, ::2, ::4, ::4]
        # discrete target: exact, except where a source coordinate sits within 1e-3 of a
        # rounding boundary (the proof used by test_warp_slice_matches_oracle)
        mism = t1[k].cpu().numpy()[0] != tt[0]
        if mism.any():
            coords, _ = W.source_coords(ps, M)
            o = np.subtract(ps, (9, 9, 9)) // 2
            ct = coords[o[0]:o[0] + 9, o[1]:o[1] + 9, o[2]:o[2] + 9][::2, ::4, ::4]
            frac = np.abs(ct - np.floor(ct) - 0.5).min(-1)
            assert frac[mism].max() < 1e-3, frac[mism].max()
        k += 1


@pytest.mark.gpu
def test_patch_sampler_from_nodes_takes_a_reference_config(ctx):
    """BatchCreatorImage's constructor protocol (cnndata.py:110-176, 134-140): geometry
    from the model's input / target nodes, the rest from the config's data_init_kwargs
    (examples/neuro3d.py:17-24) passed through unchanged."""
    from elektronn2_amd import nets, neuromancer as nm
    from elektronn2_amd.data import PatchSampler
    nm.model_manager.reset()
    model = nets.neuro3d_lite((None, 1, 13, 47, 47))
    data_init_kwargs = {                      # as in examples/neuro3d.py
        'd_path': '~/neuro_data_zxy/', 'l_path': '~/neuro_data_zxy/',
        'd_files': [('raw_%i.h5' % i, 'raw') for i in range(3)],
        'l_files': [('barrier_int16_%i.h5' % i, 'lab') for i in range(3)],
        'aniso_factor': 2, 'valid_cubes': [2],
    }
    rng = np.random.RandomState(1)
    data = [rng.rand(1, 30, 100, 100).astype(np.float32) for _ in range(3)]
    tgts = [rng.randint(0, 2, (1, 30, 100, 100)).astype(np.float32) for _ in range(3)]
    s = PatchSampler.from_nodes(model.input_node, model.target_node, data=data, targets=tgts,
                                seed=5, **data_init_kwargs)
    assert s.patch_size == (13, 47, 47) and s.strides == (2, 4, 4) and s.offsets == (2, 19, 19)
    assert s.target_ps == (9, 9, 9) and s.valid == [2] and s.train == [0, 1]
    d, t = s.getbatch(2, 'train', grey_augment_channels=[0], warp=0.5,
                      warp_args={'sample_aniso': True, 'perspective': True})
    assert tuple(d.shape) == (2, 1, 13, 47, 47)
    assert tuple(t.shape) == (2,) + tuple(model.target_node.shape.shape[1:])
    dv, tv = s.getbatch(1, 'valid')
    loss, _, _ = model.trainingstep(d, t, optimiser='Adam')
    assert np.isfinite(loss)
    # sampling priorities instead of cube sizes (cnndata.py:535-541)
    s2 = PatchSampler.from_nodes(model.input_node, model.target_node, data=data, targets=tgts,
                                 cube_prios=[1, 3, 1], valid_cubes=[2], seed=5)
    assert np.allclose(s2._sampling_weight, [0, 0.25, 1.0])
    with pytest.raises(ValueError):
        PatchSampler.from_nodes(model.input_node, model.target_node, data=data, targets=tgts[:2])
    with pytest.raises(ValueError):
        PatchSampler.from_nodes(model.input_node, None, data=data, targets=tgts)


@pytest.mark.gpu
def test_training_fed_by_the_device_sampler(ctx):
    """the intended pipeline: volumes in HBM -> PatchSampler (device tensors) ->
    Model.trainingstep, no host copy of the patch; the loss on a learnable synthetic
    task (label = bright voxel) must fall."""
    from elektronn2_amd import nets, neuromancer as nm
    from elektronn2_amd.data import PatchSampler
    nm.model_manager.reset()
    np.random.seed(0)
    model = nets.neuro3d_lite((None, 1, 13, 47, 47))
    model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    rng = np.random.RandomState(0)
    vol = rng.rand(1, 40, 120, 120).astype(np.float32)
    import scipy.ndimage as ndi
    vol[0] = ndi.gaussian_filter(vol[0], 2.0)
    vol = (vol - vol.min()) / (vol.max() - vol.min())
    lab = (vol > np.median(vol)).astype(np.float32)
    tn = model.target_node
    smp = PatchSampler([vol], [lab], model.input_node.shape.spatial_shape, tn.shape.strides,
                       tn.shape.offsets, seed=3)
    val = [smp.getbatch(1, 'train') for _ in range(6)]           # fixed evaluation patches

    def val_loss():
        return float(np.mean([float(model.loss(d, t)) for d, t in val]))
    before = val_loss()
    losses = []
    for i in range(150):
        d, t = smp.getbatch(1, 'train', grey_augment_channels=[0], warp=0.5)
        assert d.is_cuda and t.is_cuda
        losses.append(float(model.trainingstep(d, t, optimiser='Adam')[0]))
    after = val_loss()
    assert np.all(np.isfinite(losses))
    assert after < 0.9 * before, (before, after)


@pytest.mark.gpu
def test_ring_feeder_and_deferred_multi_step_launches(ctx):
    """data.RingFeeder + Model.trainingsteps(k, ring, sync=False): the sampler fills the ring's
    next k slots while a launch of k steps runs; every launch returns the losses of the one before
    it.  The same sampler seed through single trainingstep calls gives the same losses step by
    step (to the weight gradients' atomic order) -- i.e. every step read the batch that was meant
    for it, across the ring's wrap-around and the two halves."""
    from elektronn2_amd import nets, neuromancer as nm
    from elektronn2_amd.data import PatchSampler, RingFeeder
    rng = np.random.RandomState(0)
    vol = rng.rand(1, 30, 100, 100).astype(np.float32)
    lab = (vol > 0.5).astype(np.float32)
    k, launches = 3, 4

    def make():
        nm.model_manager.reset()
        np.random.seed(0)
        model = nets.neuro3d_lite((None, 1, 9, 47, 47))
        model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
        tn = model.target_node
        smp = PatchSampler([vol], [lab], model.input_node.shape.spatial_shape, tn.shape.strides,
                           tn.shape.offsets, seed=5)
        return model, smp
    kw = dict(grey_augment_channels=[0], warp=0.5)
    model, smp = make()
    ref = []
    for i in range(1 + k * launches):
        d, t = smp.getbatch(1, 'train', **kw)
        ref.append(float(model.trainingstep(d, t, optimiser='Adam')[0]))

    model, smp = make()
    d, t = smp.getbatch(1, 'train', **kw)
    got = [float(model.trainingstep(d, t, optimiser='Adam')[0])]
    feeder = RingFeeder(smp, model, 'Adam', k, **kw)
    assert feeder.ring.shape[0] == 2 * k
    feeder.fill()
    for b in range(launches):
        losses, tl = model.trainingsteps(k, optimiser='Adam', ring=feeder.ring, sync=False)
        feeder.fill()
        assert (losses is None) == (b == 0)
        if losses is not None:
            assert len(losses) == k
            got += [float(v) for v in losses]
    plan = model.optimisers['Adam'].step.func
    got += [float(v) for v in plan.loss_history(k)]          # the last launch's (waits for it)
    assert model.iterations == 1 + k * launches and k in plan._multi
    assert len(got) == len(ref)
    for i, (u, v) in enumerate(zip(ref, got)):
        assert abs(u - v) <= 1e-5 * abs(u), (i, ref, got)


def torch_eq(a, b):
    import torch
    return bool(torch.equal(a, b))
