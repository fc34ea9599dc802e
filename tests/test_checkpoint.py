"""Model.save / modelload (model.py:229-235, 623-729) -- the graph
description round trip and the patch-size rules on the CPU, the training-state round
trip and the rebuilt graphs' numerics on the GPU."""
import os

import numpy as np
import pytest

from oracle import e2_oracle as O


def _lite(sp=(7, 47, 47), seed=1):
    from elektronn2_amd import nets, neuromancer as nm
    nm.model_manager.reset()
    params = O.init_net(O.NEURO3D_LITE, 1, seed=seed)
    m = nets.neuro3d_lite((None, 1) + sp, params=params)
    m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    return m, params


def test_closest_valid_patch_size_matches_the_reference_rule():
    """valid xy extents quoted in SURVEY.md 8(d): lite 151,155,...,183,187; neuro3d
    153,157,...,181,185,189 (validity: neural.py:746-750, cnncalculator.py:21-39);
    'closest' = the largest valid extent <= the wish, the smallest one if the wish is
    below it (cnncalculator.py:83-103)."""
    from elektronn2_amd.neuromancer.model import closest_valid_patch_size
    lite_f = [(1, 4, 4), (3, 3, 3), (2, 4, 4), (1, 3, 3), (1, 3, 3), (1, 1, 1), (1, 1, 1)]
    lite_p = [(1, 2, 2), (1, 2, 2), (2, 1, 1)] + [(1, 1, 1)] * 4
    assert closest_valid_patch_size(lite_f, lite_p, (23, 183, 183)) == (23, 183, 183)
    assert closest_valid_patch_size(lite_f, lite_p, (23, 186, 184)) == (23, 183, 183)
    assert closest_valid_patch_size(lite_f, lite_p, (24, 187, 190)) == (23, 187, 187)
    full_f = [(1, 6, 6), (1, 5, 5), (1, 5, 5), (4, 4, 4), (3, 4, 4), (3, 4, 4), (2, 4, 4),
              (1, 4, 4), (1, 4, 4), (1, 1, 1), (1, 1, 1)]
    full_p = [(1, 2, 2), (1, 2, 2), (1, 1, 1), (2, 1, 1)] + [(1, 1, 1)] * 7
    assert closest_valid_patch_size(full_f, full_p, (23, 183, 188)) == (23, 181, 185)
    small = closest_valid_patch_size(full_f, full_p, (2, 2, 2))
    assert closest_valid_patch_size(full_f, full_p, small) == small       # the smallest valid one
    # max-fragment pooling turns "divisible" into "remainder 1" (cnncalculator.py:32-35)
    assert closest_valid_patch_size(lite_f, lite_p, (24, 186, 186), [True] * 7) == (24, 186, 186)


def test_graph_description_round_trip(tmp_path):
    """save -> modelload(file) rebuilds every node (same names, classes, shapes, strides,
    fov) and restores every parameter; the file name is used as given (np.savez would
    append '.npz'); batch / patch overrides follow model.py:649-713."""
    from elektronn2_amd import nets, neuromancer as nm
    from elektronn2_amd.neuromancer.model import modelload, params_from_model_file
    m, params = _lite((23, 183, 183))
    f = str(tmp_path / "net.mdl")
    m.save(f)
    assert os.path.exists(f) and not os.path.exists(f + ".npz")
    m2 = modelload(f, name='rebuilt')
    assert list(m2.nodes.keys()) == list(m.nodes.keys())
    for k in m.nodes:
        a, b = m.nodes[k], m2.nodes[k]
        assert type(a).__name__ == type(b).__name__
        assert a.shape.shape == b.shape.shape and a.shape.tags == b.shape.tags
        assert list(a.shape.strides) == list(b.shape.strides) and a.shape.fov == b.shape.fov
        for pk, pv in a.get_param_values().items():
            assert np.array_equal(pv, b.get_param_values()[pk])
    assert m2.input_node.name == 'raw' and m2.loss_node.name == 'loss'
    assert [n.name for n in m2.prediction_ext] == [n.name for n in m.prediction_ext]
    pf = params_from_model_file(f)
    assert np.array_equal(pf['conv3']['w'], np.asarray(params[3][0], np.float32))
    # batch and patch size imposed at load time; the target follows the prediction
    m3 = modelload(f, name='bigger', imposed_patch_size=(23, 200, 190), imposed_batch_size=2)
    assert m3.input_node.shape.shape == [2, 1, 23, 199, 187]
    assert m3.prediction_node.shape.shape == [2, 2, 10, 41, 38]
    assert m3.target_node.shape.shape == [2, 1, 10, 41, 38]
    with pytest.raises(ValueError):
        modelload(f, name='bad', imposed_patch_size=(23, 183))
    # max-fragment pooling switched on at load time: dense output, prediction path only
    m4 = modelload(f, name='mfp', override_mfp_to_active=True,
                   imposed_patch_size=(24, 186, 186), imposed_batch_size=1)
    assert m4.prediction_node.shape.shape == [1, 2, 20, 148, 148]
    assert m4.loss_node is None
    # a U-Net (UpConvMerge -> UpConv + Crop + Concat nodes) rebuilds too
    nm.model_manager.reset()
    mu = nets.unet3d_lite()
    mu.save(f)
    mu2 = modelload(f, name='unet')
    assert [type(n).__name__ for n in mu2.nodes.values()] == [type(n).__name__ for n in mu.nodes.values()]
    assert mu2.prediction_node.shape.shape == mu.prediction_node.shape.shape
    assert mu2.prediction_node.shape.fov == mu.prediction_node.shape.fov   # fixed up in designate_nodes


def test_config1_graph_round_trip(tmp_path):
    """the mnist graph (2-D convs with batch norm, Perceptrons, 'b,f' tensors) through
    save -> modelload: classes, shapes, every parameter incl. the running statistics; batch
    override"""
    from elektronn2_amd import nets, neuromancer as nm
    from elektronn2_amd.neuromancer.model import modelload
    nm.model_manager.reset()
    np.random.seed(0)
    m = nets.mnist()
    m.nodes['conv'].mean.set_value(np.arange(12, dtype=np.float32))
    f = str(tmp_path / "mnist.mdl")
    m.save(f)
    m2 = modelload(f, name='rebuilt')
    assert list(m2.nodes.keys()) == list(m.nodes.keys())
    for k in m.nodes:
        a, b = m.nodes[k], m2.nodes[k]
        assert type(a).__name__ == type(b).__name__, k
        assert list(a.shape.shape) == list(b.shape.shape) and list(a.shape.tags) == list(b.shape.tags)
        for pk in a.params:
            assert np.array_equal(a.params[pk].get_value(), b.params[pk].get_value()), (k, pk)
            assert a.params[pk].apply_train == b.params[pk].apply_train
            assert a.params[pk].apply_reg == b.params[pk].apply_reg
    assert m2.nodes['conv'].batch_normalisation == 'train' and m2.nodes['dot'].flatten
    m3 = modelload(f, name='b5', imposed_batch_size=5)
    assert list(m3.input_node.shape.shape) == [5, 1, 26, 26]


def test_optimiser_state_survives_load_save_load_before_any_step(tmp_path):
    """modelload(f) followed by save() BEFORE the first step (the device buffers of the
    optimiser do not exist yet, the loaded state is parked): the second file must still
    hold Adam's m, s, t and SGD's last_dir -- a resume from it continues the moments and
    the bias correction instead of silently restarting them (the reference drops the
    optimiser state altogether, model.py:229-235)."""
    from elektronn2_amd.neuromancer.model import modelload
    a, _ = _lite()
    n = sum(int(np.prod(p.get_value().shape)) for p in a.loss_node.all_trainable_params.values())
    rng = np.random.RandomState(11)
    m0, s0, d0 = (rng.randn(max(n, 4)).astype(np.float32), rng.rand(max(n, 4)).astype(np.float32),
                  rng.randn(max(n, 4)).astype(np.float32))
    assert a.optimisers['Adam'].state_dict() == {} and a.optimisers['SGD'].state_dict() == {}
    a.optimisers['Adam'].load_state_dict({'m': m0, 's': s0, 't': 7})
    a.optimisers['SGD'].load_state_dict({'last_dir': d0})
    f1, f2 = str(tmp_path / "one.mdl"), str(tmp_path / "two.mdl")
    a.save(f1)                                   # state_dict() hands on the parked state
    b, _ = _lite(seed=5)
    modelload(f1, b)
    b.save(f2)                                   # load -> save, still no step
    c, _ = _lite(seed=6)
    modelload(f2, c)
    st = c.optimisers['Adam'].state_dict()
    assert float(st['t']) == 7.0 and c.optimisers['Adam'].t == 7.0
    assert np.array_equal(st['m'], m0) and np.array_equal(st['s'], s0)
    assert np.array_equal(c.optimisers['SGD'].state_dict()['last_dir'], d0)
    raw = np.load(f2)
    assert 'o/Adam/m' in raw.files and 'o/Adam/t' in raw.files and 'o/SGD/last_dir' in raw.files


@pytest.mark.gpu
def test_resume_restores_parameters_and_adam_state(tmp_path):
    """train 3 steps, save, go on for 2 steps; a FRESH model (other initial weights, no
    device state yet) loads the file and takes the same 2 steps: parameters, Adam m / s /
    t right after loading are bit-identical to what was saved, and the continued run
    matches the uninterrupted one (to the order of the split-K atomics)."""
    from elektronn2_amd.neuromancer.model import modelload
    sp = (7, 47, 47)
    rng = np.random.RandomState(3)
    xs = [rng.rand(1, 1, *sp).astype(np.float32) for _ in range(5)]
    ts = [rng.randint(0, 2, (1, 1) + O.net_out_shape(O.NEURO3D_LITE, sp)).astype(np.float32)
          for _ in range(5)]
    a, _ = _lite(sp, seed=1)
    for i in range(3):
        a.trainingstep(xs[i], ts[i], optimiser='Adam')
    f = str(tmp_path / "state.mdl")
    a.save(f)
    saved_p = [p.get_value() for p in a.loss_node.all_trainable_params.values()]
    saved_o = a.optimisers['Adam'].state_dict()
    assert float(saved_o['t']) == 3.0 and np.abs(saved_o['m']).max() > 0
    cont = [float(a.trainingstep(xs[i], ts[i], optimiser='Adam')[0]) for i in (3, 4)]
    end_p = [p.get_value() for p in a.loss_node.all_trainable_params.values()]

    b, _ = _lite(sp, seed=9)                     # different weights, nothing on the device yet
    modelload(f, b)
    assert b.iterations == 3
    for v, p in zip(saved_p, b.loss_node.all_trainable_params.values()):
        assert np.array_equal(v, p.get_value())
    assert b.optimisers['Adam'].t == 3.0         # pending until the first step creates the buffers
    got = [float(b.trainingstep(xs[i], ts[i], optimiser='Adam')[0]) for i in (3, 4)]
    st = b.optimisers['Adam'].state_dict()
    assert float(st['t']) == 5.0
    # bounds: two IDENTICAL runs of this net drift apart by 2e-7 (losses), 7e-7 (parameters),
    # 1.5e-6 (m) and 3e-7 (s) over nine steps (tools/aa_spread.py, DESIGN finding 53); a resume
    # that lost anything is off by > 1e-3
    for x, y in zip(cont, got):
        assert abs(x - y) <= 2e-6 * abs(x)
    for v, p in zip(end_p, b.loss_node.all_trainable_params.values()):
        assert np.abs(v - p.get_value()).max() <= 1e-5 * np.abs(v).max()
    ref_o = a.optimisers['Adam'].state_dict()
    assert np.abs(st['m'] - ref_o['m']).max() <= 5e-5 * np.abs(ref_o['m']).max()
    assert np.abs(st['s'] - ref_o['s']).max() <= 1e-5 * np.abs(ref_o['s']).max()
    # a state that does not fit the model is refused when it is applied
    c, _ = _lite(sp, seed=4)
    c.optimisers['Adam'].load_state_dict({'m': np.zeros(3, np.float32),
                                          's': np.zeros(3, np.float32), 't': 1})
    with pytest.raises(ValueError):
        c.trainingstep(xs[0], ts[0], optimiser='Adam')


@pytest.mark.gpu
def test_rebuilt_graph_with_new_patch_size_predicts_like_the_oracle(tmp_path):
    """modelload with imposed_patch_size (model.py:691-713): the rebuilt net at another
    patch size computes what the oracle computes with the saved weights there."""
    from elektronn2_amd.neuromancer.model import modelload
    m, params = _lite((7, 47, 47))
    f = str(tmp_path / "w.mdl")
    m.save(f)
    big = modelload(f, name='big', imposed_patch_size=(9, 52, 57), imposed_batch_size=2)
    sp = tuple(big.input_node.shape.spatial_shape)
    assert sp == (9, 51, 55)
    rng = np.random.RandomState(2)
    x = rng.rand(2, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (2, 1) + O.net_out_shape(O.NEURO3D_LITE, sp)).astype(np.float32)
    loss_ref, _, probs_ref = O.net_loss_and_grads(O.NEURO3D_LITE, params, x, t)
    assert np.abs(big.predict(x) - probs_ref).max() < 1e-5
    assert abs(float(big.loss(x, t)) - loss_ref) < 1e-4 * abs(loss_ref)


@pytest.mark.gpu
def test_inputs_produced_late_on_another_stream_are_waited_for():
    """Plans run on streams of their own (non-blocking).  A device input whose producer is
    still running on the caller's stream when the step is submitted must be waited for:
    a long kernel is queued in front of the copy that fills the input; the loss has to be
    the loss of the FILLED input (checked against the oracle), not of the stale zeros."""
    import torch
    m, params = _lite((7, 47, 47))
    rng = np.random.RandomState(5)
    x = rng.rand(1, 1, 7, 47, 47).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(O.NEURO3D_LITE, (7, 47, 47))).astype(np.float32)
    ref, _, _ = O.net_loss_and_grads(O.NEURO3D_LITE, params, x, t)
    m.loss(x, t)                                   # build + first call
    xd = torch.zeros(x.shape, device='cuda')
    td = torch.from_numpy(t).cuda()
    src = torch.from_numpy(x).cuda()
    big = torch.rand(4096, 4096, device='cuda')
    torch.cuda.synchronize()
    for _ in range(20):                            # ~tens of ms of work on the current stream
        big = (big @ big).clamp_(0, 1)
    xd.copy_(src)                                  # ... and only then the input is filled
    got = float(m.loss(xd, td))
    assert abs(got - ref) < 1e-4 * abs(ref), (got, ref)
    # the plan's own input buffer may be written in place and handed back (no copy)
    plan = m.loss_node._output_func.func
    buf = plan.input_buffer(m.input_node)
    buf.copy_(src * 0.5)
    ref2, _, _ = O.net_loss_and_grads(O.NEURO3D_LITE, params, x * 0.5, t)
    assert abs(float(m.loss(buf, td)) - ref2) < 1e-4 * abs(ref2)
