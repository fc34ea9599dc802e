"""End-to-end parity of the neuromancer-shaped front end + training plan against
the CPU oracle: loss, every dL/dw and dL/db, parameters after 1 and 3 Adam steps
(lr=5e-4, mom=0.9, beta2=0.999, wd=5e-5: examples/neuro3d.py:33-39).
Tolerance 1e-4 relative (BASELINE.json north_star), measured against the
largest magnitude of each tensor."""
import numpy as np
import pytest
import torch

from oracle import e2_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
# Parameters AFTER Adam steps: Adam divides by sqrt(s + 1e-5), which turns the
# fp32 rounding of small / cancelling gradient sums into a much larger relative
# error of the update (the f64 oracle has none).  Losses and gradients are held
# to TOL; updated parameters to TOL_ADAM of the tensor's max magnitude.
TOL_ADAM = 5e-4


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def build(spec_name, sp, params):
    from elektronn2_amd import nets, neuromancer as nm
    nm.model_manager.reset()
    fn = nets.neuro3d_lite if spec_name == 'lite' else nets.neuro3d
    m = fn((None, 1) + sp, params=params)
    m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    return m


CASES = [('lite', O.NEURO3D_LITE, (7, 47, 47)), ('full', O.NEURO3D, (17, 109, 109))]


@pytest.mark.parametrize("name,spec,sp", CASES, ids=['lite', 'full'])
def test_loss_grads_and_adam_steps(name, spec, sp):
    params = O.init_net(spec, 1, seed=1)
    rng = np.random.RandomState(0)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    osp = O.net_out_shape(spec, sp)
    t = rng.randint(0, 2, (1, 1) + osp).astype(np.float32)
    t.flat[::17] = -1                      # unlabelled voxels
    m = build(name, sp, params)

    loss_ref, grads_ref, probs_ref = O.net_loss_and_grads(spec, params, x, t)
    # forward-only functions
    assert abs(float(m.loss(x, t)) - loss_ref) / abs(loss_ref) < TOL
    assert rel(m.predict(x), probs_ref) < TOL
    l2, err, pr = m.predict_ext(x, t)
    assert abs(float(l2) - loss_ref) / abs(loss_ref) < TOL
    assert abs(float(err) - O.classification_errors(probs_ref, t)) < 1e-6
    # gradients (model.gradients == T.grad wrt trainable params, model.py:182-186)
    g = m.gradients(x, t)
    names = list(m.loss_node.all_trainable_params.keys())
    assert len(g) == 2 * len(spec)
    for i in range(len(spec)):
        gw, gb = g[names.index('conv%s_w' % (i or ''))], g[names.index('conv%s_b' % (i or ''))]
        assert rel(gw, grads_ref[i][0]) < TOL, "dW layer %d" % i
        assert rel(gb, grads_ref[i][1]) < TOL, "db layer %d" % i
    # 3 Adam steps on the fixed batch (2nd+ steps run from the captured hipGraph)
    ref_losses, ref_P = O.net_train_steps(spec, params, x, t, 3)
    for s in range(3):
        loss, tsec, _ = m.trainingstep(x, t, optimiser='Adam')
        assert abs(float(loss) - ref_losses[s]) / abs(ref_losses[s]) < TOL, "step %d" % s
        assert tsec > 0
    for i in range(len(spec)):
        node = m.nodes['conv%s' % (i or '')]
        assert rel(node.w.get_value(), ref_P[i][0]) < TOL_ADAM
        assert rel(node.b.get_value(), ref_P[i][1]) < TOL_ADAM
    assert m.iterations == 3


def test_plan_gives_its_weight_images_their_own_row_length():
    """Plan option image_stride (default on, DESIGN finding 52): after the first eager step -- the
    tilings are known -- the conv weight images are re-packed with rows as long as each launch's
    tiles reach, the launches are told so, and losses / parameters are those of the plan that
    keeps the any-tiling formula (to the weight gradients' atomic order); eager, capture, replay;
    a prediction plan of the same model does the same."""
    from elektronn2_amd.neuromancer import plan_options
    spec, sp = O.NEURO3D, (17, 109, 109)
    params = O.init_net(spec, 1, seed=4)
    rng = np.random.RandomState(6)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    res = {}
    for on in (False, True):
        with plan_options(image_stride=on):
            m = build('full', sp, params)
            losses = [float(m.trainingstep(x, t, optimiser='Adam')[0]) for _ in range(4)]
            plan = m.optimisers['Adam'].step.func
            assert bool(plan._img_stride) == on
            p1 = m.predict(x)
            p2 = m.predict(x)                      # (second call: images re-packed, graph captured)
            assert np.array_equal(p1, p2) or rel(p2, p1) < 1e-6
            res[on] = (losses, [q.get_value() for q in m.loss_node.all_trainable_params.values()], p2)
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) < 1e-5 * abs(b)
    for a, b in zip(res[True][1], res[False][1]):
        assert rel(a, b) < 1e-4
    assert rel(res[True][2], res[False][2]) < 1e-5


def test_sgd_step_and_lr_change_under_graph():
    spec, sp = O.NEURO3D_LITE, (7, 47, 47)
    params = O.init_net(spec, 1, seed=2)
    rng = np.random.RandomState(5)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    m = build('lite', sp, params)
    m.lr, m.mom, m.wd = 1e-3, 0.9, 1e-4
    P = [(np.asarray(w, np.float64), np.asarray(b, np.float64)) for w, b in params]
    D = [(np.zeros_like(w), np.zeros_like(b)) for w, b in P]
    for s, lr in enumerate([1e-3, 1e-3, 5e-4, 5e-4]):
        m.lr = lr                          # schedule change must reach the replayed graph
        _, grads, _ = O.net_loss_and_grads(spec, P, x, t)
        for i in range(len(P)):
            w, dw = O.sgd_step(P[i][0], grads[i][0], D[i][0], lr, 0.9, 1e-4, True)
            b, db = O.sgd_step(P[i][1], grads[i][1], D[i][1], lr, 0.9, 1e-4, False)
            P[i], D[i] = (w, b), (dw, db)
        m.trainingstep(x, t, optimiser='SGD')
    for i in range(len(P)):
        assert rel(m.nodes['conv%s' % (i or '')].w.get_value(), P[i][0]) < TOL


def test_unet_like_merge_parity():
    """Conv/Pool/UpConvMerge/Crop/Concat path (examples/unet3d_lite.py pattern),
    checked against torch-CPU autograd of the oracle's closed forms."""
    from elektronn2_amd import neuromancer as nm
    from oracle import torch_step as TS
    nm.model_manager.reset()
    np.random.seed(3)
    inp = nm.Input((1, 1, 12, 36, 36), 'b,f,z,x,y', name='raw')
    c0 = nm.Conv(inp, 8, (1, 3, 3))
    c1 = nm.Conv(c0, 8, (1, 3, 3))
    p1 = nm.Pool(c1, (1, 2, 2))
    c2 = nm.Conv(p1, 16, (3, 3, 3))
    c3 = nm.Conv(c2, 16, (3, 3, 3))
    mrg = nm.UpConvMerge(c1, c3, 24)
    c4 = nm.Conv(mrg, 8, (1, 3, 3))
    out = nm.Conv(c4, 2, (1, 1, 1), activation_func='lin')
    probs = nm.Softmax(out)
    target = nm.Input_like(probs, override_f=1, name='target')
    loss = nm.AggregateLoss(nm.MultinoulliNLL(probs, target, target_is_sparse=True), name='loss')
    model = nm.model_manager.getmodel()
    model.designate_nodes(input_node=inp, target_node=target, loss_node=loss,
                          prediction_node=probs)
    up = [n for n in model.nodes.values() if isinstance(n, nm.UpConv)][0]
    crop = [n for n in model.nodes.values() if isinstance(n, nm.Crop)][0]
    rng = np.random.RandomState(4)
    x = rng.rand(1, 1, 12, 36, 36).astype(np.float32)
    t = rng.randint(0, 2, [1, 1] + probs.shape.spatial_shape).astype(np.float32)

    def tt(p):
        return torch.tensor(p.get_value(), dtype=torch.float64, requires_grad=True)
    convs = [c0, c1, c2, c3, c4, out]
    W = {n.name: (tt(n.w), tt(n.b)) for n in convs + [up]}
    xt = torch.tensor(x, dtype=torch.float64)
    h0 = TS.conv_node(xt, *W['conv'], (1, 1, 1), 'relu')
    h1 = TS.conv_node(h0, *W[c1.name], (1, 1, 1), 'relu')
    hp = torch.nn.functional.max_pool3d(h1, (1, 2, 2))
    h2 = TS.conv_node(hp, *W[c2.name], (1, 1, 1), 'relu')
    h3 = TS.conv_node(h2, *W[c3.name], (1, 1, 1), 'relu')
    hu = TS.upconv_node(h3, *W[up.name], up.pool_shape, 'relu')
    cr = crop.crop
    hc = h1[:, :, cr[0]:h1.shape[2] - cr[0], cr[1]:h1.shape[3] - cr[1], cr[2]:h1.shape[4] - cr[2]]
    hm = torch.cat([hu, hc], dim=1)
    h4 = TS.conv_node(hm, *W[c4.name], (1, 1, 1), 'relu')
    lg = TS.conv_node(h4, *W[out.name], (1, 1, 1), 'lin')
    L, pr = TS.nll_loss(lg, torch.tensor(t, dtype=torch.float64))
    L.backward()
    assert abs(float(model.loss(x, t)) - float(L)) / float(L) < TOL
    g = model.gradients(x, t)
    names = list(model.loss_node.all_trainable_params.keys())
    for n in convs + [up]:
        assert rel(g[names.index(n.name + '_w')], W[n.name][0].grad.numpy()) < TOL, n.name
        assert rel(g[names.index(n.name + '_b')], W[n.name][1].grad.numpy()) < TOL, n.name


def test_upconv_on_an_input_node_trains():
    """An UpConv whose parent needs no gradient (directly on the Input): the packed backward
    gets neither dx nor the data gradient's image (api.hip upconv_bwd).  Loss and
    gradients against torch-CPU float64 autograd of the oracle's closed forms, eager and
    replayed."""
    from elektronn2_amd import neuromancer as nm
    from oracle import torch_step as TS
    nm.model_manager.reset()
    np.random.seed(5)
    inp = nm.Input((1, 1, 6, 20, 20), 'b,f,z,x,y', name='raw')
    up = nm.UpConv(inp, 6, (1, 2, 2))
    c0 = nm.Conv(up, 8, (1, 3, 3))
    c1 = nm.Conv(c0, 8, (1, 3, 3), (1, 2, 2))    # (the field of view must stay centred: model.py:141-152)
    out = nm.Conv(c1, 2, (1, 1, 1), activation_func='lin')
    probs = nm.Softmax(out)
    target = nm.Input_like(probs, override_f=1, name='target')
    loss = nm.AggregateLoss(nm.MultinoulliNLL(probs, target, target_is_sparse=True), name='loss')
    model = nm.model_manager.getmodel()
    model.designate_nodes(input_node=inp, target_node=target, loss_node=loss,
                          prediction_node=probs)
    rng = np.random.RandomState(6)
    x = rng.rand(1, 1, 6, 20, 20).astype(np.float32)
    t = rng.randint(0, 2, [1, 1] + probs.shape.spatial_shape).astype(np.float32)

    def tt(p):
        return torch.tensor(p.get_value(), dtype=torch.float64, requires_grad=True)
    W = {n.name: (tt(n.w), tt(n.b)) for n in (up, c0, c1, out)}
    hu = TS.upconv_node(torch.tensor(x, dtype=torch.float64), *W[up.name], up.pool_shape, 'relu')
    h0 = TS.conv_node(hu, *W[c0.name], (1, 1, 1), 'relu')
    h1 = TS.conv_node(h0, *W[c1.name], (1, 2, 2), 'relu')
    lg = TS.conv_node(h1, *W[out.name], (1, 1, 1), 'lin')
    L, _ = TS.nll_loss(lg, torch.tensor(t, dtype=torch.float64))
    L.backward()
    names = list(model.loss_node.all_trainable_params.keys())
    for _ in range(3):                       # eager, capture, replay
        g = model.gradients(x, t)
        for n in (up, c0, c1, out):
            assert rel(g[names.index(n.name + '_w')], W[n.name][0].grad.numpy()) < TOL, n.name
            assert rel(g[names.index(n.name + '_b')], W[n.name][1].grad.numpy()) < TOL, n.name
    l0 = float(model.trainingstep(x, t, optimiser='Adam')[0])
    assert abs(l0 - float(L)) / float(L) < TOL
    l1 = float(model.trainingstep(x, t, optimiser='Adam')[0])
    assert l1 < l0


def test_graph_replay_equals_eager_at_baseline_size():
    """C-lite@183 (BASELINE configs[1]): 12 Adam steps replayed from the captured
    hipGraphs against the same 12 steps launched eagerly -- the losses must agree step
    by step and the parameters at the end (what differs is only the order of fp32
    atomics).  Guards the capture path (memset / kernel ordering, baked arguments)."""
    from elektronn2_amd import neuromancer as nm
    spec, sp = O.NEURO3D_LITE, (23, 183, 183)
    params = O.init_net(spec, 1, seed=1)
    rng = np.random.RandomState(11)
    xs = [rng.rand(1, 1, *sp).astype(np.float32) for _ in range(3)]
    ts = [rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
          for _ in range(3)]
    runs = []
    for use_graph in (True, False):
        m = build('lite', sp, params)
        m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
        opt = m.optimisers['Adam']
        opt.step.compile()
        opt.step.func.use_graph = use_graph
        losses = [float(m.trainingstep(xs[i % 3], ts[i % 3], optimiser='Adam')[0])
                  for i in range(12)]
        runs.append((losses, [p.get_value() for p in m.loss_node.all_trainable_params.values()]))
    (lg, pg), (le, pe) = runs
    for a, b in zip(lg, le):
        assert abs(a - b) / abs(b) < 1e-5, (lg, le)
    for a, b in zip(pg, pe):
        assert rel(a, b) < 1e-4


def test_trainingstep_with_deferred_loss():
    """Model.trainingstep(sync=False): the step is submitted without waiting for it and the
    loss handed back is the PREVIOUS step's; the same batches from the same initial weights
    give the losses of the synchronous run shifted by one call, and the same parameters."""
    from elektronn2_amd import neuromancer as nm
    spec, sp = O.NEURO3D_LITE, (7, 47, 47)
    params = O.init_net(spec, 1, seed=1)
    rng = np.random.RandomState(21)
    xs = [rng.rand(1, 1, *sp).astype(np.float32) for _ in range(4)]
    ts = [rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32) for _ in range(4)]
    runs = []
    for sync in (True, False):
        m = build('lite', sp, params)
        m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
        out = [m.trainingstep(xs[i % 4], ts[i % 4], optimiser='Adam', sync=sync) for i in range(9)]
        assert all(o[2] is None and o[1] > 0 for o in out)
        torch.cuda.synchronize()
        runs.append(([float(o[0]) for o in out],
                     [p.get_value() for p in m.loss_node.all_trainable_params.values()], m.iterations))
    (ls, ps, n_s), (la, pa, n_a) = runs
    assert n_s == n_a == 9
    assert abs(la[0] - ls[0]) < 1e-6 * abs(ls[0])            # the first call waits for its own
    for i in range(1, 9):                                      # then: one call late
        assert abs(la[i] - ls[i - 1]) < 1e-5 * abs(ls[i - 1]), (i, la, ls)
    for a, b in zip(pa, ps):
        assert rel(a, b) < 1e-4
    # the two modes interleaved: a deferred call that follows a synchronous one has no
    # "previous deferred step" -- it waits for its own loss instead of handing back an older
    # step's (plan.fetch_async)
    m = build('lite', sp, params)
    m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    modes = [False, False, True, False, False, True, True, False]
    got = [float(m.trainingstep(xs[i % 4], ts[i % 4], optimiser='Adam', sync=modes[i])[0])
           for i in range(8)]
    want = [ls[0], ls[0], ls[2], ls[3], ls[3], ls[5], ls[6], ls[7]]
    for i in range(8):
        assert abs(got[i] - want[i]) < 1e-5 * abs(want[i]), (i, got, want)


def _dense_model(spec, params, patch_sp):
    from elektronn2_amd import neuromancer as nm
    nm.model_manager.reset()
    inp = nm.Input((None, 1) + tuple(patch_sp), 'b,f,z,x,y', name='raw')
    out = inp
    for (n_f, k, p, act), (w, b) in zip(spec, params):
        out = nm.Conv(out, n_f, k, p, activation_func=act, w=np.asarray(w, np.float32),
                      b=np.asarray(b, np.float32))
    probs = nm.Softmax(out)
    target = nm.Input_like(probs, override_f=1, name='target')
    loss = nm.AggregateLoss(nm.MultinoulliNLL(probs, target, target_is_sparse=True), name='loss')
    model = nm.model_manager.getmodel()
    model.designate_nodes(input_node=inp, target_node=target, loss_node=loss,
                          prediction_node=probs)
    return model


def test_predict_dense_parity():
    """SURVEY 8(f)-3 without MFP: Model.predict_dense (node_basic.py:805-1012) -- block
    tiling, one captured forward graph per stride offset, interleave on the device --
    against the oracle's restatement: float / uint8 input, uint8 output, mirror
    padding, end blocks that need zero padding, a single-slice (ch, x, y) image."""
    spec = [(4, (1, 3, 3), (1, 2, 2), 'relu'), (6, (3, 3, 3), (2, 1, 1), 'relu'),
            (2, (1, 1, 1), (1, 1, 1), 'lin')]
    params = O.init_net(spec, 1, seed=3)
    patch = (6, 16, 16)
    model = _dense_model(spec, params, patch)
    assert list(model.prediction_node.shape.strides) == [2, 2, 2]
    assert list(model.prediction_node.shape.offsets) == [2, 4, 4]
    rng = np.random.RandomState(0)
    raw = rng.rand(1, 11, 23, 25).astype(np.float32)
    ref = O.predict_dense(spec, params, raw, patch)
    got = model.predict_dense(raw)
    assert got.shape == ref.shape == (2, 7, 15, 17) and got.dtype == np.float32
    assert np.abs(got - ref).max() < 2e-6
    raw8 = (raw * 255).astype(np.uint8)
    got8 = model.predict_dense(raw8, as_uint8=True)
    ref8 = O.predict_dense(spec, params, raw8, patch, as_uint8=True)
    assert got8.dtype == np.uint8 and np.abs(got8.astype(int) - ref8.astype(int)).max() <= 1
    gotp = model.predict_dense(raw, pad_raw=True)
    assert gotp.shape == (2, 11, 23, 25)
    assert np.abs(gotp - O.predict_dense(spec, params, raw, patch, pad_raw=True)).max() < 2e-6
    with pytest.raises(ValueError):
        model.predict_dense(raw[:, :3])                 # smaller than the field of view


def test_predict_dense_baseline_net_spot_check():
    """neuro3d_lite (strides [2,4,4], fov [5,39,39]) on a (1,26,190,190) volume: 32 offset
    passes per block at the BASELINE patch size; spot-checked against the oracle's
    per-voxel field-of-view evaluation, which does not depend on tiling at all."""
    spec = O.NEURO3D_LITE
    params = O.init_net(spec, 1, seed=1)
    model = _dense_model(spec, params, (23, 183, 183))
    rng = np.random.RandomState(1)
    raw = rng.rand(1, 26, 190, 190).astype(np.float32)
    dense = model.predict_dense(raw)
    assert dense.shape == (2, 22, 152, 152)
    assert np.abs(dense.sum(axis=0) - 1).max() < 1e-5
    for pos in [(0, 0, 0), (21, 151, 151), (7, 33, 90), (20, 148, 3), (1, 2, 3), (13, 77, 149)]:
        ref = O.predict_voxel(spec, params, raw, pos)
        assert np.abs(dense[(slice(None),) + pos] - ref).max() < 1e-5, pos


@pytest.mark.parametrize("batch", [2, 3])
def test_batch_larger_than_one(batch):
    """the BASELINE configs train with one sample per step, but the node API takes any
    batch (node_basic.py:1182-1243: batch axis None): loss, every gradient and three Adam
    steps of neuro3d_lite with 2 / 3 samples against the oracle"""
    spec, sp = O.NEURO3D_LITE, (7, 47, 47)
    params = O.init_net(spec, 1, seed=1)
    rng = np.random.RandomState(batch)
    x = rng.rand(batch, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (batch, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    m = build('lite', sp, params)
    loss_ref, grads_ref, _ = O.net_loss_and_grads(spec, params, x, t)
    assert abs(float(m.loss(x, t)) - loss_ref) < TOL * abs(loss_ref)
    flat = []
    for gw, gb in grads_ref:
        flat += [gw, gb]
    for g in m.gradients(x, t):
        cands = [r for r in flat if r.shape == g.shape]
        assert min(rel(g, r) for r in cands) < TOL
    ref_losses, _ = O.net_train_steps(spec, params, x, t, 3)
    for i in range(3):
        loss = float(m.trainingstep(x, t, optimiser='Adam')[0])
        assert abs(loss - ref_losses[i]) < 2 * TOL * abs(ref_losses[i])


def test_plan_tensors_carry_zeroed_slack_and_the_step_uses_the_position_split_wgrad():
    """every tensor of a launch plan is followed by Plan.SLACK zeroed floats -- the promise
    (e2_set_input_slack) under which the weight gradient "MT,NT,9,0,S" lets the last unit of a
    plane run past the plane's end -- and a step with those tilings pinned matches the oracle"""
    from elektronn2_amd import autotune
    spec, sp = O.NEURO3D_LITE, (9, 71, 71)
    params = O.init_net(spec, 1, seed=4)
    rng = np.random.RandomState(6)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    loss_ref, grads_ref, _ = O.net_loss_and_grads(spec, params, x, t)
    m = build('lite', sp, params)
    autotune.force('wgrad', "7,2,9,0,3")        # (layers it does not fit take the library's choice)
    try:
        got = m.gradients(x, t)
        plan = getattr(m._grad_func, 'func', None)
    finally:
        autotune.force('wgrad', None)
    flat_ref = []
    for gw, gb in grads_ref:
        flat_ref += [gw, gb]
    for g in got:
        cands = [r for r in flat_ref if r.shape == g.shape]
        assert min(rel(g, r) for r in cands) < TOL
    from elektronn2_amd.neuromancer.plan import Plan
    assert plan is not None and plan.out
    plans = [plan]
    for p in plans:
        for node, buf in p.out.items():
            if buf is None or not buf.is_contiguous() or node in p.inputs:
                continue              # (inputs are slices of ONE arena: the slack sits behind the arena)
            n = buf.numel()
            st = buf.untyped_storage().nbytes() // 4 - buf.storage_offset()
            assert st >= n + Plan.SLACK or buf.storage_offset() > 0, node.name
            if buf.storage_offset() == 0 and st >= n + Plan.SLACK:
                tail = torch.as_strided(buf, (Plan.SLACK,), (1,), n)
                assert float(tail.abs().max()) == 0.0, node.name
        n = p.input_arena.numel()
        assert float(torch.as_strided(p.input_arena, (Plan.SLACK,), (1,), n).abs().max()) == 0.0


def test_several_steps_in_one_graph_launch():
    """Model.trainingsteps / Plan.run_steps (DESIGN finding 55): k steps captured into ONE graph,
    the batches taken out of a device-side ring by the graph itself, the losses kept in a
    device-side history.  Nine steps over a ring of four different batches -- as a run of single
    steps fed through trainingstep, and as trainingstep + trainingsteps(5) + trainingsteps(3)
    (the first of which still contains the single step's capture) -- give the same losses step by
    step and the same parameters (to the weight gradients' atomic order: finding 53)."""
    spec, sp = O.NEURO3D_LITE, (7, 47, 47)
    params = O.init_net(spec, 1, seed=3)
    rng = np.random.RandomState(8)
    osp = O.net_out_shape(spec, sp)
    xs = [rng.rand(1, 1, *sp).astype(np.float32) for _ in range(4)]
    ts = [rng.randint(0, 2, (1, 1) + osp).astype(np.float32) for _ in range(4)]
    a = build('lite', sp, params)
    la = [float(a.trainingstep(xs[i % 4], ts[i % 4], optimiser='Adam')[0]) for i in range(9)]
    pa = [p.get_value() for p in a.loss_node.all_trainable_params.values()]

    b = build('lite', sp, params)
    lb = [float(b.trainingstep(xs[0], ts[0], optimiser='Adam')[0])]           # builds the plan (eager)
    plan = b.optimisers['Adam'].step.func
    ring = torch.zeros(4, plan.input_arena.numel(), device=plan.ctx.device)
    for j in range(4):
        for node, src in zip(plan.inputs[:2], (xs[j], ts[j])):
            o, n = plan.input_slices[node]
            ring[j, o:o + n] = torch.tensor(src.ravel(), device=ring.device)
    # the ring's next slot is the one of step 1: start the count there
    plan.set_input_ring(ring)
    plan._step_state[0] += 1
    assert plan.ring_position() == 1
    l5, t5 = b.trainingsteps(5, optimiser='Adam', ring=ring)
    assert len(l5) == 5 and plan._multi, "no multi-step graph was captured"
    k_cap = sorted(plan._multi)
    l3, t3 = b.trainingsteps(3, optimiser='Adam', ring=ring)
    assert len(l3) == 3 and 3 in plan._multi and t3 > 0
    lb += [float(v) for v in l5] + [float(v) for v in l3]
    assert plan.ring_position() == 9 and b.iterations == 9
    for i, (u, v) in enumerate(zip(la, lb)):
        assert abs(u - v) < 1e-5 * abs(u), (i, la, lb, k_cap)
    for u, v in zip(pa, [p.get_value() for p in b.loss_node.all_trainable_params.values()]):
        assert rel(v, u) < 1e-4
    # a second launch of the SAME 3-step graph goes on counting (slots 1, 2, 3)
    l3b, _ = b.trainingsteps(3, optimiser='Adam')
    assert plan.ring_position() == 12
    more = [float(a.trainingstep(xs[i % 4], ts[i % 4], optimiser='Adam')[0]) for i in range(9, 12)]
    for u, v in zip(more, l3b):
        assert abs(u - float(v)) < 1e-5 * abs(u)
    # detaching the ring returns to set_inputs
    plan.set_input_ring(None)
    one = float(b.trainingstep(xs[0], ts[0], optimiser='Adam')[0])
    ref = float(a.trainingstep(xs[0], ts[0], optimiser='Adam')[0])
    assert abs(one - ref) < 1e-5 * abs(ref)


def test_weight_gradients_on_the_side_stream_in_f32_mode():
    """Per-problem side-stream flags (autotune.side_flag / plan option side_mask; DESIGN finding
    56): in f32 mode the plan keeps one stream, except for the weight gradients the tuning table
    marks -- their launches form a side branch of the captured graph, issued behind the main
    stream's next launches (side_defer).  Any choice of layers gives the losses and parameters of
    the single-stream plan (to the weight gradients' atomic order); eager, capture, replay."""
    from elektronn2_amd.neuromancer import plan_options
    spec, sp = O.NEURO3D, (17, 109, 109)
    params = O.init_net(spec, 1, seed=7)
    rng = np.random.RandomState(9)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    res = {}
    for mask in (0, 465, 511, 2):
        with plan_options(side_mask=mask, side_table=False):
            m = build('full', sp, params)
            losses = [float(m.trainingstep(x, t, optimiser='Adam')[0]) for _ in range(4)]
            plan = m.optimisers['Adam'].step.func
            assert not plan.use_side
            forced = [n.name for n in plan._side_order or [] if plan.side_forced(n)]
            assert len(forced) == bin(mask).count("1"), (mask, forced)
            res[mask] = (losses, [q.get_value() for q in m.loss_node.all_trainable_params.values()])
    for mask in (465, 511, 2):
        for a, b in zip(res[mask][0], res[0][0]):
            assert abs(a - b) < 1e-5 * abs(b), (mask, res[mask][0], res[0][0])
        for a, b in zip(res[mask][1], res[0][1]):
            assert rel(a, b) < 1e-4
    # the shipped table's flags are honoured when no mask is given -- and ignored in bf16 mode
    from elektronn2_amd import autotune
    m = build('full', (23, 185, 185), O.init_net(spec, 1, seed=7))
    plan = m.optimisers['Adam'].step
    plan.compile()
    plan = plan.func
    plan.set_inputs([np.zeros((1, 1, 23, 185, 185), np.float32),
                     np.zeros((1, 1) + O.net_out_shape(spec, (23, 185, 185)), np.float32)])
    plan.side_rank(None)
    flagged = [n.name for n in plan._side_order if plan.side_forced(n)]
    assert flagged == [n.name for n in plan._side_order if autotune.side_flag(plan.ctx, n._sig_wgrad(plan))]
    assert len(flagged) >= 3, flagged

