"""CPU: the neuromancer-shaped front end -- shape / stride / fov bookkeeping,
naming, parameter protocol, error behaviour -- against the numbers and rules
recorded from the reference (neural.py:725-764)."""
import numpy as np
import pytest

from elektronn2_amd import nets, neuromancer as nm


@pytest.fixture(autouse=True)
def fresh():
    nm.model_manager.reset()
    yield
    nm.model_manager.reset()


def test_neuro3d_lite_at_183_shapes_and_counts():
    m = nets.neuro3d_lite()
    pn = m.prediction_node
    assert pn.shape.shape == [None, 2, 10, 37, 37]
    assert list(pn.shape.strides) == [2, 4, 4]
    assert pn.shape.fov == [5, 39, 39] and pn.shape.offsets == [2, 19, 19]
    assert m.loss_node.all_params_count == 885132
    assert list(m.nodes.keys())[:9] == ['raw', 'conv', 'conv1', 'conv2', 'conv3', 'conv4',
                                       'conv5', 'conv6', 'softmax']
    assert m.target_node.shape.shape == [None, 1, 10, 37, 37]
    assert m.target_node.output.dtype == 'float32'       # cnndata.py:140 reads this
    assert m.batch_size is None and m.ndim == 3
    # comp cost = prod(w_sh) * n_positions * b  (neural.py:767-778)
    assert m.nodes['conv1'].computational_cost == 40 * 20 * 27 * 21 * 88 * 88


def test_docs_toy_net_known_answers():
    """The ONE printed known-answer block the reference holds for `_calc_shape` /
    `_calc_comp_cost` / parameter counting: docs/examples.rst:49-94 builds a 3-layer toy net
    on (10,3,23,183,183) and docs/examples.rst:103-140 prints, per node, #Params, Comp.Cost
    and the output shape, then fov / offsets / strides of the prediction, the total cost,
    the parameter count and the cost per output pixel.  Same constructor calls here."""
    image = nm.Input((10, 3, 23, 183, 183), 'b,f,z,x,y', name='image')
    conv0 = nm.Conv(image, 32, (1, 6, 6), (1, 2, 2))
    conv1 = nm.Conv(conv0, 64, (4, 6, 6), (2, 2, 2))
    conv2 = nm.Conv(conv1, 5, (3, 3, 3), (1, 1, 1), activation_func='lin')
    class_probs = nm.Softmax(conv2)
    target = nm.Input_like(class_probs, override_f=1, name='target', dtype='int16')
    voxel_loss = nm.MultinoulliNLL(class_probs, target, target_is_sparse=True)
    scalar_loss = nm.AggregateLoss(voxel_loss, name='loss')
    errors = nm.Errors(class_probs, target, target_is_sparse=True)
    model = nm.model_manager.getmodel()
    model.designate_nodes(input_node=image, target_node=target, loss_node=scalar_loss,
                          prediction_node=class_probs,
                          prediction_ext=[scalar_loss, errors, class_probs])
    # default names and enumeration (docs: 'conv', 'conv1', 'conv2', 'softmax', 'nll', 'errors')
    assert [conv0.name, conv1.name, conv2.name, class_probs.name, voxel_loss.name,
            scalar_loss.name, errors.name] == ['conv', 'conv1', 'conv2', 'softmax', 'nll',
                                               'loss', 'errors']
    # per node: #Params, output shape (docs/examples.rst:106-116)
    assert [n.param_count for n in (conv0, conv1, conv2)] == [3488, 294976, 8645]
    assert conv0.shape.shape == [10, 32, 23, 89, 89]
    assert conv1.shape.shape == [10, 64, 10, 42, 42]
    assert conv2.shape.shape == [10, 5, 8, 40, 40]
    assert class_probs.shape.shape == [10, 5, 8, 40, 40]
    assert target.shape.shape == [10, 1, 8, 40, 40] and target.output.dtype == 'int16'
    assert voxel_loss.shape.shape == [10, 1, 8, 40, 40]
    # Comp.Cost as printed: 25.2 / 416.2 / 1.1 Giga, 640.0 kilo (softmax, nll), 128.0 kilo
    # (loss, errors) -- the integers behind them (neural.py:767-778)
    assert conv0.computational_cost == 32 * 3 * 36 * 23 * 178 * 178 * 10
    assert conv1.computational_cost == 64 * 32 * 144 * 20 * 84 * 84 * 10
    assert conv2.computational_cost == 5 * 64 * 27 * 8 * 40 * 40 * 10
    giga = lambda v: round(v / 1e9, 1)
    assert [giga(n.computational_cost) for n in (conv0, conv1, conv2)] == [25.2, 416.2, 1.1]
    assert class_probs.computational_cost == 640000 and voxel_loss.computational_cost == 640000
    assert scalar_loss.computational_cost == 128000 and errors.computational_cost == 128000
    # prediction properties (docs/examples.rst:134-136)
    pn = model.prediction_node
    assert pn.shape.fov == [9, 27, 27] and pn.shape.offsets == [4, 13, 13]
    assert list(pn.shape.strides) == [2, 4, 4] and list(pn.shape.spatial_shape) == [8, 40, 40]
    # totals (docs/examples.rst:137-139; model.py:154-165 takes them from the prediction node)
    n_comp = pn.all_computational_cost
    assert giga(n_comp) == 442.5
    assert pn.all_params_count == 307109 and scalar_loss.all_params_count == 307109
    assert round(n_comp / float(pn.shape.spatial_size) / 1e6, 1) == 34.6
    assert model.batch_size == 10 and model.ndim == 3


def test_neuro3d_needs_185_and_rejects_183():
    m = nets.neuro3d()
    assert m.prediction_node.shape.shape == [None, 2, 5, 21, 21]
    assert m.prediction_node.shape.fov == [15, 105, 105]
    assert m.loss_node.all_params_count == 2756042
    nm.model_manager.reset()
    with pytest.raises(ValueError, match="Cannot pool spatial axis"):
        nets.neuro3d((None, 1, 23, 183, 183))


def test_param_init_formulas_and_protocol():
    np.random.seed(0)
    inp = nm.Input((1, 4, 9, 20, 20), 'b,f,z,x,y')
    c = nm.Conv(inp, 200, (2, 3, 3), (2, 1, 1))
    w = c.w.get_value()
    # glorot normal: std = sqrt(2 / ((n_in + n_out/prod(pool)) * prod(kernel)))
    assert abs(w.std() - np.sqrt(2.0 / ((4 + 200 / 2.0) * 18))) / w.std() < 0.05
    assert np.allclose(c.b.get_value(), 1.0 / 18)          # relu bias = 1/prod(kernel)
    assert c.w.apply_reg is True and c.b.apply_reg is False and c.w.apply_train
    lin = nm.Conv(c, 2, (1, 1, 1), activation_func='lin')
    assert np.abs(lin.b.get_value()).max() <= 1e-6
    with pytest.raises(NotImplementedError):
        c.w.set_value(np.zeros((3, 3)))                    # variables.py:137-146
    c.w.set_value(np.ones(w.shape))                        # float64 -> downcast
    assert c.w.get_value().dtype == np.float32
    wi = np.random.rand(7, 200, 1, 1, 1).astype(np.float32)
    c2 = nm.Conv(c, 7, (1, 1, 1), w=wi, b=np.zeros(7, np.float32))
    assert np.array_equal(c2.w.get_value(), wi)
    with pytest.raises(ValueError, match="Shape mismatch"):
        nm.Conv(c, 7, (1, 1, 1), w=np.zeros((7, 3, 1, 1, 1), np.float32))


def test_uni_and_ortho_init_modes():
    """variables.py:205-266: 'uni' and 'ortho' (config.use_ortho_init) next to 'normal'."""
    from elektronn2_amd.neuromancer.variables import initweights
    from elektronn2_amd import config
    kw = dict(scale='glorot', pool=(2, 1, 1), spatial_axes=[2, 3, 4])
    sh = (40, 6, 2, 3, 3)
    std = np.sqrt(2.0 / ((6 + 40 / 2.0) * 18))
    np.random.seed(1)
    u = initweights(sh, mode='uni', **kw)
    assert u.shape == sh and u.dtype == np.float32 and abs(u).max() <= std
    assert abs(u.std() - std / np.sqrt(3)) / u.std() < 0.05
    np.random.seed(1)
    o = initweights(sh, mode='ortho', **kw).reshape(40, -1)           # 40 rows of 108
    g = o @ o.T
    assert np.abs(g - np.diag(np.diag(g))).max() < 1e-4 * np.diag(g).min()   # orthogonal rows
    assert np.allclose(o.std(axis=1), std, rtol=1e-4)                 # each at the glorot std
    # RNG stream: the first draw has the tensor's shape, whatever follows
    np.random.seed(1); np.random.normal(0, std, size=sh); nxt = np.random.rand()
    np.random.seed(1); initweights(sh, mode='ortho', **kw)
    assert np.random.rand() == nxt
    tall = initweights((30, 2, 1, 3, 3), mode='ortho', scale='glorot', pool=(1, 1, 1),
                       spatial_axes=[2, 3, 4])                        # more rows than columns
    assert tall.shape == (30, 2, 1, 3, 3) and np.isfinite(tall).all()
    config.use_ortho_init = True
    try:
        nm.model_manager.reset()
        c = nm.Conv(nm.Input((1, 4, 9, 20, 20), 'b,f,z,x,y'), 8, (2, 3, 3))
        w = c.w.get_value().reshape(8, -1)
        g = w @ w.T
        assert np.abs(g - np.diag(np.diag(g))).max() < 1e-4 * np.diag(g).min()
    finally:
        config.use_ortho_init = False
    with pytest.raises(NotImplementedError):
        initweights(sh, mode='prelu', **kw)


def test_error_behaviour_matches_reference():
    inp = nm.Input((1, 1, 8, 20, 20), 'b,f,z,x,y')
    with pytest.raises(ValueError, match="dimensionality"):
        nm.Conv(inp, 4, (3, 3))
    with pytest.raises(ValueError, match="Cannot pool"):
        nm.Conv(inp, 4, (1, 3, 3), (1, 4, 4))
    c = nm.Conv(inp, 4, (1, 3, 3))
    with pytest.raises(ValueError, match="linear activation"):
        nm.Softmax(c)
    with pytest.raises(NotImplementedError):
        nm.Pool(c, (1, 2, 2), stride=(1, 1, 1))
    with pytest.raises(ValueError, match="Cannot downsample"):
        nm.Pool(c, (1, 4, 4))
    with pytest.raises(NotImplementedError):
        nm.Conv(inp, 4, (1, 3, 3), batch_normalisation='fadeout')
    with pytest.raises(ValueError, match="Unknown value"):
        nm.Conv(inp, 4, (1, 3, 3), batch_normalisation='yes')
    with pytest.raises(ValueError, match="Cannot pass mean and std"):
        nm.Conv(inp, 4, (1, 3, 3), batch_normalisation='train', mean=np.zeros(4, np.float32))
    inp1 = nm.Input((1, 1, 20), 'b,f,x')
    with pytest.raises(NotImplementedError):
        nm.Conv(inp1, 4, (3,))
    with pytest.raises(NotImplementedError):                  # per-voxel Perceptron = 1x1x1 Conv
        nm.Perceptron(c, 5)


def test_config1_mnist_graph():
    """examples/mnist.py:29-56: shapes, tags, parameters and their training flags"""
    from elektronn2_amd import nets
    nm.model_manager.reset()
    np.random.seed(0)
    m = nets.mnist()
    sh = {n.name: (list(n.shape.shape), list(n.shape.tags)) for n in m.nodes.values()}
    assert sh['conv'] == ([None, 12, 12, 12], ['b', 'f', 'y', 'x'])
    assert sh['conv1'][0] == [None, 36, 5, 5] and sh['conv2'][0] == [None, 64, 3, 3]
    assert sh['dot'] == ([None, 200], ['b', 'f']) and sh['dot1'][0] == [None, 10]
    assert sh['target'] == ([None, 1], ['b', 'f'])
    assert list(m.nodes['conv2'].shape.strides) == [4, 4]
    tp = m.loss_node.all_trainable_params
    assert list(tp) == ['conv_w', 'conv_b', 'conv_gamma', 'conv1_w', 'conv1_b', 'conv1_gamma',
                        'conv2_w', 'conv2_b', 'conv2_gamma', 'dot_w', 'dot_b', 'dot1_w', 'dot1_b']
    assert tp['dot_w'].shape == (576, 200) and tp['conv1_w'].shape == (36, 12, 3, 3)
    assert tp['conv_gamma'].apply_reg == 3.0 and tp['conv_b'].apply_reg is False
    c = m.nodes['conv']
    assert not c.mean.apply_train and not c.std.apply_train
    assert np.all(c.gamma.get_value() == 1) and np.all(c.std.get_value() == 1) \
        and np.all(c.mean.get_value() == 0)
    # relu bias = 1 / prod(kernel) (neural.py:173-181); Perceptron: 1.0; 'lin': U(+-1e-6)
    assert np.allclose(c.b.get_value(), 1.0 / 9) and np.allclose(m.nodes['dot'].b.get_value(), 1.0)
    assert np.abs(m.nodes['dot1'].b.get_value()).max() <= 1e-6
    # glorot (variables.py:231-246): std = sqrt(2 / (n_in + n_out)) for the dot layers
    w = m.nodes['dot'].w.get_value()
    assert abs(w.std() - np.sqrt(2.0 / (576 + 200))) < 2e-3


def test_upconvmerge_unet_bookkeeping():
    """examples/unet3d_lite.py pattern: UpConv inserted for stride ratio > 1,
    symmetric Crop of the high-res branch, Concat((lo_res upconv, hi_res))."""
    inp = nm.Input((1, 1, 12, 36, 36), 'b,f,z,x,y', name='raw')
    c1 = nm.Conv(nm.Conv(inp, 8, (1, 3, 3)), 8, (1, 3, 3))
    p1 = nm.Pool(c1, (1, 2, 2))
    c3 = nm.Conv(nm.Conv(p1, 16, (3, 3, 3)), 16, (3, 3, 3))
    assert list(c3.shape.strides) == [1, 2, 2]
    mrg = nm.UpConvMerge(c1, c3, 24)
    assert isinstance(mrg, nm.Concat)
    lo, hi = mrg.parent
    assert isinstance(lo, nm.UpConv) and isinstance(hi, nm.Crop)
    assert lo.pool_shape == (1, 2, 2) and lo.shape['f'] == 24
    assert lo.shape.spatial_shape == hi.shape.spatial_shape == [8, 24, 24]
    assert hi.crop == [2, 4, 4]
    assert mrg.shape['f'] == 24 + 8
    assert lo.shape.fov == [-1, -1, -1]
    # identity_init (neural.py:977-986)
    w = lo.w.get_value()
    assert np.all(w[np.arange(16), np.arange(16)] == 1.0) and np.all(lo.b.get_value() == 0)
    # designate_nodes repairs the fov of UpConv nets (model.py:141-152)
    out = nm.Conv(nm.Conv(mrg, 8, (1, 3, 3)), 2, (1, 1, 1), activation_func='lin')
    probs = nm.Softmax(out)
    target = nm.Input_like(probs, override_f=1, name='target')
    loss = nm.AggregateLoss(nm.MultinoulliNLL(probs, target, target_is_sparse=True))
    model = nm.model_manager.getmodel()
    model.designate_nodes(input_node=inp, target_node=target, loss_node=loss,
                          prediction_node=probs)
    assert probs.shape.fov == [4, 14, 14] and target.shape.fov == [4, 14, 14]
    assert set(model.optimisers) == {'SGD', 'Adam'}


def test_optimiser_globals_are_shared():
    m = nets.neuro3d_lite((None, 1, 7, 47, 47))
    m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    assert abs(m.lr - 5e-4) < 1e-10 and abs(m.wd - 0.5e-4) < 1e-12
    m.lr = 1e-3
    assert abs(m.optimisers['SGD'].global_lr.get_value() - 1e-3) < 1e-10   # optimiser.py:19-29
    with pytest.raises(AttributeError):
        m.set_opt_meta_params('SGD', dict(beta2=0.5))


def test_choose_name_and_call_arity():
    assert nm.choose_name('conv', ['conv']) == 'conv1'
    assert nm.choose_name('conv', ['conv', 'conv1', 'conv2']) == 'conv3'
    m = nets.neuro3d_lite((None, 1, 7, 47, 47))
    x = np.zeros((1, 1, 7, 47, 47), np.float32)
    with pytest.raises(TypeError, match="inputs required"):
        m.prediction_node(x, x)          # (a call without arguments only compiles)


def test_tiling_candidate_lists():
    """the tuner's candidate strings are what e2_set_tiling accepts (csrc/api.hip) and what the
    kernels have instances for; host logic only -- the shipped choices must be among them"""
    import json
    import os
    import re
    from elektronn2_amd import autotune
    # 1x1x1 GEMM with LDS-staged weights: "1,MT,NT", only for 1x1x1 kernels with enough channels
    pw = autotune.pointwise_candidates(200, (1, 1, 1))
    assert pw and all(re.fullmatch(r"1,(4|5|6|7|8|10|13|16),(1|2)(,(64|128),0)?", c) for c in pw)
    assert "1,13,1,64,0" in pw and not any(c.startswith("1,13,2,128") for c in pw)
    assert all("," not in c[6:] for c in autotune.pointwise_candidates(200, (1, 1, 1), cin=20))
    assert "1,13,1" in pw and "1,7,2" in pw
    assert autotune.pointwise_candidates(200, (1, 3, 3)) == []
    assert autotune.pointwise_candidates(2, (1, 1, 1)) == []
    assert set(pw) <= set(autotune.igemm_candidates(200, 200, (1, 1, 1), (10, 37, 37)))
    # 1x1x1 / UpConv weight gradient as a GEMM with K-contiguous operands: "MT,NT,7,0,S"; the
    # tile list is the kernel's instance list (csrc/conv_pw_wgrad.hip)
    src = open(os.path.join(os.path.dirname(autotune.__file__), "csrc", "conv_pw_wgrad.hip")).read()
    host7, host8 = src.split("int e2i_pw_wgrad_ks(")
    inst = set((int(a), int(b)) for a, b in re.findall(r"E2_L\((\d+), (\d+)\)", host7))
    assert inst == set(autotune.PW_WGRAD_TILES)
    # ... and "MT,NT,8,0,S": one tile per work-group, its four waves split the positions
    host8, host9 = host8.split("int e2i_wgrad_ks(")
    inst8 = set((int(a), int(b)) for a, b in re.findall(r"E2_L\((\d+), (\d+)\)", host8))
    assert inst8 == set(autotune.PW_WGRAD_KS_TILES)
    # ... and "MT,NT,9,0,S": the same GEMM for kernels WITH taps (needs (kh - 1) input rows >= 31
    # zeros behind a gradient plane: not offered for short rows)
    inst9 = set((int(a), int(b)) for a, b in re.findall(r"E2_L\((\d+), (\d+)\)", host9))
    assert inst9 == set(autotune.WGRAD_KS_TILES)
    k9 = autotune.position_split_wgrad_candidates(200, 200, (1, 3, 3), (10, 37, 37))
    assert k9 and all(re.fullmatch(r"\d+,\d+,9,0,\d+", c) for c in k9) and "7,2,9,0,4" in k9
    assert set(k9) <= set(autotune.wgrad_candidates(200, 200, (1, 3, 3), (10, 37, 37)))
    assert autotune.position_split_wgrad_candidates(200, 200, (1, 1, 1), (10, 37, 37)) == []
    assert autotune.position_split_wgrad_candidates(200, 200, (1, 2, 2), (10, 9, 9)) == []    # 10 + 1 < 31 zeros
    pg = autotune.pointwise_wgrad_candidates(2048, 256, (1, 1, 1), (18, 10, 10))
    assert pg and all(re.fullmatch(r"\d+,\d+,[78],0,\d+", c) for c in pg)
    ks = [c for c in autotune.pointwise_wgrad_candidates(200, 200, (1, 1, 1), (10, 37, 37)) if ",8,0," in c]
    assert "7,2,8,0,18" in ks and "13,2,8,0,36" in ks       # (14 / 7 tiles x S = 252 work-groups)
    assert all((int(c.split(",")[0]), int(c.split(",")[1])) in (inst8 if ",8,0," in c else inst) for c in pg)
    assert autotune.pointwise_wgrad_candidates(200, 200, (1, 3, 3), (10, 37, 37)) == []
    assert set(pg) <= set(autotune.wgrad_candidates(2048, 256, (1, 1, 1), (18, 10, 10)))
    # bf16 weight gradient: "32,MB,NB,R,S"; the row form (R = 1) for few input channels only,
    # the column form (R = 0) when at least 48
    for cin, forms in ((20, {"1"}), (40, {"1"}), (100, {"0", "1"}), (200, {"0"}), (8, set())):
        c = autotune.bf16_wgrad_candidates(cin, (1, 3, 3))
        assert {x.split(",")[3] for x in c} == forms, (cin, c)
        assert all(re.fullmatch(r"32,[12],[1-3],[01],(0|8|16)", x) for x in c)
    assert max(int(x.split(",")[2]) for x in autotune.bf16_wgrad_candidates(64, (2, 4, 4))) == 4
    # every shipped choice is a well-formed string of one of the known forms
    shipped = json.load(open(os.path.join(os.path.dirname(autotune.__file__), "tuned.json")))
    forms = {"igemm": r"\d+,\d+,\d+,\d+|4(,\d+){7}|1,\d+,[12](,(32|64|128),0)?|32,\d+,\d+",
             "wgrad": r"\d+,\d+,\d+,\d+,\d+",
             "side": r"1"}         # f32: this weight gradient runs on the side stream (tools/tune_side.py)
    for key, val in shipped.items():
        kind = key.split("|")[0].replace("_bf16", "")
        assert val == "" or re.fullmatch(forms[kind], val), (key, val)
    # a side flag belongs to a weight-gradient problem that has a shipped tiling too
    for key in shipped:
        if key.startswith("side|"):
            assert "wgrad|" + key.split("|", 1)[1] in shipped, key


def test_per_kernel_regression_gate_on_the_committed_profiles():
    """tools/kstats_diff.py: run on round 3's own profiles it names the
    kernels that regressed between r03_a and r03_c -- the first-layer backward (+16 us, run-time
    debug branches) among them -- and it passes a profile against itself; adopt_profile.py
    imports the same function."""
    import io
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import kstats_diff
    prof = os.path.join(root, "profiles")
    a = os.path.join(prof, "r03_a_bench_lite183_kernel_stats.csv")
    c = os.path.join(prof, "r03_c_bench_lite183_kernel_stats.csv")
    out = io.StringIO()
    bad = kstats_diff.diff(c, a, out=out)
    assert "firstm_bwd_kernel<4, 4, 2, 2, 5>" in bad and "SUM" in bad
    assert "REGRESSION" in out.getvalue()
    assert kstats_diff.diff(c, c, out=io.StringIO()) == []
    # the profile of record exists for both nets of the driver's line and names real files
    import json
    cur = json.load(open(os.path.join(prof, "CURRENT.json")))
    for wl in ("lite183", "full185"):
        assert os.path.exists(kstats_diff.record_path(wl)), wl
        base = os.path.join(prof, "%s_bench_%s" % (cur[wl]["tag"], wl))
        for suf in ("_pmc_mfma.csv", "_pmc_traffic.csv", ".json"):
            assert os.path.exists(base + suf), base + suf


def test_build_check_enforces_the_register_budget(tmp_path):
    """csrc/check_scratch.py + reg_budget.txt: a kernel that uses scratch, or falls below the
    occupancy its launch geometry assumes, fails the BUILD (round 3 shipped a first-layer
    backward at occupancy 2 under a 3-per-CU persistent grid)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "elektronn2_amd", "csrc")

    def log(name, vgprs, scratch, occ):
        return ("x.hip:1:1: remark: Function Name: %s [-Rpass-analysis=kernel-resource-usage]\n"
                "x.hip:1:1: remark:     VGPRs: %d [-Rpass-analysis=kernel-resource-usage]\n"
                "x.hip:1:1: remark:     ScratchSize [bytes/lane]: %d [-Rpass-analysis=kernel-resource-usage]\n"
                "x.hip:1:1: remark:     Occupancy [waves/SIMD]: %d [-Rpass-analysis=kernel-resource-usage]\n"
                % (name, vgprs, scratch, occ))
    fm = "_ZN12_GLOBAL__N_117firstm_bwd_kernelILi4ELi4ELi2ELi2ELi5EEEvNS_6FirstMEPf"
    ig = "_Z12igemm_kernelILi7ELi2ELi3ELi1ELb0EEv6IgemmP"
    cases = [(log(fm, 103, 0, 3) + log(ig, 125, 0, 4), 0),
             (log(fm, 128, 0, 2), 1),                     # round 3's regression
             (log(fm, 110, 0, 3), 1),                     # above the VGPR budget
             (log(ig, 125, 8, 4), 1),                     # scratch in a hand-scheduled GEMM
             (log(ig, 150, 0, 3), 1)]                     # no second co-resident work-group
    for i, (text, want) in enumerate(cases):
        f = tmp_path / ("k%d.log" % i)
        f.write_text(text)
        r = subprocess.run([sys.executable, os.path.join(csrc, "check_scratch.py"), str(f),
                            "igemm4_kernel|wgrad_direct_kernel|igemm_kernel|pw_gemm_kernel",
                            os.path.join(csrc, "reg_budget.txt")], capture_output=True, text=True)
        assert (r.returncode != 0) == bool(want), (i, r.stderr)


def test_plan_options_are_named_arguments_with_environment_defaults(monkeypatch):
    """neuromancer/options.py: every host switch of the launch plan is a named
    option -- set process-wide, inside a context manager, or per plan; the E2_* variable is the
    DEFAULT only; unknown names are errors; tools/kstats_diff.py takes the number of steps in a
    profile from the optimiser launch's call count."""
    from elektronn2_amd.neuromancer import options as opt
    assert opt.get("graph") is True and opt.get("side_defer") is True and opt.get("adam_pack") is False
    monkeypatch.setenv("E2_NO_GRAPH", "1")
    monkeypatch.setenv("E2_BF16_AHEAD_MIN", "0.5")
    assert opt.get("graph") is False and opt.get("bf16_ahead_min") == 0.5
    nm.set_plan_options(graph=True)                       # an explicit value beats the environment
    try:
        assert opt.get("graph") is True
        with nm.plan_options(graph=False, fuse_tail=False):
            snap = opt.snapshot()
            assert snap["graph"] is False and snap["fuse_tail"] is False
        assert opt.get("graph") is True and opt.get("fuse_tail") is True
        assert opt.snapshot({"dp_overlap": False})["dp_overlap"] is False
        with pytest.raises(KeyError):
            nm.set_plan_options(no_such_switch=1)
        with pytest.raises(KeyError):
            opt.snapshot({"no_such_switch": 1})
        nm.set_plan_options(side_stream=True)
        assert opt.get("side_stream") is True
        nm.set_plan_options(reset=("side_stream",))
        assert opt.get("side_stream") is None
    finally:
        nm.set_plan_options(graph=None)
    assert opt.get("graph") is False                      # back to the environment's default
    # every option documents itself and names its variable exactly once
    envs = [v[0] for v in opt.SPEC.values()]
    assert len(set(envs)) == len(envs) and all(e.startswith("E2_") for e in envs)
    assert all(len(v) == 4 and len(v[3]) > 10 for v in opt.SPEC.values())


def test_kstats_diff_counts_steps_by_the_optimiser_launch(tmp_path):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import kstats_diff
    finally:
        sys.path.pop(0)
    head = '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'

    def write(name, steps, gemm_us):
        f = tmp_path / name
        f.write_text(head + '"adam_kernel(float*)",%d,%d,10000,1,1,1,0\n' % (steps, steps * 10000)
                     + '"void igemm_kernel<7, 2, 3, 1, false>(IgemmP)",%d,%d,1,1,1,1,0\n'
                     % (3 * steps, int(3 * steps * gemm_us * 1000)))
        return str(f)
    import io
    old = write("old.csv", 25, 100.0)
    assert kstats_diff.steps_of(old, 7) == 25
    same = write("same.csv", 37, 100.0)                   # more steps in the file, same kernels
    assert kstats_diff.diff(same, old, out=io.StringIO()) == []
    slow = write("slow.csv", 37, 104.0)                   # + 12 us per step, + 4 %
    assert kstats_diff.diff(slow, old, out=io.StringIO()) == ["igemm_kernel<7, 2, 3, 1, false>", "SUM"]
