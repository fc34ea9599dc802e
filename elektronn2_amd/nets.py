"""The reference example networks, built through the neuromancer-shaped API.
``create_model`` bodies follow examples/neuro3d_lite.py:46-75,
examples/neuro3d.py:46-76, examples/unet3d_lite.py:59-118 line by line (only
the import differs)."""
from __future__ import annotations

import numpy as np


def neuro3d_lite(in_sh=(None, 1, 23, 183, 183), params=None, name=None, mfp=False):
    """``mfp=True``: the prediction-time rewrite with max-fragment pooling (batch 1, input
    extents shifted so that every pooled axis satisfies the MFP rule) ending in
    ``FragmentsToDense`` -- one pass predicts a DENSE block."""
    from . import neuromancer as nm
    if name is not None:
        nm.model_manager.newmodel(name)
    P = _pget(params)
    inp = nm.Input(in_sh, 'b,f,z,x,y', name='raw')
    out = nm.Conv(inp, 20, (1, 4, 4), (1, 2, 2), mfp=mfp, **P(0))
    out = nm.Conv(out, 40, (3, 3, 3), (1, 2, 2), mfp=mfp, **P(1))
    out = nm.Conv(out, 150, (2, 4, 4), (2, 1, 1), mfp=mfp, **P(2))
    out = nm.Conv(out, 200, (1, 3, 3), mfp=mfp, **P(3))
    out = nm.Conv(out, 200, (1, 3, 3), mfp=mfp, **P(4))
    out = nm.Conv(out, 200, (1, 1, 1), mfp=mfp, **P(5))
    out = nm.Conv(out, 2, (1, 1, 1), activation_func='lin', mfp=mfp, **P(6))
    return _finish(nm, inp, out, name, mfp)


def neuro3d(in_sh=(None, 1, 23, 185, 185), params=None, name=None, mfp=False):
    from . import neuromancer as nm
    if name is not None:
        nm.model_manager.newmodel(name)
    P = _pget(params)
    inp = nm.Input(in_sh, 'b,f,z,x,y', name='raw')
    out = nm.Conv(inp, 20, (1, 6, 6), (1, 2, 2), mfp=mfp, **P(0))
    out = nm.Conv(out, 30, (1, 5, 5), (1, 2, 2), mfp=mfp, **P(1))
    out = nm.Conv(out, 40, (1, 5, 5), mfp=mfp, **P(2))
    out = nm.Conv(out, 80, (4, 4, 4), (2, 1, 1), mfp=mfp, **P(3))
    out = nm.Conv(out, 100, (3, 4, 4), mfp=mfp, **P(4))
    out = nm.Conv(out, 100, (3, 4, 4), mfp=mfp, **P(5))
    out = nm.Conv(out, 150, (2, 4, 4), mfp=mfp, **P(6))
    out = nm.Conv(out, 200, (1, 4, 4), mfp=mfp, **P(7))
    out = nm.Conv(out, 200, (1, 4, 4), mfp=mfp, **P(8))
    out = nm.Conv(out, 200, (1, 1, 1), mfp=mfp, **P(9))
    out = nm.Conv(out, 2, (1, 1, 1), activation_func='lin', mfp=mfp, **P(10))
    return _finish(nm, inp, out, name, mfp)


def unet3d_lite(in_sh=(None, 1, 22, 140, 140), name=None):
    """examples/unet3d_lite.py:59-118: three (1,2,2) poolings, two convs per level,
    three UpConvMerge (crop + concat) stages, 2-class head."""
    from . import neuromancer as nm
    if name is not None:
        nm.model_manager.newmodel(name)
    inp = nm.Input(in_sh, 'b,f,z,x,y', name='raw')
    skips = []
    out = inp
    for n_f in (32, 64, 128):                      # contracting path
        out = nm.Conv(out, n_f, (1, 3, 3))
        out = nm.Conv(out, n_f, (1, 3, 3))
        skips.append(out)
        out = nm.Pool(out, (1, 2, 2), mode='max')
    out = nm.Conv(out, 256, (3, 3, 3))
    out = nm.Conv(out, 256, (3, 3, 3))
    for skip, up_f, n_f, k in ((skips[2], 512, 256, (1, 3, 3)),   # expanding path
                               (skips[1], 256, 128, (3, 3, 3)),
                               (skips[0], 128, 64, (3, 3, 3))):
        out = nm.UpConvMerge(skip, out, up_f)
        out = nm.Conv(out, n_f, k)
        out = nm.Conv(out, n_f, k)
    feat = out
    barr = nm.Conv(feat, 2, (1, 1, 1), activation_func='lin', name='barr')
    probs = nm.Softmax(barr)
    target = nm.Input_like(feat, override_f=1, name='target')
    loss_pix = nm.MultinoulliNLL(probs, target, target_is_sparse=True, name='nll_barr')
    loss = nm.AggregateLoss(loss_pix, name='loss')
    errors = nm.Errors(probs, target, target_is_sparse=True)
    model = nm.model_manager.current if name is not None else nm.model_manager.getmodel()
    model.designate_nodes(input_node=inp, target_node=target, loss_node=loss,
                          prediction_node=probs, prediction_ext=[loss, errors, probs])
    return model


def unet3d(in_sh=(None, 1, 116, 132, 132), name=None):
    """examples/unet3d.py:61-100 (BASELINE configs[4]): three (2,2,2) poolings, two
    (3,3,3) convs per level, three UpConvMerge stages with UpConv p=(2,2,2), 2-class head;
    (1,1,116,132,132) -> (1,2,28,44,44)."""
    from . import neuromancer as nm
    if name is not None:
        nm.model_manager.newmodel(name)
    inp = nm.Input(in_sh, 'b,f,z,x,y', name='raw')
    conv0 = nm.Conv(inp, 32, (3, 3, 3))
    conv1 = nm.Conv(conv0, 64, (3, 3, 3))
    down0 = nm.Pool(conv1, (2, 2, 2), mode='max')
    conv2 = nm.Conv(down0, 64, (3, 3, 3))
    conv3 = nm.Conv(conv2, 128, (3, 3, 3))
    down1 = nm.Pool(conv3, (2, 2, 2), mode='max')
    conv4 = nm.Conv(down1, 128, (3, 3, 3))
    conv5 = nm.Conv(conv4, 256, (3, 3, 3))
    down2 = nm.Pool(conv5, (2, 2, 2), mode='max')
    conv6 = nm.Conv(down2, 256, (3, 3, 3))
    conv7 = nm.Conv(conv6, 512, (3, 3, 3))
    mrg0 = nm.UpConvMerge(conv5, conv7, 512)
    mconv0 = nm.Conv(mrg0, 256, (3, 3, 3))
    mconv1 = nm.Conv(mconv0, 256, (3, 3, 3))
    mrg1 = nm.UpConvMerge(conv3, mconv1, 256)
    mconv2 = nm.Conv(mrg1, 128, (3, 3, 3))
    mconv3 = nm.Conv(mconv2, 128, (3, 3, 3))
    mrg2 = nm.UpConvMerge(conv1, mconv3, 128)
    mconv4 = nm.Conv(mrg2, 64, (3, 3, 3))
    mconv5 = nm.Conv(mconv4, 64, (3, 3, 3))
    barr = nm.Conv(mconv5, 2, (1, 1, 1), activation_func='lin', name='barr')
    probs = nm.Softmax(barr)
    target = nm.Input_like(mconv5, override_f=1, name='target')
    loss_pix = nm.MultinoulliNLL(probs, target, target_is_sparse=True, name='nll_barr')
    loss = nm.AggregateLoss(loss_pix, name='loss')
    errors = nm.Errors(probs, target, target_is_sparse=True)
    model = nm.model_manager.current if name is not None else nm.model_manager.getmodel()
    model.designate_nodes(input_node=inp, target_node=target, loss_node=loss,
                          prediction_node=probs, prediction_ext=[loss, errors, probs])
    return model


def mnist(in_sh=(None, 1, 26, 26), name=None):
    """examples/mnist.py:29-56 (BASELINE config 1, the reference's CPU-runnable case): three
    2-D convs with train-mode batch normalisation, two Perceptrons, 10-class softmax."""
    from . import neuromancer as nm
    if name is not None:
        nm.model_manager.newmodel(name)
    inp = nm.Input(in_sh, 'b,f,y,x', name='raw')
    out = nm.Conv(inp, 12, (3, 3), (2, 2), batch_normalisation='train')
    out = nm.Conv(out, 36, (3, 3), (2, 2), batch_normalisation='train')
    out = nm.Conv(out, 64, (3, 3), (1, 1), batch_normalisation='train')
    out = nm.Perceptron(out, 200, flatten=True)
    out = nm.Perceptron(out, 10, activation_func='lin')
    out = nm.Softmax(out)
    target = nm.Input_like(out, override_f=1, name='target')
    loss = nm.MultinoulliNLL(out, target, name='nll_', target_is_sparse=True)
    loss = nm.AggregateLoss(loss)
    errors = nm.Errors(out, target, target_is_sparse=True)
    model = nm.model_manager.current if name is not None else nm.model_manager.getmodel()
    model.designate_nodes(input_node=inp, target_node=target, loss_node=loss,
                          prediction_node=out, prediction_ext=[loss, errors, out])
    return model


def _pget(params):
    def P(i):
        if params is None:
            return {}
        w, b = params[i]
        return dict(w=np.asarray(w, np.float32), b=np.asarray(b, np.float32))
    return P


def _finish(nm, inp, out, name, mfp=False):
    probs = nm.Softmax(out)
    if mfp:
        dense = nm.FragmentsToDense(probs)
        model = nm.model_manager.current if name is not None else nm.model_manager.getmodel()
        model.designate_nodes(input_node=inp, prediction_node=dense)
        return model
    target = nm.Input_like(probs, override_f=1, name='target')
    loss_pix = nm.MultinoulliNLL(probs, target, target_is_sparse=True)
    loss = nm.AggregateLoss(loss_pix, name='loss')
    errors = nm.Errors(probs, target, target_is_sparse=True)
    model = nm.model_manager.current if name is not None else nm.model_manager.getmodel()
    model.designate_nodes(input_node=inp, target_node=target, loss_node=loss,
                          prediction_node=probs, prediction_ext=[loss, errors, probs])
    return model
