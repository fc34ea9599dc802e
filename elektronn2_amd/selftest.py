"""smoke(): one tiny training step of neuro3d_lite on cuda:0 through the HIP
path, checked against the CPU oracle (the oracle is the CHECKER here, never the
thing being run)."""
from __future__ import annotations

import numpy as np


def smoke(verbose=True):
    import torch
    from . import backend, nets          # raises if libe2hip.so is missing
    from . import neuromancer as nm
    from oracle import e2_oracle as O
    assert torch.cuda.is_available(), "smoke() needs the GPU"
    sp = (7, 47, 47)
    spec = O.NEURO3D_LITE
    params = O.init_net(spec, 1, seed=1)
    rng = np.random.RandomState(0)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    osp = O.net_out_shape(spec, sp)
    t = rng.randint(0, 2, (1, 1) + osp).astype(np.float32)
    nm.model_manager.reset()
    model = nets.neuro3d_lite((None, 1) + sp, params=params)
    model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    loss, tsec, _ = model.trainingstep(x, t, optimiser='Adam')
    ref_losses, ref_P = O.net_train_steps(spec, params, x, t, 1)
    rel = abs(float(loss) - ref_losses[0]) / abs(ref_losses[0])
    assert rel < 1e-4, "loss %r vs oracle %r" % (loss, ref_losses[0])
    w_last = model.nodes['conv6'].w.get_value()
    err = np.abs(w_last - ref_P[6][0]).max() / np.abs(ref_P[6][0]).max()
    assert err < 5e-4, "updated weights differ from the oracle: %g" % err
    w0 = model.nodes['conv'].w.get_value()
    err0 = np.abs(w0 - ref_P[0][0]).max() / np.abs(ref_P[0][0]).max()
    assert err0 < 5e-4, "updated first-layer weights differ from the oracle: %g" % err0
    if verbose:
        print("smoke ok: loss %.6f (oracle %.6f), rel %.2e, dW rel %.2e / %.2e, step %.3f ms"
              % (loss, ref_losses[0], rel, err, err0, tsec * 1e3))
