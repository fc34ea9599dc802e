"""Where does a gradient tensor of a deep net start to deviate from float64?
Per node: forward output and dL/d(output) of the HIP plan against the float64 CPU
evaluation of the same graph that takes the SAME relu decisions (tests/
test_native_size_gpu.py: mirror / hip_relu_decisions).  Diagnostic only.
usage: python tools/unet_diag.py [unet3d|unet3d_lite|neuro3d|neuro3d_lite]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import torch_step as TS
import test_native_size_gpu as T


def main():
    from elektronn2_amd import nets, neuromancer as nm
    which = sys.argv[1] if len(sys.argv) > 1 else "unet3d"
    nm.model_manager.reset()
    np.random.seed(7)
    model = getattr(nets, which)()
    sp = tuple(model.input_node.shape.spatial_shape)
    osp = tuple(model.prediction_node.shape.spatial_shape)
    rng = np.random.RandomState(8)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + osp).astype(np.float32)
    torch.set_num_threads(16)
    if os.environ.get('DIAG_LOSS_FIRST'):
        model.loss(x, t)
    g = model.gradients(x, t)
    plan = model._grad_func.func
    masks = T.hip_relu_decisions(model)
    # float64 evaluation with the same decisions, keeping every node's value and gradient
    dtype = torch.float64
    P = {k: torch.tensor(p.get_value(), dtype=dtype, requires_grad=True)
         for k, p in model.loss_node.all_trainable_params.items()}
    val, logits = {}, None
    for node in model.loss_node.all_parents.values():
        cls = type(node).__name__
        if node is model.input_node:
            val[node] = torch.tensor(x, dtype=dtype)
            continue
        if cls == 'UpConv':
            y = F.conv_transpose3d(val[node.parent], P[node.name + '_w'].permute(1, 0, 2, 3, 4),
                                   stride=tuple(node.pool_shape)) + P[node.name + '_b'].view(1, -1, 1, 1, 1)
            v = y * torch.as_tensor(masks[node.name]).to(dtype) if node.activation_func == 'relu' else y
        elif cls == 'Conv':
            y = F.conv3d(val[node.parent], P[node.name + '_w'].flip(2, 3, 4))
            if tuple(node.pool_shape) != (1, 1, 1):
                y = F.max_pool3d(y, tuple(node.pool_shape))
            y = y + P[node.name + '_b'].view(1, -1, 1, 1, 1)
            v = y * torch.as_tensor(masks[node.name]).to(dtype) if node.activation_func == 'relu' else y
        elif cls == 'Pool':
            v = F.max_pool3d(val[node.parent], node.pool_shape)
        elif cls == 'Crop':
            u, c = val[node.parent], node.crop
            v = u[:, :, c[0]:u.shape[2] - c[0], c[1]:u.shape[3] - c[1], c[2]:u.shape[4] - c[2]]
        elif cls == 'Concat':
            v = torch.cat([val[q] for q in node.parent], dim=1)
        elif cls == 'Softmax':
            logits = val[node.parent]
            continue
        else:
            continue
        v.retain_grad()
        val[node] = v
    L, _ = TS.nll_loss(logits, torch.tensor(t, dtype=dtype))
    L.backward()
    print("%-10s %-8s %12s %12s" % ("node", "class", "fwd err", "dL/dout err"))
    for node, v in val.items():
        if node is model.input_node:
            continue
        ho, hg = plan.out.get(node), plan.grad.get(node)
        ef = T.relmax(ho.cpu().numpy(), v.detach().numpy()) if ho is not None else float('nan')
        eg = (T.relmax(hg.cpu().numpy(), v.grad.numpy())
              if (hg is not None and v.grad is not None) else float('nan'))
        print("%-10s %-8s %12.2e %12.2e" % (node.name, type(node).__name__, ef, eg))
    names = list(model.loss_node.all_trainable_params.keys())
    for i, nme in enumerate(names):
        print("%-10s param grad err %.2e" % (nme, T.relmax(g[i], P[nme].grad.numpy())))


if __name__ == "__main__":
    main()
