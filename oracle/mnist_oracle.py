"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone; never by the product).

CPU restatement (torch, float64, autograd) of BASELINE config 1, the reference's own
CPU-runnable case: examples/mnist.py:29-56 --

    Input (b,1,26,26) 'b,f,y,x'
    Conv 12 (3,3) pool (2,2) batch_normalisation='train'      neural.py:640-712
    Conv 36 (3,3) pool (2,2) batch_normalisation='train'
    Conv 64 (3,3) pool (1,1) batch_normalisation='train'
    Perceptron 200, flatten=True                                neural.py:258-378
    Perceptron 10, 'lin'
    Softmax -> MultinoulliNLL(target_is_sparse) -> AggregateLoss   loss.py:261-347,1357-1363

Arithmetic it follows:
  * 2-D conv = true convolution (kernel flipped in both axes), 'valid'
    (computations.py:350-362; the same flip rule the 3-D path is pinned on by
    tests/test_conv.py:89-104's np.convolve relation);
  * max-pool, non-overlapping (computations.py:633-639);
  * batch norm in 'train' mode, applied to the POOLED conv output / the dot product:
    mean and population std over every axis but 'f', std + 1e-6,
    out = act((gamma / std) * lin + b - gamma * mean / std)   (neural.py:681-711, 352-378);
    the gradient flows through the batch statistics (T.grad of the same expression);
    running statistics m <- 0.9995 m + 0.0005 mean, s <- 0.9995 s + 0.0005 std (extra updates);
  * gamma carries apply_reg = 3.0: three times the weight decay (neural.py:213-214,
    optimiser.py:311-315); mean / std are not trained;
  * relu(x) = 0.5 (x + |x|)  (T.nnet.relu; slope 0.5 at exactly 0).

parity unpinned: Theano is not installable here and the reference's tests hold no numbers
for this path; what this file pins the HIP kernels against is autograd over
the restated forward pass, which is independent of the hand-derived backward formulas in
csrc/dense_bn.hip.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

EPS_NLL = 1e-5

CONVS = [(12, (3, 3), (2, 2)), (36, (3, 3), (2, 2)), (64, (3, 3), (1, 1))]
DENSE = [(200, 'relu'), (10, 'lin')]


def relu(x):
    return 0.5 * (x + torch.abs(x))


def batchnorm(lin, gamma, b, axis_f=1):
    """train-mode statistics; returns (normalised + shifted, mean, std)"""
    red = [i for i in range(lin.dim()) if i != axis_f]
    shape = [1] * lin.dim()
    shape[axis_f] = -1
    mean = lin.mean(dim=red)
    std = torch.sqrt(((lin - mean.view(shape)) ** 2).mean(dim=red)) + 1e-6
    out = (gamma / std).view(shape) * lin + (b - gamma * mean / std).view(shape)
    return out, mean, std


def forward(P, x, train=True):
    """P: name -> float64 tensor (conv{i}_{w,b,gamma,mean,std}, dot{,1}_{w,b}).
    Returns (logits, [(mean, std) of every batch-norm layer])."""
    h = x
    stats = []
    for i, (n_f, k, pool) in enumerate(CONVS):
        nm = 'conv%s' % (i or '')
        lin = F.conv2d(h, P[nm + '_w'].flip(2, 3))
        if pool != (1, 1):
            lin = F.max_pool2d(lin, pool)
        if train:
            y, mean, std = batchnorm(lin, P[nm + '_gamma'], P[nm + '_b'])
            stats.append((mean.detach(), std.detach()))
        else:
            g, m, s = P[nm + '_gamma'], P[nm + '_mean'], P[nm + '_std']
            y = (g / s).view(1, -1, 1, 1) * lin + (P[nm + '_b'] - g * m / s).view(1, -1, 1, 1)
        h = relu(y)
    h = h.flatten(1)
    for i, (n_f, act) in enumerate(DENSE):
        nm = 'dot%s' % (i or '')
        h = h @ P[nm + '_w'] + P[nm + '_b']
        if act == 'relu':
            h = relu(h)
    return h, stats


def nll(logits, target):
    """softmax -> MultinoulliNLL (sparse target (b,1)) -> AggregateLoss; loss.py:261-347"""
    p = torch.softmax(logits, dim=1)
    C = p.shape[1]
    onehot = (target.view(-1, 1) == torch.arange(C, dtype=target.dtype).view(1, C)).to(p.dtype)
    n_tot = onehot.sum()
    nll_el = -(onehot * torch.log(p + EPS_NLL)) * p.numel() / (n_tot + EPS_NLL) / C
    return nll_el.sum(dim=1, keepdim=True).mean(), p


def loss_and_grads(params, x, t, train=True):
    """params: name -> numpy array.  Returns (loss, {name: grad} for the trainable ones,
    probabilities, batch statistics)."""
    P = OrderedDict((k, torch.tensor(np.asarray(v, np.float64),
                                     requires_grad=not (k.endswith('_mean') or k.endswith('_std'))))
                    for k, v in params.items())
    logits, stats = forward(P, torch.tensor(np.asarray(x, np.float64)), train)
    loss, probs = nll(logits, torch.tensor(np.asarray(t, np.float64)))
    loss.backward()
    grads = {k: p.grad.numpy() for k, p in P.items() if p.requires_grad and p.grad is not None}
    return float(loss.detach()), grads, probs.detach().numpy(), [(m.numpy(), s.numpy()) for m, s in stats]


def classification_errors(probs, t):
    return float(np.mean(np.argmax(probs, axis=1) != np.asarray(t).reshape(-1)))
