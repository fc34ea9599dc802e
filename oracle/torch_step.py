"""CPU ORACLE, second implementation (test infrastructure, NOT product code).

torch-CPU fp32 (oneDNN) restatement of one ELEKTRONN2 training step:
flipped conv3d -> max_pool3d -> +bias -> relu per Conv node
(neuromancer/neural.py:662-712), channel softmax (computations.py:175-176),
MultinoulliNLL + AggregateLoss (loss.py:261-347, 1357-1363), autograd backward
(stands in for T.grad, model.py:182), the reference's Adam
(optimiser.py:273-334).

Two uses:
  * independent second opinion on ``e2_oracle`` (different code, fp32);
  * ``bench.py``'s ``cpu_baseline`` leg, kind "port": the literal Theano-CPU
    path cannot be timed (Theano absent, no network) -- see BASELINE.md §3.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
PARITY STATUS: parity unpinned (see e2_oracle.py header).
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn.functional as F

EPS_NLL = 1e-5
EPS_ADAM = 1e-5


def conv_node(x, w, b, pool, act):
    y = F.conv3d(x, w.flip(2, 3, 4))                 # F1: true convolution
    if tuple(pool) != (1, 1, 1):
        y = F.max_pool3d(y, tuple(pool))
    y = y + b.view(1, -1, 1, 1, 1)
    if act == 'relu':
        y = torch.relu(y)
    elif act != 'lin':
        raise NotImplementedError(act)
    return y


def upconv_node(x, w, b, pool, act):
    """F2: y[n,co,p*i+r] = sum_ci w[co,ci,r] x[n,ci,i]."""
    y = F.conv_transpose3d(x, w.permute(1, 0, 2, 3, 4), stride=tuple(pool))
    y = y + b.view(1, -1, 1, 1, 1)
    if act == 'relu':
        y = torch.relu(y)
    return y


def nll_loss(logits, target):
    p = torch.softmax(logits, dim=1)
    C = p.shape[1]
    classes = torch.arange(C, dtype=target.dtype).view(1, C, 1, 1, 1)
    onehot = (target == classes).to(p.dtype)
    n_tot = onehot.sum()
    nll = -(onehot * torch.log(p + EPS_NLL)) * p.numel() / (n_tot + EPS_NLL) / C
    nll = nll.sum(dim=1, keepdim=True)
    return nll.mean(), p


class TorchNet:
    """Sequential Conv-node net (neuro3d / neuro3d_lite) with reference Adam."""

    def __init__(self, spec, params, dtype=torch.float32):
        self.spec = spec
        self.w = [torch.tensor(np.asarray(w), dtype=dtype, requires_grad=True)
                  for w, _ in params]
        self.b = [torch.tensor(np.asarray(b), dtype=dtype, requires_grad=True)
                  for _, b in params]
        self.m = [torch.zeros_like(p) for p in self.w + self.b]
        self.s = [torch.zeros_like(p) for p in self.w + self.b]
        self.t = 0

    def forward(self, x):
        h = x
        for (n_f, k, p, act), w, b in zip(self.spec, self.w, self.b):
            h = conv_node(h, w, b, p, act)
        return h

    def loss_and_grads(self, x, target):
        for p in self.w + self.b:
            p.grad = None
        loss, probs = nll_loss(self.forward(x), target)
        loss.backward()
        return loss.detach(), probs.detach()

    @torch.no_grad()
    def adam(self, lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4):
        self.t += 1
        t = self.t
        factor = np.sqrt(1 - beta2 ** t) / (1 - mom ** t)
        params = self.w + self.b
        for i, p in enumerate(params):
            g = p.grad
            self.m[i].mul_(mom).add_(g, alpha=1 - mom)
            self.s[i].mul_(beta2).addcmul_(g, g, value=1 - beta2)
            direction = factor * self.m[i] / torch.sqrt(self.s[i] + EPS_ADAM)
            if i < len(self.w):                      # apply_reg only on w
                direction = direction + wd * p
            p.sub_(lr * direction)

    def trainingstep(self, x, target, **kw):
        loss, _ = self.loss_and_grads(x, target)
        self.adam(**kw)
        return float(loss)


def time_cpu_step(spec, params, x, target, n_threads, warmup=1, steps=3):
    """Median wall seconds of one full training step on ``n_threads`` host
    threads.  Returns (median_s, all_times)."""
    torch.set_num_threads(n_threads)
    net = TorchNet(spec, params)
    xt = torch.tensor(np.asarray(x), dtype=torch.float32)
    tt = torch.tensor(np.asarray(target), dtype=torch.float32)
    for _ in range(warmup):
        net.trainingstep(xt, tt)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        net.trainingstep(xt, tt)
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts
