"""Softmax / MultinoulliNLL / AggregateLoss / Errors with the reference's
constructor signatures (elektronn2/neuromancer/loss.py:33-93, 141-351,
693-826, 1279-1370).

HIP execution covers the pattern every BASELINE config uses:
``AggregateLoss(MultinoulliNLL(Softmax(lin-Conv), target, target_is_sparse=True))``
which reduces to  loss = sum_labelled -log(p_target + 1e-5) / (n_labelled + 1e-5)
(the pred.size / n_class / mean factors of loss.py:342-346,1357-1363 cancel).
Class / example weights, masks, weakness and dense targets are outside the hot path
and raise NotImplementedError.  ``Softmax(n_indep > 1)`` (independent softmaxes over
consecutive feature groups, loss.py:82-92) exists for ``MalisNLL`` (loss.py:560-690,
SURVEY.md 8f-4): forward and gradient on the device, the MALIS counts by the host C++
of csrc/malis.cpp between the forward and the backward segment of the step.
"""
from __future__ import annotations

import numpy as np

from .graphutils import TaggedShape, floatX
from .node_basic import Node, Sym
from .variables import VariableParam

__all__ = ['Softmax', 'MultinoulliNLL', 'MalisNLL', 'AggregateLoss', 'Classification',
           'Errors']

EPS = 1e-5     # loss.py:30


class Softmax(Node):
    def __init__(self, parent, n_class='auto', n_indep=1, name="softmax", print_repr=True):
        super(Softmax, self).__init__(parent, name, print_repr)
        n_f = parent.shape['f']
        if hasattr(parent, 'activation_func'):
            if parent.activation_func != 'lin':
                raise ValueError("The parent of a Softmax-node must have a "
                                 "linear activation function.")
        if n_class == 'auto':
            if n_f % n_indep == 0:
                n_class = n_f // n_indep
            else:
                raise ValueError("Cannot create %i-fold %i-class softmax from %i features."
                                 % (n_indep, n_f // n_indep, n_f))
        elif n_class * n_indep != n_f:
            raise ValueError("Cannot create %i-fold %i-class softmax " % (n_indep, n_class))
        self.n_class = n_class
        self.n_indep = n_indep

    def _calc_comp_cost(self):
        self.computational_cost = self.parent.shape.stripnone_prod

    def _plan_alloc(self, plan):
        plan.alloc_out(self)
        plan.scratch[self, 'stats'] = plan.zeros_flat(2)
        plan.scratch[self, 'dummy_t'] = None

    def _plan_fwd(self, plan):
        # probs only; when an NLL node hangs on this softmax it re-runs the same
        # kernel with the target and also fills the loss statistics.
        if plan.scratch.get((self, 'fused_nll')):
            return
        head = self._head(plan)
        if head is not None:                 # conv + softmax in one kernel
            plan.ctx.head_fwd(plan.out[head.parent], plan.param(head.w), plan.param(head.b),
                              None, plan.out[self], None)
            return
        t = plan.scratch.get((self, 'dummy_t'))
        if t is None:
            sh = list(plan.out_shape(self))
            sh[1] = 1
            t = plan.full(tuple(sh), -1.0)
            plan.scratch[self, 'dummy_t'] = t
        lg, pr, k = plan.out[self.parent], plan.out[self], self.n_class
        for i in range(self.n_indep):            # loss.py:82-92: one softmax per group
            sl = slice(i * k, (i + 1) * k)
            plan.ctx.softmax_nll_fwd(lg[:, sl], t, pr[:, sl], plan.scratch[self, 'stats'])

    def _head(self, plan):
        """the parent Conv when it runs as a fused classifier head (csrc/head.hip)"""
        p = self.parent
        f = getattr(p, '_fused_head', None)
        return p if (f is not None and f(plan) is self) else None

    def _plan_bwd(self, plan):
        if plan.scratch.get((self, 'fused_nll')) or plan.scratch.get((self, 'loss_writes_dlogits')):
            return            # the NLL node wrote d(loss)/d(logits) directly
        raise NotImplementedError("gradient through a bare Softmax node")


class MultinoulliNLL(Node):
    def __init__(self, pred, target, target_is_sparse=False, class_weights=None,
                 example_weights=None, weakness=0, mask_class_labeled=None,
                 mask_class_not_present=None, name="nll", print_repr=True):
        super(MultinoulliNLL, self).__init__([pred, target], name, print_repr)
        if not isinstance(pred, Softmax):
            raise ValueError("The prob input to a MultinoulliNLL-node must be "
                             "a Softmax-Node.")
        if pred.n_indep != 1:
            raise NotImplementedError("MultinoulliNLL over n_indep > 1 is outside the "
                                      "HIP hot path")
        if (class_weights is not None or example_weights is not None or weakness or
                mask_class_labeled is not None or mask_class_not_present is not None):
            raise NotImplementedError("class/example weights, masks and weak training "
                                      "are outside the HIP hot path")
        if not target_is_sparse:
            raise NotImplementedError("dense (one-hot) targets are outside the HIP hot path")
        self.target = target
        self.pred = pred
        self.axis = pred.shape.tag2index('f')
        self.n_class = pred.n_class
        self.n_indep = pred.n_indep
        self.target_is_sparse = target_is_sparse
        self.class_weights = None
        self.example_weights = None
        self.weakness = 0

    def _calc_shape(self):
        self.shape = self.parent[0].shape.updateshape(self.axis, 1)

    def _calc_comp_cost(self):
        self.computational_cost = self.parent[0].shape.stripnone_prod

    def _plan_alloc(self, plan):
        plan.scratch[self.pred, 'fused_nll'] = True
        plan.scratch[self, 'loss'] = plan.zeros_flat(1)
        head = self.pred._head(plan)
        if head is not None and plan.training and self._tail(plan) is None:
            nb = plan.ctx.head_bwd_ws_bytes(plan.out_shape(head.parent), self.n_class)
            plan.scratch[self, 'head_ws'] = plan.empty_flat(nb // 4 + 16)
        plan.out[self] = None        # the element-wise nll array is never materialised

    def _tail(self, plan):
        """the (1,1,1) relu Conv in front of the fused head when the pair runs as ONE launch,
        forward and backward (csrc/tail.hip, Conv._tail), else None"""
        head = self.pred._head(plan)
        if head is None:
            return None
        c = head.parent
        f = getattr(c, '_tail', None)
        t = f(plan) if f is not None else None
        return c if (t is not None and t[2] is self) else None

    def _plan_fwd(self, plan):
        stats = plan.scratch[self.pred, 'stats']
        tail = self._tail(plan)
        if tail is not None:
            # forward AND backward of [1x1x1 conv + relu] -> head -> loss: probabilities,
            # stats[1] = #labelled, the conv's pre-activation gradient, its parent's output
            # gradient; the partial sums wait in the workspace for the backward half
            head = self.pred._head(plan)
            par = tail.parent
            plan.join_side()
            dx = plan.grad[par] if plan.needs_grad(par) else None
            gm = tail._tail_gm(plan) if dx is not None else None
            kw = {}
            if gm is not None:
                # the parent's activation backward rides along: the launch writes the parent's
                # zero-padded gradient buffer (interior) and its bias gradient's slots
                dx = plan.scratch[par, 'dy']
                kw = dict(gm_mode=gm[0], gm_src=gm[1], gm_bias=gm[2])
            with plan.loss_grad_mode():
                plan.scratch[self, 'tail_slots'] = plan.ctx.tail_fwd_bwd(
                    plan.out[par], plan.scratch[tail, 'wp_f'], plan.scratch.get((tail, 'wp_d')),
                    plan.param(tail.b), plan.param(head.w).reshape(head.n_f, -1),
                    plan.param(head.b), plan.out[self.target], plan.out[self.pred],
                    plan.scratch[tail, 'dy'], dx, stats, plan.scratch[tail, 'tail_ws'], **kw)
            return
        plan.zero_early(stats)
        head = self.pred._head(plan)
        if head is not None:
            plan.ctx.head_fwd(plan.out[head.parent], plan.param(head.w), plan.param(head.b),
                              plan.out[self.target], plan.out[self.pred], stats)
            return
        plan.ctx.softmax_nll_fwd(plan.out[self.pred.parent], plan.out[self.target],
                                 plan.out[self.pred], stats)

    def _plan_bwd(self, plan):
        with plan.loss_grad_mode():
            self._plan_bwd_launches(plan)

    def _plan_bwd_launches(self, plan):
        head = self.pred._head(plan)
        tail = self._tail(plan)
        if tail is not None:
            # (behind the zero fill of the gradient arena: the slots are ADDED into it)
            par = tail.parent
            gm = tail._tail_gm(plan) if plan.needs_grad(par) else None
            plan.ctx.tail_reduce(plan.scratch[tail, 'tail_ws'], plan.scratch[self, 'tail_slots'],
                                 tail.n_f, head.n_f, plan.pgrad(head.w).reshape(head.n_f, -1),
                                 plan.pgrad(head.b), plan.pgrad(tail.b),
                                 plan.scratch[self.pred, 'stats'], plan.scratch[self, 'loss'],
                                 db_parent=plan.pgrad(par.b) if gm is not None else None)
            return
        if head is not None:
            dst, first = (plan.grad_slot(head.parent) if plan.needs_grad(head.parent)
                          else (None, True))
            plan.ctx.head_bwd(plan.out[head.parent], plan.param(head.w), plan.out[self.pred],
                              plan.out[self.target], plan.scratch[self.pred, 'stats'], dst,
                              not first, plan.pgrad(head.w), plan.pgrad(head.b),
                              plan.scratch[self, 'loss'], ws=plan.scratch[self, 'head_ws'])
            return
        logits = self.pred.parent
        dst, first = plan.grad_slot(logits)
        if not first:
            raise NotImplementedError("logits consumed by several nodes")
        plan.ctx.softmax_nll_bwd(plan.out[self.pred], plan.out[self.target],
                                 plan.scratch[self.pred, 'stats'], dst,
                                 plan.scratch[self, 'loss'])

    def loss_value(self, plan):
        """device scalar: loss_sum / (n_labelled + EPS)."""
        s = plan.scratch[self.pred, 'stats']
        return s[0] / (s[1] + EPS)


class MalisNLL(Node):
    """loss.py:560-690.  ``pred``: Softmax with ``n_indep`` = #edges and 2 classes
    (feature 2e = "disconnected", 2e+1 = affinity of edge e), ``aff_gt`` (1, E, z, x, y)
    and ``seg_gt`` (1, 1, z, x, y) Input nodes, ``nhood`` (E, 3).

    total loss (after AggregateLoss's mean; the nll.size factors cancel, loss.py:664-670)
        = -sum(pos * log(p_aff + EPS) + neg * log(p_dis + EPS)) / (n_pos + n_neg + EPS)
    with the MALIS counts as constants of the gradient (malisop.py:114-120).  The counts
    come from the host (csrc/malis.cpp, Kruskal is sequential): the step is cut after
    the forward pass, the affinities go to the host, the counts come back, the loss
    kernel and the backward pass follow as a second captured segment.  After each step
    ``rand_index``, ``false_splits``, ``false_merges``, ``pos_count``, ``neg_count``
    hold the values of the reference's inspection outputs (loss.py:672-682)."""

    def __init__(self, pred, aff_gt, seg_gt, nhood, unrestrict_neg=True, class_weights=None,
                 example_weights=None, name="nll", print_repr=True):
        super(MalisNLL, self).__init__([pred, aff_gt, seg_gt], name, print_repr)
        if not isinstance(pred, Softmax):
            raise ValueError("The prob input to a MultinoulliNLL-node must be "
                             "a Softmax-Node.")
        if pred.shape['b'] != 1:
            raise NotImplementedError("Malis can only be used with batch size 1.")
        if class_weights is not None or example_weights is not None:
            raise NotImplementedError("class / example weights are outside the HIP hot path")
        if pred.n_class != 2:
            raise NotImplementedError("MalisNLL needs 2-class softmaxes (one per edge)")
        self.aff_gt = aff_gt
        self.seg_gt = seg_gt
        self.pred = pred
        self.nhood = np.asarray(nhood, dtype=np.int32)
        if self.nhood.shape != (pred.n_indep, 3):
            raise ValueError("nhood must be (%i, 3) for this prediction" % pred.n_indep)
        self.unrestrict_neg = unrestrict_neg
        self.axis = pred.shape.tag2index('f')
        self.n_class = pred.n_class
        self.n_indep = pred.n_indep
        self.class_weights = None
        self.example_weights = None
        self.rand_index = self.false_splits = self.false_merges = None
        self.pos_count = self.neg_count = None

    def _calc_shape(self):
        self.shape = self.parent[0].shape.updateshape(self.axis, 1)

    def _calc_comp_cost(self):
        self.computational_cost = self.parent[0].shape.stripnone_prod

    def _plan_alloc(self, plan):
        plan.scratch[self.pred, 'loss_writes_dlogits'] = True
        sh = plan.out_shape(self.pred)
        n = self.n_indep * int(np.prod(sh[2:]))
        plan.scratch[self, 'pos'] = plan.zeros_flat(n)
        plan.scratch[self, 'neg'] = plan.zeros_flat(n)
        plan.scratch[self, 'norm'] = plan.zeros_flat(4)
        plan.scratch[self, 'loss'] = plan.zeros_flat(1)
        plan.out[self] = None

    def _plan_fwd(self, plan):
        pass                   # everything happens after the host step

    def _plan_host(self, plan):
        """affinities -> host, MALIS counts (two Kruskal passes) -> device"""
        import torch
        from .. import malis
        plan.stream.synchronize()
        probs = plan.out[self.pred]
        aff = probs[0, 1::2].cpu().numpy()
        aff_gt = plan.out[self.aff_gt][0].cpu().numpy().astype(np.int16)
        seg_gt = plan.out[self.seg_gt][0, 0].cpu().numpy().astype(np.int32)
        pos, neg = malis.malis_weights(aff, aff_gt, seg_gt, self.nhood, self.unrestrict_neg)
        n_pos, n_neg = float(pos.sum(dtype=np.float64)), float(neg.sum(dtype=np.float64))
        n_tot = n_pos + n_neg
        norm = np.array([1.0 / (n_tot + EPS), n_tot, n_pos, n_neg], np.float32)
        plan.scratch[self, 'pos'].copy_(torch.from_numpy(pos.astype(np.float32).ravel()))
        plan.scratch[self, 'neg'].copy_(torch.from_numpy(neg.astype(np.float32).ravel()))
        plan.scratch[self, 'norm'].copy_(torch.from_numpy(norm))
        self.pos_count, self.neg_count = pos, neg
        self.false_splits = int(pos[aff < 0.5].sum(dtype=np.uint64))
        self.false_merges = int(neg[aff > 0.5].sum(dtype=np.uint64))
        self.rand_index = np.float32((self.false_splits + self.false_merges) / (n_tot + EPS))

    def _launch(self, plan, dlogits):
        plan.ctx.fill(plan.scratch[self, 'loss'], 0.0)
        plan.ctx.malis_nll(plan.out[self.pred], plan.scratch[self, 'pos'],
                           plan.scratch[self, 'neg'], plan.scratch[self, 'norm'], dlogits,
                           plan.scratch[self, 'loss'])

    def _plan_fwd_post(self, plan):
        if not plan.training:
            self._launch(plan, None)

    def _plan_bwd(self, plan):
        logits = self.pred.parent
        dst, first = plan.grad_slot(logits)
        if not first:
            raise NotImplementedError("logits consumed by several nodes")
        self._launch(plan, dst)          # loss and d(loss)/d(logits) in one launch

    def loss_value(self, plan):
        return plan.scratch[self, 'loss'][0]


class AggregateLoss(Node):
    def __init__(self, parent_nodes, mixing_weights=None, name="total_loss", print_repr=True):
        if not isinstance(parent_nodes, (tuple, list)):
            parent_nodes = [parent_nodes, ]
        super(AggregateLoss, self).__init__(parent_nodes, name, print_repr)
        if mixing_weights is None:
            mixing_weights = np.ones(len(parent_nodes))
        if isinstance(mixing_weights, (tuple, list, np.ndarray)):
            if len(parent_nodes) != len(mixing_weights):
                raise ValueError("Mismatch: len(parent_nodes)=%i, len(weights)=%i"
                                 % (len(parent_nodes), len(mixing_weights)))
            mixing_weights = VariableParam(value=np.array(mixing_weights, dtype=floatX),
                                           name="loss_mixing_weights", dtype=floatX,
                                           apply_train=False)
        else:
            raise ValueError("Unsupported weight format")
        self.params['mixing_weights'] = mixing_weights
        self.mixing_weights = mixing_weights
        if len(parent_nodes) != 1 or not isinstance(parent_nodes[0], (MultinoulliNLL, MalisNLL)):
            raise NotImplementedError("the HIP hot path aggregates exactly one "
                                      "MultinoulliNLL / MalisNLL loss")

    def _calc_shape(self):
        self.shape = TaggedShape([1, ], ['f', ])

    def _calc_comp_cost(self):
        self.computational_cost = np.sum([inp.shape.stripnone_prod for inp in self.parent])

    def _plan_alloc(self, plan):
        plan.out[self] = None

    def _plan_fwd(self, plan):
        pass

    def _plan_bwd(self, plan):
        pass          # d(total)/d(nll) = mixing_weight(=1) / 1 ; folded into the NLL backward

    def host_value(self, plan):
        w = float(self.mixing_weights.get_value()[0])
        return np.float32(float(self.parent[0].loss_value(plan).item()) * w)


class Classification(Node):
    def __init__(self, pred, n_class='auto', n_indep='auto', name="cls", print_repr=True):
        super(Classification, self).__init__(pred, name, print_repr)
        if not isinstance(pred, Softmax):
            raise NotImplementedError("Classification of non-softmax predictions")
        self.n_class = pred.n_class
        self.n_indep = pred.n_indep
        self.sm_input = True
        self.pred = pred

    def _calc_shape(self):
        self.shape = self.parent.shape.updateshape(self.pred.shape.tag2index('f'),
                                                   self.n_indep)

    def _plan_alloc(self, plan):
        plan.out[self] = None

    def _plan_fwd(self, plan):
        pass


class _Errors(Node):
    def __init__(self, cls, target, target_is_sparse=False, name="errors", print_repr=True):
        super(_Errors, self).__init__([cls, target], name, print_repr)
        self.n_class = cls.n_class
        self.n_indep = cls.n_indep
        self.target = target
        self.cls = cls
        self.target_is_sparse = target_is_sparse
        if not target_is_sparse:
            raise NotImplementedError("dense targets are outside the HIP hot path")

    def _calc_shape(self):
        self.shape = TaggedShape([1, ], ['f', ])

    def _plan_alloc(self, plan):
        plan.out[self] = None

    def _plan_fwd(self, plan):
        pass

    def host_value(self, plan):
        """mean(int16(target) != argmax_f(pred))  (loss.py:789-817); evaluated
        with torch ops on the device tensors, outside any captured graph."""
        import torch
        probs = plan.out[self.cls.pred]
        cls = torch.argmax(probs, dim=1, keepdim=True)
        gt = plan.out[self.target].to(torch.int16).to(cls.dtype)
        return np.float32((gt != cls).float().mean().item())


def Errors(pred, target, target_is_sparse=False, n_class='auto', n_indep='auto',
           name="errors", print_repr=True):
    if not isinstance(pred, Classification):
        pred = Classification(pred, n_class=n_class, n_indep=n_indep,
                              name='cls for errors', print_repr=False)
    return _Errors(pred, target, target_is_sparse=target_is_sparse, name=name,
                   print_repr=print_repr)
