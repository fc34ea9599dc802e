"""SGD and Adam with the reference's update rules and globals
(elektronn2/neuromancer/optimiser.py:19-129, 135-165, 273-334).

``lr`` / ``mom`` / ``wd`` are CLASS-level parameters shared by all optimisers
(optimiser.py:19-29); ``beta2`` belongs to Adam.  The update itself is one fused
HIP launch over the model's flat parameter arena (``e2_adam_step`` /
``e2_sgd_step``); the hyper-parameters live in a small device array that is
refreshed from the host values before every step, so learning-rate schedules
work under hipGraph replay.  Adam's step counter ``t`` and bias factor are kept
on the device.  The 3-deep parameter rotation for ``repair()``
(optimiser.py:103-118) is not kept (its only call site is commented out in the
reference, trainer.py:207).  AdaGrad / AdaDelta: not used by any BASELINE config.
"""
from __future__ import annotations

import numpy as np

from . import graphutils
from .variables import VariableParam

__all__ = ['Optimiser', 'SGD', 'Adam']


class Optimiser(object):
    global_lr = VariableParam(value=1, name='lr', dtype=graphutils.floatX)
    global_weight_decay = VariableParam(value=0, name='weight_decay', dtype=graphutils.floatX)
    global_mom = VariableParam(value=0.9, name='mom', dtype=graphutils.floatX)

    @classmethod
    def setlr(cls, val):
        cls.global_lr.set_value(graphutils.as_floatX(val))

    @classmethod
    def setwd(cls, val):
        cls.global_weight_decay.set_value(graphutils.as_floatX(val))

    @classmethod
    def setmom(cls, val):
        cls.global_mom.set_value(graphutils.as_floatX(val))

    step_name = None

    def __init__(self, inputs, loss, params, additional_outputs, model):
        self.meta_params = dict(lr=self.global_lr, mom=self.global_mom,
                                wd=self.global_weight_decay)
        self.input = inputs
        self.output = [loss] + list(additional_outputs or [])
        self.loss = loss
        self.params = params
        self.model = model
        self.last_exec_time = None
        self._hyper = None
        self._hyper_host = None
        self.step = graphutils.make_func(self.input, self.output, name='%s step' % self.step_name,
                                         model=model, step=self.step_name)

    def set_opt_meta_params(self, value_dict):
        for k, v in value_dict.items():
            try:
                self.meta_params[k].set_value(graphutils.as_floatX(v))
            except KeyError:
                raise AttributeError("optimiser has no meta parameter %r" % (k,))

    def _beta2(self):
        return 0.0

    def _sync_hyper(self, plan):
        """refresh {lr, mom, beta2, wd} on the device when the host values changed."""
        import torch
        vals = (float(self.global_lr.get_value()), float(self.global_mom.get_value()),
                float(self._beta2()), float(self.global_weight_decay.get_value()))
        if self._hyper is None:
            self._hyper = torch.zeros(24, dtype=torch.float32, device=plan.ctx.device)   # ([8..23]: arrival counters of e2_adam_pack_step)
            self._apply_pending_hyper()
        if vals != self._hyper_host:
            self._hyper[:4].copy_(torch.tensor(vals, dtype=torch.float32))
            self._hyper_host = vals

    def __call__(self, *args, **kwargs):
        """[data (,labels ...)] -> [loss (, additional outputs ...)].  ``sync=False``: the
        step is only SUBMITTED; the loss returned is the one of the step before it (read
        back without waiting for the new step; the very first call waits for its own)."""
        if self.step.func is None:
            self.step.compile()
        plan = self.step.func
        plan.set_inputs(args)           # builds the plan (and the arena) on first use
        import torch
        with torch.cuda.stream(plan.stream):
            self._ensure_state(plan)
            self._sync_hyper(plan)
        plan.run()
        if not kwargs.get('sync', True) and len(self.output) == 1:
            loss, t = plan.fetch_async()
            if loss is not None:
                self.last_exec_time = t
                self.step.last_exec_time = t
                return [graphutils.as_floatX(loss)]
        ret = list(plan.fetch())
        ret[0] = graphutils.as_floatX(ret[0])
        self.last_exec_time = plan.last_device_time
        self.step.last_exec_time = plan.last_device_time
        return ret

    def steps(self, k, sync=True):
        """``k`` steps with one graph launch (Plan.run_steps) on the batches of the plan's input
        ring -- or k times on the inputs last set; -> (losses of the k steps, device seconds of
        the launch).  ``sync=False``: the launch is only SUBMITTED and the call returns the losses
        and time of the launch BEFORE it (None, None the first time): the host fills the next
        slots of the ring while the device works.  No reference counterpart (training/trainer.py:
        186-194 runs one step per batch and reads its loss back before the next)."""
        plan = self.step.func
        if plan is None or not plan._built:
            raise RuntimeError("steps: run one ordinary step first (it builds the plan)")
        import torch
        # batches written into the ring on the caller's stream: the launch is ordered behind them
        plan.stream.wait_stream(torch.cuda.current_stream(plan.ctx.device))
        with torch.cuda.stream(plan.stream):
            self._ensure_state(plan)
            self._sync_hyper(plan)
        plan.keep_loss_history()
        n0 = plan._n_runs
        plan.run_steps(k)
        one = plan._n_runs - n0 == k
        if sync:
            losses = plan.loss_history(k)               # (waits for the steps)
            t = plan.ctx.elapsed_ms(plan._ev0, plan._ev1) * 1e-3 if one else 0.0   # (the last launch: all k when it was one)
            self.last_exec_time = t
            self.step.last_exec_time = t
            self._steps_async = None
            return losses, t
        # deferred: a snapshot of (history, count, newest loss) goes into pinned memory behind the launch
        a = getattr(self, '_steps_async', None)
        if a is None:
            ns = plan._hist.shape[0]
            a = self._steps_async = dict(n=0, slots=[dict(
                hist=torch.empty(ns, dtype=torch.float32).pin_memory(),
                state=torch.empty(2, dtype=torch.int64).pin_memory(),
                loss=torch.empty(1, dtype=torch.float32).pin_memory(),
                ev=torch.cuda.Event(), k=0, evs=None) for _ in range(2)])
        cur, prev = a['slots'][a['n'] & 1], a['slots'][(a['n'] & 1) ^ 1]
        with torch.cuda.stream(plan.stream):
            cur['hist'].copy_(plan._hist[:, 0], non_blocking=True)
            cur['state'].copy_(plan._step_state, non_blocking=True)
            cur['loss'].copy_(plan._loss_dev().reshape(1), non_blocking=True)
            cur['ev'].record(plan.stream)
        cur['k'], cur['evs'] = k, (plan._ev0, plan._ev1) if one else None
        a['n'] += 1
        if prev['k'] == 0:
            return None, None
        prev['ev'].synchronize()
        t_cnt, ns, kk = int(prev['state'][0]), prev['hist'].numel(), prev['k']
        h = prev['hist'].numpy()
        import numpy as np
        losses = np.array([h[(t_cnt - kk + j) % ns] for j in range(kk - 1)] + [float(prev['loss'][0])], np.float32)
        t = plan.ctx.elapsed_ms(*prev['evs']) * 1e-3 if prev['evs'] else 0.0
        prev['k'] = 0
        self.last_exec_time = t
        self.step.last_exec_time = t
        return losses, t

    # subclasses
    def _ensure_state(self, plan):
        raise NotImplementedError

    def device_update(self, plan):
        raise NotImplementedError

    def _grad_scaling(self, plan):
        """keyword arguments of the update launch: the data-parallel normalisation applied
        on the way in (Plan._dp_scale), the gradient arena cleared on the way out"""
        kw = dict(zero_g=plan._upd_zeroes_g())
        sc = plan._dp_scale()
        if sc is not None:
            if sc[0] == 'sum':
                kw['gdiv'] = self.model.G_spare
            else:
                kw['gmul'] = sc[1]
        return kw

    def state_dict(self):
        return {}

    def load_state_dict(self, d):
        pass

    def _apply_pending_hyper(self):
        pass


class SGD(Optimiser):
    """d' = g + mom*d ; p' = p - lr*(d' + wd*p [*apply_reg])  (optimiser.py:146-160)."""
    step_name = 'SGD'

    def __init__(self, *a):
        self.last_dir = None
        self._pending = None
        super(SGD, self).__init__(*a)

    def _ensure_state(self, plan):
        if self.last_dir is None:
            self.last_dir = plan.zeros_flat(max(self.model.n_train, 4))
        if self._pending is not None:
            self._apply(self._pending)
            self._pending = None

    def device_update(self, plan):
        m = self.model
        plan.ctx.sgd_step(m.P[:m.n_train] if m.n_train < m.P.numel() else m.P, m.G,
                          self.last_dir, m.seg_off, m.seg_reg, self._hyper,
                          **self._grad_scaling(plan))

    def clear_last_dir(self):
        if self.last_dir is not None:
            self.last_dir.zero_()

    def state_dict(self):
        if self.last_dir is None:
            # a state loaded into a model that has not stepped yet is still parked in
            # _pending: hand it on, so load -> save -> load does not drop it
            if self._pending is not None and 'last_dir' in self._pending:
                return {'last_dir': np.array(self._pending['last_dir'], dtype=np.float32)}
            return {}
        return {'last_dir': self.last_dir.cpu().numpy()}

    def _apply(self, d):
        import torch
        if 'last_dir' in d:
            v = np.ascontiguousarray(d['last_dir'], dtype=np.float32)
            if v.size != self.last_dir.numel():
                raise ValueError("SGD state of %i elements does not fit this model (%i)"
                                 % (v.size, self.last_dir.numel()))
            self.last_dir.copy_(torch.from_numpy(v))

    def load_state_dict(self, d):
        """the buffers are created lazily by the first step: a state loaded into a fresh
        model is kept and applied when they exist (``create_model(); modelload(f, model)``)"""
        if self.last_dir is None:
            self._pending = dict(d)
        else:
            self._apply(d)


class Adam(Optimiser):
    """optimiser.py:273-334: eps = 1e-5 INSIDE the sqrt, bias factor
    sqrt(1-beta2^t)/(1-mom^t), L2 term outside the adaptive scaling."""
    step_name = 'Adam'

    def __init__(self, *a):
        self.beta2 = VariableParam(value=0.999, name='beta2', dtype=graphutils.floatX)
        self.squared_accum = None
        self.momentum = None
        self._pending = None
        self._pending_t = None
        super(Adam, self).__init__(*a)
        self.meta_params['beta2'] = self.beta2

    def _beta2(self):
        return self.beta2.get_value()

    def _ensure_state(self, plan):
        if self.momentum is None:
            n = max(self.model.n_train, 4)
            self.momentum = plan.zeros_flat(n)
            self.squared_accum = plan.zeros_flat(n)
        if self._pending is not None:
            self._apply(self._pending)
            self._pending = None

    def device_update(self, plan):
        m = self.model
        if getattr(plan, '_upd', None) is not None:
            # the update that also writes the packed conv weight images (csrc/update_pack.hip)
            plan.ctx.adam_pack_step(m.P, m.G, self.momentum, self.squared_accum, plan._upd,
                                    self._hyper, **self._grad_scaling(plan))
            return
        plan.ctx.adam_step(m.P[:m.n_train] if m.n_train < m.P.numel() else m.P, m.G,
                           self.momentum, self.squared_accum, m.seg_off, m.seg_reg,
                           self._hyper, **self._grad_scaling(plan))

    @property
    def t(self):
        if self._hyper is None:
            return 0.0 if self._pending_t is None else float(self._pending_t)
        return float(self._hyper[4].item())

    def state_dict(self):
        if self.momentum is None:
            # loaded but not stepped yet (modelload(f) followed by save()): the moments and
            # the step counter are still parked in _pending / _pending_t -- hand them on
            if self._pending is not None and 'm' in self._pending:
                return {'m': np.array(self._pending['m'], dtype=np.float32),
                        's': np.array(self._pending['s'], dtype=np.float32),
                        't': np.array(self.t)}
            return {}
        return {'m': self.momentum.cpu().numpy(), 's': self.squared_accum.cpu().numpy(),
                't': np.array(self.t)}

    def _apply(self, d):
        import torch
        if 'm' in d:
            m = np.ascontiguousarray(d['m'], dtype=np.float32)
            s = np.ascontiguousarray(d['s'], dtype=np.float32)
            if m.size != self.momentum.numel() or s.size != self.momentum.numel():
                raise ValueError("Adam state of %i elements does not fit this model (%i)"
                                 % (m.size, self.momentum.numel()))
            self.momentum.copy_(torch.from_numpy(m))
            self.squared_accum.copy_(torch.from_numpy(s))

    def _apply_pending_hyper(self):
        if self._pending_t is not None:
            self._hyper[4] = float(self._pending_t)
            self._pending_t = None

    def load_state_dict(self, d):
        """m, s and the step counter t.  The device buffers are created lazily by the
        first step; a state loaded before that (``create_model(); modelload(f, model)``)
        is kept and applied when they exist, so resuming continues the bias correction
        and the moments instead of silently restarting them."""
        if self.momentum is None:
            self._pending = {k: d[k] for k in ('m', 's') if k in d} or None
        else:
            self._apply(d)
        if 't' in d:
            self._pending_t = float(np.asarray(d['t']))
            if self._hyper is not None:
                self._apply_pending_hyper()
