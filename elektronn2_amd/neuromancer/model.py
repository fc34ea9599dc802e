"""Model: the graph registry plus the training state
(elektronn2/neuromancer/model.py:39-619, graphmanager.py:192-313).

Kept from the reference: ``designate_nodes`` (model.py:95-227) including the
fov fix-up for UpConv nets (141-152), ``trainingstep(*batch, optimiser=)`` ->
``(loss, t_seconds, None)`` (548-600), ``loss / predict / predict_ext /
gradients``, the ``lr / mom / wd`` properties that act on optimiser globals
(282-322), ``set_opt_meta_params``, ``get/set_param_values``.

Replaced: ``T.grad`` + Theano update lists become a training ``Plan``
(plan.py) over hand-written HIP kernels.  All trainable parameters live in one
flat device arena ``P`` with a same-layout gradient arena ``G`` (so the
optimiser is one fused launch and data-parallel all-reduce is one collective).
``save`` writes parameter values AND optimiser state as ``.npz`` (the
reference pickles class references and drops the Adam state).
"""
from __future__ import annotations

import logging
from collections import OrderedDict

import numpy as np

from . import graphutils, optimiser
from .. import config

logger = logging.getLogger('elektronn2log')

__all__ = ['Model', 'modelload', 'params_from_model_file']


class _CircularBuffer(object):
    def __init__(self, n):
        self.n, self.buf = int(n), []

    def append(self, v):
        self.buf.append(float(v))
        if len(self.buf) > self.n:
            self.buf.pop(0)

    def mean(self):
        return float(np.mean(self.buf)) if self.buf else 0.0


class GraphManager(object):
    """graphmanager.py:192-313, reduced to the registry."""

    def __init__(self, name=""):
        self.name = name
        self.nodes = OrderedDict()
        self.node_descriptors = OrderedDict()

    def __repr__(self):
        return repr(list(self.nodes.keys()))

    def __getitem__(self, sl):
        if isinstance(sl, str):
            return self.nodes[sl]
        return list(self.nodes.values())[sl]

    def reset(self):
        self.nodes = OrderedDict()
        self.node_descriptors = OrderedDict()

    def register_node(self, node, name, args, kwargs):
        self.node_descriptors[name] = (node.__class__.__name__, args, kwargs)
        self.nodes[name] = node
        node._model = self

    @property
    def sources(self):
        return [n for n in self.nodes.values() if n.is_source]

    @property
    def sinks(self):
        return [n for n in self.nodes.values() if len(n.children) == 0]

    @property
    def node_count(self):
        return len(self.nodes)


class Model(GraphManager):
    def __init__(self, name=""):
        super(Model, self).__init__(name=name)
        self.batch_size = None
        self.ndim = None
        self._desig_descr = dict()
        self.iterations = 0
        self.elapsed_time = 0
        self._last_exec_times = _CircularBuffer(config.time_per_step_smoothing_length)
        self._last_losses = _CircularBuffer(config.loss_smoothing_length)
        self.prediction_node = None
        self.prediction_ext = None
        self._prediction_ext_func = None
        self.loss_node = None
        self.target_node = None
        self.error_node = None
        self.input_node = None
        self.trainable_params = None
        self.nontrainable_params = None
        self._grad_func = None
        self.optimisers = dict()
        self.debug_outputs = []
        # device state
        self.P = self.G = None
        self._slots = None          # id(param) -> (offset, size, shape)
        self._ctx = None
        self._dp_group = None

    # ------------------------------------------------------------------ designate
    def designate_nodes(self, input_node='input', target_node=None, loss_node=None,
                        prediction_node=None, prediction_ext=None, error_node=None,
                        debug_outputs=None):
        def designate(purpose, name):
            if isinstance(name, (list, tuple)):
                if purpose not in ['debug_outputs', 'prediction_ext']:
                    raise ValueError("Can only designate several nodes for "
                                     "debug outputs and prediction_ext")
                name = [n if isinstance(n, str) else n.name for n in name]
                setattr(self, purpose, [self.nodes[n] for n in name])
            elif name:
                name = name if isinstance(name, str) else name.name
                setattr(self, purpose, self.nodes[name])
            self._desig_descr[purpose] = name

        designate('input_node', input_node)
        designate('target_node', target_node)
        designate('loss_node', loss_node)
        designate('prediction_node', prediction_node)
        designate('error_node', error_node)
        designate('prediction_ext', prediction_ext)
        designate('debug_outputs', debug_outputs or [])

        if self.prediction_node:
            self.batch_size = self.prediction_node.shape['b']
            self.ndim = self.prediction_node.shape.ndim
            if np.any(np.less(self.prediction_node.shape.fov, 0)):   # UpConvs contained
                in_sh = self.input_node.shape.spatial_shape
                sh = np.array(self.prediction_node.shape.spatial_shape)
                st = np.array(self.prediction_node.shape.strides)
                out_sh = st * (sh - 1) + 1
                diff = np.subtract(in_sh, out_sh)
                if np.any(np.mod(diff, 2)):
                    raise ValueError("FOV is not centered. In_sh=%s, out_sh*strides=%s, "
                                     "diff=%s" % (in_sh, out_sh, diff))
                self.prediction_node.shape._fov = np.array(diff)     # model.py:151-152
                self.target_node.shape._fov = np.array(diff)
            elif not self.prediction_node.shape.fov_all_centered:
                logger.warning("Not all field of views are centered (odd) "
                               "this might cause problems for many setups")

        if self.prediction_ext:
            inp = _inputs_for(self.prediction_ext)
            self._prediction_ext_func = graphutils.make_func(
                inp, list(self.prediction_ext), name='Predictor Extended', model=self)

        if self.loss_node:
            self.trainable_params = list(self.loss_node.all_trainable_params.values())
            self.nontrainable_params = self.loss_node.all_nontrainable_params
            inp = self.loss_node.input_nodes
            self._grad_func = graphutils.make_func(inp, [self.loss_node],
                                                   name='Gradient Func', model=self,
                                                   step='grad')
            extras = list(self.debug_outputs)
            opt_init = (inp, self.loss_node, self.trainable_params, extras, self)
            # AdaGrad / AdaDelta are not used by any BASELINE config
            self.optimisers = dict(SGD=optimiser.SGD(*opt_init),
                                   Adam=optimiser.Adam(*opt_init))

    # ------------------------------------------------------------------ device arena
    def ensure_arena(self, ctx):
        """Bind every parameter of every node to a slice of the flat arena."""
        if self.P is not None:
            return
        import torch
        self._ctx = ctx
        params, seen = [], set()
        for node in self.nodes.values():
            for p in node.params.values():
                if id(p) not in seen and not getattr(p, 'constant', False):
                    seen.add(id(p))
                    params.append(p)
        # trainable first, so the optimiser acts on one contiguous prefix
        params.sort(key=lambda p: 0 if p.apply_train else 1)
        self._slots = {}
        off = 0
        for p in params:
            n = int(np.prod(p.shape)) if len(p.shape) else 1
            self._slots[id(p)] = (off, n, tuple(p.shape) if len(p.shape) else (1,))
            off += (n + 3) // 4 * 4              # keep 16-byte alignment per tensor
        self.n_arena = off
        self.n_train = max([self._slots[id(p)][0] + (self._slots[id(p)][1] + 3) // 4 * 4
                            for p in params if p.apply_train] or [0])
        self.P = torch.zeros(max(off, 4), dtype=torch.float32, device=ctx.device)
        self.G = torch.zeros(max(self.n_train, 4), dtype=torch.float32, device=ctx.device)
        seg_off, seg_reg = [], []
        for p in params:
            o, n, sh = self._slots[id(p)]
            p.bind(self.P[o:o + n].view(sh))
            if p.apply_train:
                seg_off.append(o)
                r = p.apply_reg
                seg_reg.append(float(r) if (r and r is not True) else (1.0 if r else 0.0))
        seg_off.append(self.n_train)
        self.seg_off = torch.tensor(seg_off, dtype=torch.int64, device=ctx.device)
        self.seg_reg = torch.tensor(seg_reg or [0.0], dtype=torch.float32, device=ctx.device)
        self._param_list = params

    def device_param(self, p):
        return p._dev

    def device_grad(self, p):
        o, n, sh = self._slots[id(p)]
        return self.G[o:o + n].view(sh)

    def device_grads_list(self):
        return [self.device_grad(p) for p in self.trainable_params]

    # ------------------------------------------------------------------ data parallel
    def enable_data_parallel(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._dp_group = group if group is not None else dist.group.WORLD
        self.broadcast_params()

    def dp_world(self):
        if self._dp_group is None:
            return 1
        import torch.distributed as dist
        return dist.get_world_size(self._dp_group)

    def broadcast_params(self):
        """rank 0's parameters -> everyone (replicas start identical)."""
        import torch.distributed as dist
        if self.P is None:
            from .plan import get_ctx
            self.ensure_arena(get_ctx())
        dist.broadcast(self.P, src=0, group=self._dp_group)

    def allreduce_grads(self):
        from ..parallel import allreduce_mean_
        allreduce_mean_(self.G, self._dp_group)

    def grad_exchange(self):
        """sliced exchange of the gradient arena (parallel.BucketedMean)"""
        from ..parallel import BucketedMean
        return BucketedMean(self.G, self._dp_group)

    # ------------------------------------------------------------------ functions
    def save(self, file_name):
        """Parameter values + optimiser state as .npz."""
        d = OrderedDict()
        for name, node in self.nodes.items():
            for k, v in node.get_param_values().items():
                d["p/%s/%s" % (name, k)] = v
        for oname, opt in self.optimisers.items():
            for k, v in opt.state_dict().items():
                d["o/%s/%s" % (oname, k)] = v
        d["meta/iterations"] = np.array(self.iterations)
        np.savez(file_name, **d)

    def load(self, file_name):
        z = np.load(file_name)
        for key in z.files:
            parts = key.split('/')
            if parts[0] == 'p':
                self.nodes[parts[1]].params[parts[2]].set_value(z[key])
        st = {}
        for key in z.files:
            parts = key.split('/')
            if parts[0] == 'o':
                st.setdefault(parts[1], {})[parts[2]] = z[key]
        for oname, s in st.items():
            if oname in self.optimisers:
                self.optimisers[oname].load_state_dict(s)
        if "meta/iterations" in z.files:
            self.iterations = int(z["meta/iterations"])

    def loss(self, *args, **kwargs):
        return self.loss_node(*args, **kwargs)

    def gradients(self, *args, **kwargs):
        return self._grad_func(*args, **kwargs)

    def predict(self, *args, **kwargs):
        return self.prediction_node(*args, **kwargs)

    def predict_dense(self, raw_img, as_uint8=False, pad_raw=False):
        """model.py:658-713 (without MFP): dense prediction of a whole volume"""
        return self.prediction_node.predict_dense(raw_img, as_uint8=as_uint8, pad_raw=pad_raw)

    def predict_ext(self, *args, **kwargs):
        return self._prediction_ext_func(*args, **kwargs)

    def paramstats(self):
        print("Parameter statistics")
        for k, W in self.loss_node.all_trainable_params.items():
            W = W.get_value()
            print("Param %s:\tshape=%s,\tmean=%f,\tstd=%f,\tmedian(abs)=%f"
                  % (k, W.shape, W.mean(), W.std(), np.median(np.abs(W))))

    def gradstats(self, *args, **kwargs):
        grads = self.gradients(*args, **kwargs)
        print("Gradient statistics")
        for g in grads:
            print("\tshape=%s,\tmean=%f,\tstd=%f,\tmedian(abs)=%f"
                  % (g.shape, np.mean(g), np.std(g), np.median(np.abs(g))))

    def set_opt_meta_params(self, opt_name, value_dict):
        self.optimisers[opt_name].set_opt_meta_params(value_dict)

    lr = property(lambda self: optimiser.Optimiser.global_lr.get_value(),
                  lambda self, v: optimiser.Optimiser.setlr(v))
    mom = property(lambda self: optimiser.Optimiser.global_mom.get_value(),
                   lambda self, v: optimiser.Optimiser.setmom(v))
    wd = property(lambda self: optimiser.Optimiser.global_weight_decay.get_value(),
                  lambda self, v: optimiser.Optimiser.setwd(v))

    @property
    def mixing(self):
        return self.loss_node.mixing_weights.get_value()

    def get_param_values(self, skip_const=False, as_list=False):
        p_dict = OrderedDict()
        for name, node in self.nodes.items():
            p_dict[name] = node.get_param_values(skip_const)
        return list(p_dict.values()) if as_list else p_dict

    def set_param_values(self, value_dict, skip_const=False):
        if isinstance(value_dict, dict):
            for k, v in value_dict.items():
                if k not in self.nodes:
                    raise KeyError("Graph Manager has no node %s" % (k,))
                self.nodes[k].set_param_values(v, skip_const)
        else:
            for p, n in zip(value_dict, self.nodes.values()):
                n.set_param_values(p, skip_const)

    @property
    def time_per_step(self):
        return self._last_exec_times.mean() + 1e-6

    @property
    def loss_smooth(self):
        return self._last_losses.mean()

    def trainingstep(self, *args, **kwargs):
        """One optimiser iteration: ``trainingstep(data, target, optimiser='Adam')``
        -> ``(loss, t, None)`` (model.py:548-600).  ``t`` = device seconds
        (HIP events around the step)."""
        opt_name = kwargs.get('optimiser', 'SGD')
        if opt_name not in self.optimisers:
            logger.warning("No optimiser '%s'. Falling back to SGD" % (opt_name,))
            opt_name = 'SGD'
        ret = self.optimisers[opt_name](*args)
        loss = ret[0]
        if kwargs.get('update_loss', False):
            loss = self.loss(*args)
        t = self.optimisers[opt_name].last_exec_time
        self.elapsed_time += t
        self._last_exec_times.append(t + 1e-10)
        self._last_losses.append(loss)
        self.iterations += 1
        if len(ret) > 1:
            return loss, t, ret[1:]
        return loss, t, None

    def test_run_prediction(self):
        self.prediction_node.test_run()


def _inputs_for(nodes):
    inp, seen = [], set()
    for n in nodes:
        for s in n.input_nodes:
            if id(s) not in seen:
                seen.add(id(s))
                inp.append(s)
    return inp


def params_from_model_file(file_name):
    z = np.load(file_name)
    out = OrderedDict()
    for key in z.files:
        parts = key.split('/')
        if parts[0] == 'p':
            out.setdefault(parts[1], OrderedDict())[parts[2]] = z[key]
    return out


def modelload(file_name, model):
    """Load parameter values (and optimiser state) saved by ``Model.save`` into
    an already constructed ``model`` (the reference re-executes pickled ctors,
    model.py:623-729; here the config's ``create_model()`` rebuilds the graph)."""
    model.load(file_name)
    return model
