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

import json
import logging
from collections import OrderedDict

import numpy as np

from . import graphutils, optimiser
from .. import config

logger = logging.getLogger('elektronn2log')

__all__ = ['Model', 'modelload', 'params_from_model_file', 'closest_valid_patch_size',
           'kernel_lists_from_node_descr']


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
        self._img_owner = None    # the plan whose optimiser launch last wrote the conv weight images
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
                if self.target_node is not None:      # (prediction-only designation: the
                    self.target_node.shape._fov = np.array(diff)   # reference needs a target here)
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
        # (+4: a spare slot behind the arena -- the labelled-voxel count of the data-parallel
        # exchange travels there, parallel.BucketedMean)
        self._G_store = torch.zeros(max(self.n_train, 4) + 4, dtype=torch.float32,
                                    device=ctx.device)
        self.G = self._G_store[:max(self.n_train, 4)]
        self.G_spare = self._G_store[max(self.n_train, 4):max(self.n_train, 4) + 1]
        # the gradient arena is all zeros: a training plan whose optimiser kernel clears it
        # (e2_adam_step_ex zero_g) skips the fill launch of the next backward pass while this
        # holds; any other writer of G (a gradient plan) resets it
        self._g_clean = True
        seg_off, seg_reg = [], []
        for p in params:
            o, n, sh = self._slots[id(p)]
            p.bind(self.P[o:o + n].view(sh))
            p._owner = self           # (set_value tells the model that its parameters changed)
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
    def enable_data_parallel(self, group=None, weight_by_labelled=True, exchange_at_world_1=False):
        """replicas + gradient exchange.  ``weight_by_labelled``: combine the
        ranks' gradients as the reference's whole-batch normalisation does when the ranks
        see different numbers of labelled voxels (parallel.BucketedMean); one extra
        one-element all-reduce per step, the plain mean when the counts are equal.
        ``exchange_at_world_1``: run the segmented step and its collectives even in a
        one-rank group (the mean over one rank is the identity) -- the only way to take the
        RCCL path through its paces on a one-GPU box (tests/test_dp_gpu.py)."""
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._dp_group = group if group is not None else dist.group.WORLD
        self._dp_weighted = bool(weight_by_labelled)
        self._dp_force = bool(exchange_at_world_1)
        self.broadcast_params()

    def dp_world(self):
        """ranks that exchange gradients (1: no exchange).  A one-rank group counts as 2
        when the exchange was forced (enable_data_parallel(exchange_at_world_1=True))."""
        if self._dp_group is None:
            return 1
        import torch.distributed as dist
        w = dist.get_world_size(self._dp_group)
        return 2 if (w == 1 and getattr(self, '_dp_force', False)) else w

    def broadcast_params(self):
        """rank 0's parameters -> everyone (replicas start identical)."""
        import torch.distributed as dist
        if self.P is None:
            from .plan import get_ctx
            self.ensure_arena(get_ctx())
        dist.broadcast(self.P, src=0, group=self._dp_group)
        self._img_owner = None
        # the broadcast is ordered on the current stream only; plans run on streams of
        # their own -- a one-time host sync keeps the first step from reading P early
        if self.P.is_cuda:
            import torch
            torch.cuda.current_stream(self.P.device).synchronize()

    def allreduce_grads(self, count=None, raw=False):
        ex = self.grad_exchange(count, raw=raw)
        ex.start(0, self.G.numel())
        ex.finish()

    def grad_exchange(self, count=None, raw=False):
        """sliced exchange of the gradient arena (parallel.BucketedMean); ``count``: this
        rank's labelled-voxel count on the device (weighted mean, see enable_data_parallel);
        ``raw``: sums only -- the plan's kernels normalise (Plan._dp_scale)"""
        from ..parallel import BucketedMean
        if raw:
            return BucketedMean(self.G, self._dp_group, spare=True, raw=True,
                                force=getattr(self, '_dp_force', False))
        return BucketedMean(self.G, self._dp_group,
                            count=count if getattr(self, '_dp_weighted', False) else None,
                            spare=True, force=getattr(self, '_dp_force', False))

    # ------------------------------------------------------------------ functions
    def save(self, file_name):
        """Save the model INCLUDING its training state to ``file_name`` (model.py:229-235).

        The reference pickles ``(node descriptors, designations)`` -- class references,
        ctor args and parameter values -- and drops the optimiser state.  Here one ``.npz``
        written under exactly the given name holds: every parameter value (``p/<node>/
        <param>``), the optimiser state (Adam m, s, t; SGD last_dir: ``o/<opt>/...``), the
        iteration count, and a JSON graph description (class NAME, args, kwargs per node +
        the designations) from which ``modelload`` re-executes the constructors, as
        graphmanager.py:120-189 does, without unpickling code."""
        d = OrderedDict()
        for name, node in self.nodes.items():
            for k, v in node.get_param_values().items():
                d["p/%s/%s" % (name, k)] = v
        for oname, opt in self.optimisers.items():
            for k, v in opt.state_dict().items():
                d["o/%s/%s" % (oname, k)] = v
        d["meta/iterations"] = np.array(self.iterations)
        d["meta/graph"] = np.array(json.dumps(self.serialise()))
        with open(file_name, 'wb') as f:          # (np.savez(name) would append '.npz')
            np.savez(f, **d)

    def serialise(self):
        """JSON-able graph description: [[name, class, args, kwargs], ...] in creation
        order, parents as ``{"__node__": name}``, shared parameters as ``{"__param__":
        [node, key]}``; initial-value arrays are dropped (the values are saved per node)."""
        owner = {}
        for name, node in self.nodes.items():
            for k, p in node.params.items():
                owner.setdefault(id(p), (name, k))

        def enc(v, key=None):
            if hasattr(v, '_finalize_init'):               # a Node
                return {"__node__": v.name}
            if id(v) in owner and hasattr(v, 'get_value'):
                return {"__param__": list(owner[id(v)])}
            if isinstance(v, tuple):
                return {"__tuple__": [enc(x) for x in v]}
            if isinstance(v, list):
                return [enc(x) for x in v]
            if isinstance(v, dict):
                return {"__dict__": {str(k): enc(x) for k, x in v.items()}}
            if isinstance(v, np.ndarray):
                if key in ('w', 'b', 'gamma', 'mean', 'std'):
                    return None
                return {"__nd__": v.tolist(), "dtype": str(v.dtype)}
            if isinstance(v, (np.integer,)):
                return int(v)
            if isinstance(v, (np.floating,)):
                return float(v)
            if isinstance(v, (np.bool_,)):
                return bool(v)
            if v is None or isinstance(v, (bool, int, float, str)):
                return v
            if hasattr(v, 'shape') and hasattr(v, 'tags') and hasattr(v, 'strides'):
                return {"__tagged__": [list(v.shape), ",".join(v.tags)]}
            raise TypeError("Model.save: cannot describe ctor argument %r" % (v,))

        nodes = []
        for name, (cls, args, kwargs) in self.node_descriptors.items():
            nodes.append([name, cls, [enc(a) for a in args],
                          {k: enc(v, k) for k, v in kwargs.items()}])
        return {"nodes": nodes, "desig": self._desig_descr, "name": self.name}

    def load(self, file_name):
        """parameter values + optimiser state + iteration count into THIS graph"""
        z = np.load(file_name, allow_pickle=False)
        for key in z.files:
            parts = key.split('/')
            if parts[0] == 'p':
                if parts[1] not in self.nodes or parts[2] not in self.nodes[parts[1]].params:
                    raise KeyError("model file has parameter %s which this graph lacks" % key)
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

    def predict_dense(self, raw_img, as_uint8=False, pad_raw=False, tile_batch=None):
        """model.py:658-713 (without MFP): dense prediction of a whole volume"""
        return self.prediction_node.predict_dense(raw_img, as_uint8=as_uint8, pad_raw=pad_raw,
                                                  tile_batch=tile_batch)

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
        (HIP events around the step).

        ``sync=False`` (no reference counterpart; the default keeps the reference's
        semantics): the step is submitted and the call returns at once with the loss and
        time of the PREVIOUS step, so the host prepares and queues the next batch while the
        device works (a 1.7 ms step otherwise pays ~0.3 ms of submit + read-back latency
        per iteration).  The first call of a model returns its own loss."""
        opt_name = kwargs.get('optimiser', 'SGD')
        if opt_name not in self.optimisers:
            logger.warning("No optimiser '%s'. Falling back to SGD" % (opt_name,))
            opt_name = 'SGD'
        ret = self.optimisers[opt_name](*args, sync=kwargs.get('sync', True))
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

    def trainingsteps(self, k, optimiser='SGD', ring=None, sync=True):
        """``k`` optimiser iterations with ONE graph launch -> ``(losses[k], t)``, t = device
        seconds of the launch (``sync=False``: the launch is submitted and the losses / time of
        the launch BEFORE it are returned -- ``(None, None)`` the first time -- so that the host
        fills the ring's next slots while the device works: data/batch.py RingFeeder).  ``ring``: a float32 device tensor (n_slots, plan.input_arena.numel())
        of batches in the layout of the training plan's input arena (``plan.input_slices``); step i
        takes slot (steps so far) % n_slots.  Without a ring every step re-reads the inputs of
        the last ``trainingstep``.  The first ``trainingstep`` must have run (it builds the plan).
        No reference counterpart: training/trainer.py:186-194 is one batch, one step, one loss
        read-back per iteration; here the ~19 us the device idles between two launches are paid
        once per k steps (DESIGN finding 55)."""
        if optimiser not in self.optimisers:
            logger.warning("No optimiser '%s'. Falling back to SGD" % (optimiser,))
            optimiser = 'SGD'
        opt = self.optimisers[optimiser]
        plan = opt.step.func
        if plan is None or not plan._built:
            raise RuntimeError("trainingsteps: call trainingstep once first (it builds the plan)")
        if ring is not None and plan._ring is not ring:
            plan.set_input_ring(ring)
        losses, t = opt.steps(k, sync=sync)
        self.iterations += int(k)
        if losses is not None:
            self.elapsed_time += t
            for l in losses:
                self._last_exec_times.append(t / len(losses) + 1e-10)
                self._last_losses.append(l)
        return losses, t

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


def kernel_lists_from_node_descr(nodes):
    """(filter_shapes, pool_shapes, mfp) of the Conv nodes of a serialised graph
    (model.py:870-895; UpConv excluded, as there)."""
    filters, pools, mfps = [], [], []
    for name, cls, args, kwargs in nodes:
        if cls != 'Conv':
            continue
        f = _dec_plain(args[2]) if len(args) > 2 else _dec_plain(kwargs['filter_shape'])
        if len(args) > 3:
            p = _dec_plain(args[3])
        else:
            p = _dec_plain(kwargs.get('pool_shape'))
        filters.append(tuple(f))
        pools.append(tuple(p) if p is not None else tuple(1 for _ in f))
        mfps.append(bool(kwargs.get('mfp', False)))
    return filters, pools, mfps


def closest_valid_patch_size(filters, pools, desired, mfp=None):
    """Per spatial axis the largest valid input extent <= ``desired`` (the smallest valid
    one if ``desired`` is below it): utils/cnncalculator.py:57-103,309-313.  Valid means
    every layer has output and ``(s - k + 1) % p == 0`` (``== 1`` with max-fragment
    pooling), neural.py:746-750."""
    nd = len(filters[0])
    mfp = mfp or [False] * len(filters)
    out = []
    for ax in range(nd):
        def ok(s):
            for f, p, m in zip(filters, pools, mfp):
                s = s - f[ax] + 1
                if s <= 0:
                    return False
                if p[ax] > 1:
                    if (s % p[ax]) != (1 if m else 0):
                        return False
                    s //= p[ax]
            return True
        valid = [s for s in range(2, 5000) if ok(s)]
        if not valid:
            raise ValueError("no valid patch size on axis %i" % ax)
        smaller = [s for s in valid if s <= int(desired[ax])]
        out.append(smaller[-1] if smaller else valid[0])
    return tuple(out)


def _dec_plain(v):
    if isinstance(v, dict):
        if "__tuple__" in v:
            return tuple(_dec_plain(x) for x in v["__tuple__"])
        if "__nd__" in v:
            return np.asarray(v["__nd__"], dtype=v.get("dtype", "float32"))
        if "__dict__" in v:
            return {k: _dec_plain(x) for k, x in v["__dict__"].items()}
    if isinstance(v, list):
        return [_dec_plain(x) for x in v]
    return v


def modelload(file_name, model=None, override_mfp_to_active=False, imposed_patch_size=None,
              imposed_batch_size=None, name=None, **model_load_kwargs):
    """Load a Model saved by ``Model.save`` (model.py:623-729).

    ``model`` given: parameter values, optimiser state and iteration count go into that
    already constructed graph (``create_model(); modelload(f, model)`` -- training resumes
    where it stopped, Adam moments and step counter included).

    ``model`` None: the graph is rebuilt by re-executing the saved constructor calls
    (graphmanager.py:120-189) in a new model ``name``.  ``imposed_batch_size`` /
    ``imposed_patch_size`` change the input node's shape first; the patch size is moved to
    the closest valid one unless the net contains UpConvs (then it is taken as given and
    the constructors raise if it does not fit, model.py:691-713).
    ``override_mfp_to_active`` switches every Conv to max-fragment pooling and inserts
    ``FragmentsToDense`` in front of the prediction node (model.py:658-689); only the
    prediction path is rebuilt in that case (targets and losses have no dense shape).
    ``make_weights_constant=True`` freezes all parameters."""
    if model is not None:
        model.load(file_name)
        return model
    from . import node_basic, neural, loss as loss_mod
    z = np.load(file_name, allow_pickle=False)
    if "meta/graph" not in z.files:
        raise ValueError("%s holds parameters only (no graph description): pass the "
                         "constructed model, modelload(file, model)" % (file_name,))
    descr = json.loads(str(z["meta/graph"]))
    nodes, desig = descr["nodes"], dict(descr["desig"])
    logger.info("Loading model from %s" % file_name)
    changed_input = imposed_patch_size is not None or imposed_batch_size is not None \
        or override_mfp_to_active
    if changed_input and not desig.get('input_node'):
        raise ValueError("To use 'override_mfp_to_active' or 'imposed_patch_size', the "
                         "saved model must have a designated 'input_node'")
    by_name = {n[0]: n for n in nodes}
    if changed_input:
        inp = by_name[desig['input_node']]
        shape = list(_dec_plain(inp[2][0]) if inp[2] else _dec_plain(inp[3]['shape']))
        tags = (inp[2][1] if len(inp[2]) > 1 else inp[3]['tags'])
        tags = tags.split(',') if isinstance(tags, str) else list(tags)
        spatial = [i for i, t in enumerate(tags) if t.strip() in ('z', 'y', 'x')]
        if imposed_batch_size is not None:
            shape[[t.strip() for t in tags].index('b')] = imposed_batch_size
        filters, pools, mfps = kernel_lists_from_node_descr(nodes)
        if override_mfp_to_active:
            mfps = [True] * len(mfps)
            if imposed_patch_size is None:
                imposed_patch_size = [shape[i] for i in spatial]
        if imposed_patch_size is not None:
            if len(imposed_patch_size) != len(spatial):
                raise ValueError("The dimensionality of the model and the imposed "
                                 "patchsize do not match.")
            if any(n[1] == 'UpConv' for n in nodes):
                valid = tuple(int(v) for v in imposed_patch_size)
                logger.warning("Imposed patch size is not failsafe for UpConvs")
            else:
                valid = closest_valid_patch_size(filters, pools, imposed_patch_size, mfps)
            for i, v in zip(spatial, valid):
                shape[i] = int(v)
        if inp[2]:
            inp[2][0] = {"__tuple__": shape}
        else:
            inp[3]['shape'] = {"__tuple__": shape}
    keep = None
    if override_mfp_to_active:
        pred = by_name[desig['prediction_node']]
        # ancestors of the prediction node only
        keep, todo = set(), [pred[0]]
        def parents_of(v, acc):
            if isinstance(v, dict):
                if "__node__" in v:
                    acc.append(v["__node__"])
                for x in v.values():
                    parents_of(x, acc)
            elif isinstance(v, list):
                for x in v:
                    parents_of(x, acc)
        while todo:
            nm_ = todo.pop()
            if nm_ in keep:
                continue
            keep.add(nm_)
            acc = []
            parents_of(by_name[nm_][2], acc)
            parents_of(by_name[nm_][3], acc)
            todo.extend(acc)
        for n in nodes:
            if n[1] == 'Conv' and n[0] in keep:
                n[3]['mfp'] = True
        dense_name = 'to_dense_' + pred[0]
        dense = [dense_name, 'FragmentsToDense', [pred[2][0]], {}]
        pred[2][0] = {"__node__": dense_name}
        nodes.insert(nodes.index(pred), dense)
        keep.add(dense_name)
        desig = dict(input_node=desig['input_node'], prediction_node=desig['prediction_node'])

    new = node_basic.model_manager.newmodel(name)
    classes = {}
    for mod in (node_basic, neural, loss_mod):
        for k in dir(mod):
            v = getattr(mod, k, None)
            if isinstance(v, type) and issubclass(v, node_basic.Node):
                classes[k] = v
    classes['Input_like'] = node_basic.Input_like
    built = {}

    def dec(v):
        if isinstance(v, dict):
            if "__node__" in v:
                return built[v["__node__"]]
            if "__param__" in v:
                return built[v["__param__"][0]].params[v["__param__"][1]]
            if "__tuple__" in v:
                return tuple(dec(x) for x in v["__tuple__"])
            if "__nd__" in v:
                return np.asarray(v["__nd__"], dtype=v.get("dtype", "float32"))
            if "__dict__" in v:
                return {k: dec(x) for k, x in v["__dict__"].items()}
            if "__tagged__" in v:
                from .graphutils import TaggedShape
                return TaggedShape(v["__tagged__"][0], v["__tagged__"][1])
        if isinstance(v, list):
            return [dec(x) for x in v]
        return v

    for nname, cls, args, kwargs in nodes:
        if keep is not None and nname not in keep:
            continue
        if cls not in classes:
            raise ValueError("model file names node class %r, which this build does not "
                             "provide" % (cls,))
        kw = {k: dec(v) for k, v in kwargs.items()}
        kw['name'] = nname
        built[nname] = classes[cls](*[dec(a) for a in args], **kw)
    if desig:
        new.designate_nodes(**{k: v for k, v in desig.items() if v})
    # values: every saved parameter whose node was rebuilt
    for key in z.files:
        parts = key.split('/')
        if parts[0] == 'p' and parts[1] in new.nodes:
            new.nodes[parts[1]].params[parts[2]].set_value(z[key])
    if keep is None:
        st = {}
        for key in z.files:
            parts = key.split('/')
            if parts[0] == 'o':
                st.setdefault(parts[1], {})[parts[2]] = z[key]
        for oname, sd in st.items():
            if oname in new.optimisers:
                new.optimisers[oname].load_state_dict(sd)
        if "meta/iterations" in z.files:
            new.iterations = int(z["meta/iterations"])
    if model_load_kwargs.get('make_weights_constant'):
        for node in new.nodes.values():
            for p_ in node.params.values():
                p_.apply_train = False
    return new
