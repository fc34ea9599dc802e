"""Node protocol, model registry and the structural nodes.

Mirrors elektronn2/neuromancer/node_basic.py: ``ModelContainer`` /
``model_manager`` (:52-156), ``choose_name`` (:160-196), automatic
registration of every node in the current model (``MetaNode`` :200-302),
``Node`` (:307-574 -- ``_make_output/_calc_shape/_calc_comp_cost``, ``params``,
``shape``, callable on numpy arrays), ``Input`` (:1182-1243), ``Input_like``
(:1246-1271), ``Concat`` (:1403-1451), ``Add`` (:1455-1483).

There is no symbolic tensor engine here: ``node.output`` is a small ``Sym``
handle (dtype + owner) and the numerics run through a static launch plan
(``plan.py``) over libe2hip.so.  Each node contributes ``_plan_fwd`` /
``_plan_bwd`` hooks instead of a Theano expression.
"""
from __future__ import annotations

import inspect
import logging
import re
import uuid
from collections import OrderedDict
from functools import reduce

import os

import numpy as np

from . import graphutils
from .graphutils import TaggedShape, floatX

logger = logging.getLogger('elektronn2log')

__all__ = ['Node', 'Input', 'Input_like', 'Concat', 'Add', 'model_manager',
           'choose_name', 'Sym']


class Sym(object):
    """Stand-in for the Theano variable a node's ``output`` used to be."""

    def __init__(self, owner, dtype=floatX):
        self.owner = owner
        self.dtype = dtype
        self.ndim = None

    def __repr__(self):
        return "<Sym of %s (%s)>" % (getattr(self.owner, 'name', '?'), self.dtype)


class ModelContainer(object):
    """node_basic.py:52-153."""
    _count = 0

    def __init__(self):
        if ModelContainer._count > 0:
            raise RuntimeError("There may exist only one ModelContainer "
                               "(which may contain several models)")
        ModelContainer._count += 1
        self._default_set = False
        self.models = OrderedDict()
        self.last = None
        self.current = None

    def __getitem__(self, item):
        return self.models[item]

    def __repr__(self):
        return repr(list(self.models.keys()))

    def setdefault(self):
        if self._default_set:
            raise RuntimeError("The default model has already been set.")
        from .model import Model
        self.models["default"] = Model(name="default")
        self.current = self["default"]
        self.last = self["default"]
        self._default_set = True

    def newmodel(self, name):
        from .model import Model
        if name in self.models:
            raise ValueError("Model of the same name %s exists already." % (name,))
        elif name is None:
            name = str(uuid.uuid4())
        self.models[name] = Model(name=name)
        self.last = self.current
        self.current = self[name]
        return self[name]

    def getmodel(self, *args):
        if len(args) == 1:
            current = self[args[0]]
        elif len(args) == 0:
            current = self["default"]
        else:
            raise ValueError("Either provide name or nothing!")
        self.last = self.current
        self.current = current
        return current

    def togglemodel(self):
        self.last, self.current = self.current, self.last
        return self.current

    def reset(self):
        """(new) forget all models -- lets tests build several graphs."""
        self.models = OrderedDict()
        self.last = self.current = None
        self._default_set = False


model_manager = ModelContainer()


def choose_name(proposal, names):
    """node_basic.py:160-196: "conv", "conv1", "conv2", ..."""
    if proposal in names:
        numbers = re.findall(r'(\d+)$', proposal)
        if not len(numbers):
            proposal = proposal + str(1)
        while proposal in names:
            proposal = re.sub(r'(\d+)$', lambda x: str(int(x.group(0)) + 1), proposal)
    return proposal


class MetaNode(type):
    """Registers every node in ``model_manager.current`` under a unique name
    BEFORE ``__init__`` runs, then finalises it (node_basic.py:230-300)."""

    def __call__(cls, *args, **kwargs):
        if model_manager.current is None:
            model_manager.setdefault()
        default_name = ''
        try:
            sig = inspect.signature(cls.__init__)
            d = sig.parameters['name'].default
            if isinstance(d, str):
                default_name = d
        except Exception:
            pass
        # positional ``name``?  find its index in the signature
        try:
            names = list(inspect.signature(cls.__init__).parameters)[1:]
            if 'name' in names and len(args) > names.index('name'):
                args = list(args)
                kwargs['name'] = args.pop(names.index('name'))
                args = tuple(args)
        except Exception:
            pass
        name = kwargs.get('name', default_name)
        name = choose_name(name, model_manager.current.node_descriptors.keys())
        kwargs['name'] = name
        node = cls.__new__(cls)
        model_manager.current.register_node(node, name, args, kwargs)
        node.__init__(*args, **kwargs)
        node._finalize_init()
        return node


class Node(object, metaclass=MetaNode):
    """Basic node (node_basic.py:307-574)."""

    def __init__(self, parent, name="", print_repr=False):
        self.parent = parent
        self.children = OrderedDict()
        self.name = name
        self._features_names = None
        self.params = OrderedDict()
        self.computational_cost = 0
        self.is_source = False
        self.output = None
        self.shape = None
        self._output_func = None
        self._debug_outputs = []
        self._local_exec_time = None
        self._total_exec_time = None
        self._finalized = False
        self._print_repr = print_repr

    # ---- protocol ---------------------------------------------------------
    def _make_output(self):
        self.output = Sym(self, self._parents()[0].output.dtype if self._parents() else floatX)

    def _calc_shape(self):
        self.shape = self._parents()[0].shape.copy()

    def _calc_comp_cost(self):
        self.computational_cost = self._parents()[0].shape.stripnone_prod

    def _parents(self):
        if self.parent is None:
            return []
        if isinstance(self.parent, (list, tuple)):
            return list(self.parent)
        return [self.parent]

    def _finalize_init(self):
        if self._finalized:
            return
        self._make_output()
        self._calc_shape()
        self._calc_comp_cost()
        for p in self._parents():
            p._register_child(self)
        self._output_func = graphutils.make_func(self.input_nodes, self, name=self.name)
        if self._print_repr:
            logger.info("-" * 87)
            logger.info(self)
        self._finalized = True

    def _register_child(self, child):
        self.children[child.name] = child

    def __repr__(self):
        if self.name == '':
            s = "<%s-Node>\n" % (self.__class__.__name__,)
        else:
            s = "<%s-Node> '%s' \n" % (self.__class__.__name__, self.name)
        s += '  '
        if self.param_count > 0:
            s += "#Params={0:,d} ".format(self.param_count)
        if self.computational_cost > 0:
            s += "Comp.Cost=%.3g, " % (float(self.computational_cost),)
        s += "Out:%s" % (str(self.shape))
        if len(self.input_nodes) > 1:
            s += "\n  Order of sources=%s, " % (str([n.name for n in self.input_nodes]))
        return s

    def __call__(self, *args):
        """Compute the node's output for numpy inputs (first call "compiles"); a call
        without arguments only compiles (node_basic.py:464-481, graphutils.py:353-354)."""
        if len(args) == 0:
            return self._output_func()
        if len(args) != len(self.input_nodes):
            raise TypeError("%s: %i inputs required, %i were given."
                            % (self.name, len(self.input_nodes), len(args)))
        try:
            return self._output_func(*args)
        except TypeError as e:
            add_info = '\nShapes (required - given):\n'
            for ni, ar in zip(self.input_nodes, args):
                add_info += " %s %s\t%s %s\n" % (tuple(ni.shape.shape), ni.dtype,
                                                 getattr(ar, 'shape', '?'),
                                                 getattr(ar, 'dtype', '?'))
            raise TypeError(str(e) + add_info)

    # ---- parameters -------------------------------------------------------------
    def get_param_values(self, skip_const=False):
        p_dict = OrderedDict()
        for k, v in self.params.items():
            if v.constant and skip_const:
                continue
            p_dict[k] = v.get_value()
        return p_dict

    def set_param_values(self, value_dict, skip_const=False):
        for k, v in value_dict.items():
            if k not in self.params:
                if k in ['gamma', 'std', 'mean']:
                    continue
                raise KeyError("Layer has no parameter %s" % (k,))
            if self.params[k].constant:
                if skip_const:
                    continue
                raise ValueError("Cannot set value of constant parameter %s in node %s"
                                 % (k, self.name))
            self.params[k].set_value(v)

    # ---- graph queries --------------------------------------------------------------
    @property
    def all_parents(self):
        parents = OrderedDict()
        for node in self._parents():
            parents.update(node.all_parents)
        parents[self.name] = self
        return parents

    @property
    def input_nodes(self):
        return [n for n in self.all_parents.values() if n.is_source]

    @property
    def input_tensors(self):
        return [s.output for s in self.input_nodes]

    @property
    def all_params(self):
        p = OrderedDict()
        for node in self.all_parents.values():
            for k, v in node.params.items():
                p[str(node.name) + '_' + k] = v
        return p

    @property
    def all_trainable_params(self):
        return OrderedDict((k, v) for k, v in self.all_params.items() if v.apply_train)

    @property
    def all_nontrainable_params(self):
        return OrderedDict((k, v) for k, v in self.all_params.items() if not v.apply_train)

    @property
    def all_extra_updates(self):
        return [p.updates for n in self.all_parents.values()
                for p in n.params.values() if p.updates]

    @property
    def param_count(self):
        return int(sum(int(np.prod(x.shape)) for x in self.params.values() if x.apply_train))

    @property
    def all_params_count(self):
        return int(sum(int(np.prod(x.shape)) for x in self.all_trainable_params.values()))

    @property
    def all_computational_cost(self):
        return reduce(lambda x, y: x + y.computational_cost, self.all_parents.values(), 0)

    @property
    def all_children(self):
        children = OrderedDict()
        for child in self.children.values():
            children[child.name] = child
            children.update(child.all_children)
        return children

    @property
    def last_exec_time(self):
        return self._output_func.last_exec_time

    @property
    def feature_names(self):
        return self._features_names

    @feature_names.setter
    def feature_names(self, value):
        if len(value) != self.shape['f']:
            raise ValueError("Feature names must match feature count")
        self._features_names = tuple(value)

    # ---- self test (node_basic.py:1014-1063) -------------------------------------
    def test_run(self, on_shape_mismatch='warn', debug_outputs=False, warmup=False):
        """Random input, check computed vs declared shape, report MPix/s."""
        import time
        inp = []
        for n in self.input_nodes:
            sh = [1 if s is None else s for s in n.shape.shape]
            inp.append(np.random.rand(*sh).astype(n.dtype) if 'float' in n.dtype
                       else np.random.randint(0, 2, sh).astype(n.dtype))
        if warmup:
            self(*inp)
        t0 = time.time()
        y = self(*inp)
        t = time.time() - t0
        declared = [1 if s is None else s for s in self.shape.shape]
        if list(y.shape) != list(declared):
            msg = "Node %s: declared shape %s != computed %s" % (self.name, declared, y.shape)
            if on_shape_mismatch == 'raise':
                raise ValueError(msg)
            logger.warning(msg)
        n_out = float(np.prod(y.shape[-self.shape.ndim:])) if self.shape.ndim else 1.0
        logger.info("Compute time %.4f s, %.3f MPix/s", t, n_out / max(t, 1e-9) / 1e6)
        return y

    # ---- dense inference (node_basic.py:805-1012) ------------------------------------
    def predict_dense(self, raw_img, as_uint8=False, pad_raw=False, tile_batch=None):
        """Dense (stride-1) prediction of a whole image / volume by block tiling plus
        stride-offset interleaving: a net with output strides ``s`` predicts every
        s-th voxel, so each block is predicted ``prod(s)`` times from inputs shifted
        by all offsets in ``[0, s)`` and the outputs are interleaved.

        raw_img: ``(ch, z, x, y)`` (or ``(ch, x, y)`` for a single z-slice of a net
        with ``z`` patch size 1).  Integer input is scaled by 1/255.  Returns
        ``(n_lab, z, x, y)`` of extent ``raw - 2*offsets`` (``pad_raw`` mirrors the
        borders first so that the full image domain is predicted).

        Nets with UpConvs (U-Nets): the reference refuses a node whose field of view is
        unknown (node_basic.py:899-902); the PREDICTION node of a designated model has it
        (``designate_nodes`` derives fov = input - output extent for UpConv nets,
        model.py:141-152), and such nets predict at stride 1: plain tiles that overlap by
        ``input - output`` extent, one pass per tile.

        MI355X design: the volume is uploaded once, tiles are device views, each
        shifted patch is copied into the static input buffer of the captured forward
        graph, and the interleave is a strided device copy -- no host round trip per
        block (the reference pays one per offset).  ``tile_batch`` passes (tile x offset)
        share one launch of the forward graph on the batch axis (None: as many as fit
        ``E2_DENSE_ACT_GIB`` = 48 GiB of activations, at most 16; nets whose input node
        fixes the batch size use that size): small tiles alone leave most of the 256 CUs
        idle, and 288 GB of HBM hold many tiles' activations at once."""
        import time
        import torch
        if self.shape.ndim != 3:
            raise NotImplementedError("predict_dense: the HIP hot path covers 3-d nets")
        if len(self.input_nodes) != 1:
            raise ValueError("predict_dense needs a node with exactly one input")
        offset = np.asarray(self.shape.offsets)
        if np.any(offset < 0):
            raise ValueError("Cannot predict dense because the CNN contains "
                             "UpConvs which cause unknown FOVs. If you use "
                             "UpConvs you should not need predict dense anyway!")
        offset = offset.astype(np.int64)
        raw_img = np.asarray(raw_img)
        scale = 255.0 if raw_img.dtype.kind in 'iu' else 1.0
        strip_z = False
        if raw_img.ndim == 3:
            strip_z = True
            raw_img = raw_img[:, None]
        inp = self.input_nodes[0]
        if raw_img.ndim != 4 or raw_img.shape[0] != inp.shape['f']:
            raise ValueError("predict_dense: raw image must be (ch=%i, z, x, y), got %s"
                             % (inp.shape['f'], raw_img.shape))
        raw = np.ascontiguousarray(raw_img, dtype=np.float32) / np.float32(scale)
        if pad_raw:
            raw = np.pad(raw, [(0, 0)] + [(int(o), int(o)) for o in offset], mode='symmetric')
        n_lab = self.shape['f']
        out_sh = np.asarray(self.shape.spatial_shape, dtype=np.int64)
        ps = np.asarray(inp.shape.spatial_shape, dtype=np.int64)
        strides = np.asarray(self.shape.strides, dtype=np.int64)
        raw_sh = np.asarray(raw.shape[1:], dtype=np.int64)
        tile_sh = ps + strides - 1              # input extent of one block
        prob_sh = out_sh * strides              # dense extent it predicts
        pred_sh = raw_sh - 2 * offset
        if np.any(pred_sh <= 0):
            raise ValueError("predict_dense: image %s is smaller than the field of view"
                             % (tuple(raw_sh),))
        t0 = time.time()
        self()                                  # compile (zero-argument call)
        plan = self._output_func.func
        n_t = [int(-(-int(pred_sh[i]) // int(prob_sh[i]))) for i in range(3)]
        sz, sx, sy = (int(v) for v in strides)
        n_pass = int(np.prod(n_t)) * sz * sx * sy
        b_decl = inp.shape['b']
        if b_decl is not None:
            B = int(b_decl)
        elif tile_batch is not None:
            B = max(1, min(int(tile_batch), n_pass))
        else:
            # activations of one sample: every node's output, conv nodes twice (pre-activation)
            per = sum(4.0 * float(np.prod([1 if v is None else v for v in n.shape.shape]))
                      * (2 if hasattr(n, 'filter_shape') else 1) for n in plan.nodes)
            budget = float(plan.opt['dense_act_gib']) * 2.0 ** 30
            B = int(max(1, min(16, n_pass, budget // max(per, 1.0))))
        x_sh = (B, raw.shape[0]) + tuple(int(v) for v in ps)
        plan.set_inputs([np.zeros(x_sh, np.float32)])      # builds the plan for batch B
        dev = plan.ctx.device
        # the volume, zero-padded on the far side up to a whole number of blocks
        need = [int((n_t[i] - 1) * prob_sh[i] + tile_sh[i]) for i in range(3)]
        vol = torch.zeros((raw.shape[0],) + tuple(max(need[i], int(raw_sh[i])) for i in range(3)),
                          dtype=torch.float32, device=dev)
        vol[:, :raw_sh[0], :raw_sh[1], :raw_sh[2]] = torch.from_numpy(raw).to(dev)
        dense = torch.zeros((n_lab,) + tuple(int(n_t[i] * prob_sh[i]) for i in range(3)),
                            dtype=torch.float32, device=dev)
        x_buf = plan.input_buffer(inp, None)
        # vol / dense were filled on the current stream; the blocks run on the plan's
        plan.stream.wait_stream(torch.cuda.current_stream(dev))
        vol.record_stream(plan.stream)
        dense.record_stream(plan.stream)
        passes = []                              # (input view, output view) of every pass
        for zt in range(n_t[0]):
            for xt in range(n_t[1]):
                for yt in range(n_t[2]):
                    z0, x0, y0 = zt * int(prob_sh[0]), xt * int(prob_sh[1]), yt * int(prob_sh[2])
                    block = dense[:, z0:z0 + int(prob_sh[0]), x0:x0 + int(prob_sh[1]),
                                  y0:y0 + int(prob_sh[2])]
                    for oz in range(sz):
                        for ox in range(sx):
                            for oy in range(sy):
                                passes.append((vol[:, z0 + oz:z0 + oz + int(ps[0]),
                                                   x0 + ox:x0 + ox + int(ps[1]),
                                                   y0 + oy:y0 + oy + int(ps[2])],
                                               block[:, oz::sz, ox::sx, oy::sy]))
        for i0 in range(0, len(passes), B):
            chunk = passes[i0:i0 + B]            # (a short last chunk leaves stale slots: ignored)
            with torch.cuda.stream(plan.stream):
                for i, (src, _) in enumerate(chunk):
                    x_buf[i].copy_(src, non_blocking=True)
            plan.run()
            with torch.cuda.stream(plan.stream):
                for i, (_, dst) in enumerate(chunk):
                    dst.copy_(plan.out[self][i], non_blocking=True)
        plan.stream.synchronize()
        pred = dense[:, :int(pred_sh[0]), :int(pred_sh[1]), :int(pred_sh[2])]
        if as_uint8:
            pred = (pred * 255.0).to(torch.uint8)
        pred = pred.cpu().numpy()
        dt = max(time.time() - t0, 1e-9)
        logger.info(" Inference speed: %.3f MPix/s, %i blocks x %i offsets, %i per launch, %.2f s",
                    float(np.prod(pred_sh)) / 1e6 / dt, int(np.prod(n_t)), sz * sx * sy, B, dt)
        if strip_z:
            pred = pred[:, 0]
        return pred

    # ---- plan hooks (device execution); overridden by compute nodes ----------------
    def _plan_out_shape(self, batch):
        return tuple(batch if s is None else s for s in self.shape.shape)

    def _plan_alloc(self, plan):
        plan.alloc_out(self)

    def _plan_fwd(self, plan):
        raise NotImplementedError("%s has no HIP forward" % self.__class__.__name__)

    def _plan_bwd(self, plan):
        raise NotImplementedError("%s has no HIP backward" % self.__class__.__name__)


class Input(Node):
    """Source node (node_basic.py:1182-1243)."""

    def __init__(self, shape, tags, strides=None, fov=None, dtype=floatX,
                 hardcoded_shape=False, name='input', print_repr=True):
        super(Input, self).__init__(None, name, print_repr)
        self.is_source = True
        self._shape = TaggedShape(shape, tags, strides, fov=fov)
        if not isinstance(dtype, str):
            raise ValueError("dtype must be a string.")
        self.dtype = dtype
        self.hardcoded_shape = hardcoded_shape
        self._local_exec_time = 0

    def _make_output(self):
        self.output = Sym(self, self.dtype)

    def _calc_shape(self):
        self.shape = self._shape

    def _calc_comp_cost(self):
        self.computational_cost = 0

    def _plan_fwd(self, plan):
        pass

    def _plan_bwd(self, plan):
        pass


def Input_like(ref, dtype=None, name='input', print_repr=True, override_f=False,
               hardcoded_shape=False):
    """node_basic.py:1246-1271."""
    if isinstance(ref, Node):
        shape = list(ref.shape.shape)
        tags = ref.shape.tags
        strides = ref.shape.strides
        fov = ref.shape.fov
        if override_f:
            shape[ref.shape.tag2index('f')] = override_f
        if dtype is None:
            dtype = ref.output.dtype
    elif isinstance(ref, TaggedShape):
        shape, tags, strides, fov = ref.shape, ref.tags, ref.strides, ref.fov
        assert dtype is not None
    else:
        raise ValueError("ref must be Node or TaggedShape.")
    node = Input(shape, tags, strides, fov=fov, dtype=dtype, name=name,
                 print_repr=print_repr, hardcoded_shape=hardcoded_shape)
    if isinstance(ref, Node) and getattr(node, '_model', None) is not None:
        # described by its reference, so that a graph rebuilt with another patch or batch
        # size (modelload) gets targets of the new shape
        node._model.node_descriptors[node.name] = (
            'Input_like', (ref,), dict(dtype=dtype, print_repr=print_repr,
                                       override_f=override_f,
                                       hardcoded_shape=hardcoded_shape))
    return node


class Concat(Node):
    """Channel concat (node_basic.py:1403-1451).  Device side: each parent's
    output is copied into its channel slice of one buffer; the gradient of a
    slice is a zero-copy view of the concat gradient."""

    def __init__(self, parent_nodes, axis='f', name="concat", print_repr=True):
        super(Concat, self).__init__(parent_nodes, name, print_repr)
        if not isinstance(parent_nodes, (tuple, list)):
            raise ValueError("Can only join list/tuple of nodes")
        self.axis = (parent_nodes[0].shape.tag2index(axis) if isinstance(axis, str)
                     else axis)

    def _calc_shape(self):
        joint = reduce(lambda x, y: x + y.shape[self.axis], self.parent, 0)
        self.shape = self.parent[0].shape.updateshape(self.axis, joint)

    def _calc_comp_cost(self):
        self.computational_cost = 0

    def _plan_alloc(self, plan):
        """A parent that feeds ONLY this concat does not need buffers of its own: an UpConv
        (its GEMM and bias / activation kernels write any strided view) gets its channel slice
        of the concat buffer as output and the slice of the concat gradient as gradient; a
        Crop (a view of ITS parent's output, which others read too) keeps the forward copy
        but takes the gradient slice.  Per U-Net merge that removes one copy launch from the
        forward and two from the backward pass (option concat_alias=False: every parent is copied)."""
        plan.alloc_out(self)
        alias = plan.scratch[self, 'alias'] = {}
        if self.axis != 1 or not plan.opt['concat_alias']:
            return
        out, g = plan.out[self], plan.grad.get(self)
        c0 = 0
        for p in self.parent:
            c = p.shape['f']
            sole = (len(p.children) == 1 and not any(p is o for o in plan.outputs)
                    and sum(1 for q in self.parent if q is p) == 1)
            kind = type(p).__name__
            if sole and kind == 'UpConv' and p in plan.out:
                plan.out[p] = out[:, c0:c0 + c]
                alias[p] = 'out'
                if g is not None and p in plan.grad:
                    plan.grad[p] = g[:, c0:c0 + c]
                    alias[p] = 'out+grad'
            elif sole and kind == 'Crop' and g is not None and p in plan.grad:
                plan.grad[p] = g[:, c0:c0 + c]
                alias[p] = 'grad'
            c0 += c

    def _plan_fwd(self, plan):
        if self.axis != 1:
            raise NotImplementedError("HIP Concat only along the feature axis")
        out = plan.out[self]
        alias = plan.scratch[self, 'alias']
        c0 = 0
        for p in self.parent:
            c = p.shape['f']
            if 'out' not in alias.get(p, ''):         # (an aliased parent wrote its slice itself)
                plan.ctx.copy5(plan.out[p], out[:, c0:c0 + c])
            c0 += c

    def _plan_bwd(self, plan):
        g = plan.grad[self]
        alias = plan.scratch[self, 'alias']
        c0 = 0
        for p in self.parent:
            c = p.shape['f']
            if 'grad' in alias.get(p, ''):
                plan._grad_written.add(id(p))         # its gradient buffer IS this slice
            else:
                plan.add_grad(p, g[:, c0:c0 + c])
            c0 += c


class Add(Node):
    """node_basic.py:1455-1483."""

    def __init__(self, n1, n2, name="add", print_repr=True):
        super(Add, self).__init__((n1, n2), name, print_repr)
        assert list(n1.shape.shape) == list(n2.shape.shape)

    def _calc_comp_cost(self):
        self.computational_cost = 0

    def _plan_fwd(self, plan):
        out = plan.out[self]
        plan.ctx.copy5(plan.out[self.parent[0]], out)
        plan.ctx.copy5(plan.out[self.parent[1]], out, accumulate=True)

    def _plan_bwd(self, plan):
        for p in self.parent:
            plan.add_grad(p, plan.grad[self])
