"""Conv / UpConv / Pool / Crop / AutoMerge nodes with the reference's
constructor signatures, parameter initialisation and shape / stride / fov
bookkeeping (elektronn2/neuromancer/neural.py), executing through libe2hip.so.

  Conv      neural.py:500-859   conv -> max-pool -> +bias -> act   (F3 order)
  UpConv    neural.py:907-1129  F2 closed form, identity_init
  Crop      neural.py:1132-1190 zero-copy view
  AutoMerge neural.py:1282-1407 (= UpConvMerge)
  Pool      neural.py:1409-1559

  Perceptron neural.py:258-410  dot product (+ batch norm) -> +bias -> act  (config 1)

2-D convolutions ('b,f,y,x', BASELINE config 1 = examples/mnist.py) run through the same
kernels with a unit z axis; batch normalisation ('train' / 'predict', neural.py:681-711)
has its own kernel (csrc/dense_bn.hip).

Outside the hot path and therefore rejected with NotImplementedError here:
dropout, gradnet, batch_normalisation='fadeout', activations other than 'relu' / 'lin',
conv modes other than 'valid', 1-D convolutions.
"""
from __future__ import annotations

import logging

import numpy as np

from .graphutils import floatX, TaggedShape
from .. import autotune
from . import bf16_ahead
from .node_basic import Node, Concat, Add, Sym
from .variables import VariableWeight, ConstantParam, VariableParam

logger = logging.getLogger('elektronn2log')

__all__ = ['Conv', 'UpConv', 'Pool', 'Crop', 'AutoMerge', 'UpConvMerge', 'NeuralLayer',
           'FragmentsToDense', 'Perceptron']

_HIP_ACTS = ('relu', 'lin')


class NeuralLayer(Node):
    """Parameter plumbing shared by Conv / UpConv (neural.py:37-256)."""

    def _register_param(self, param, shape, name, init_kwargs=None,
                        apply_train=False, apply_reg=False):
        add_to_params = True
        if self.name == '':
            p_name = '<%s%s>' % (name, tuple(shape))
        else:
            p_name = '<%s_%s%s>' % (self.name, name, tuple(shape))
        if param is None:
            p = VariableWeight(shape=shape, init_kwargs=init_kwargs, name=p_name,
                               apply_train=apply_train, apply_reg=apply_reg)
        elif isinstance(param, np.ndarray):
            if param.shape != tuple(shape):
                if not (param.ndim == 0 and tuple(shape) == (1,)):
                    raise ValueError("Shape mismatch. Required %s, given %s"
                                     % (shape, param.shape))
            p = VariableWeight(value=param, name=p_name, apply_train=apply_train,
                               apply_reg=apply_reg, dtype=floatX)
        elif isinstance(param, (VariableParam, ConstantParam)):   # shared from elsewhere
            if tuple(param.get_value().shape) != tuple(shape):
                raise ValueError("Shape mismatch. Required %s, given %s"
                                 % (shape, param.get_value().shape))
            p = param
            add_to_params = False
        elif isinstance(param, (list, tuple)):
            if not isinstance(param[0], np.ndarray) or param[1] not in ('const', 'trainable'):
                raise ValueError("If a parameter is passed as a list, the first entry "
                                 "must be an np.ndarray and the second 'const' or "
                                 "'trainable'.")
            if param[0].shape != tuple(shape):
                raise ValueError("Shape mismatch. Required %s, given %s"
                                 % (shape, param[0].shape))
            value = np.ascontiguousarray(param[0], dtype=floatX)
            if param[1] == 'const':
                p = ConstantParam(value, p_name)
            else:
                p = VariableWeight(value=value, name=p_name, apply_train=True,
                                   apply_reg=apply_reg)
        else:
            raise ValueError("Parameter %s must be either <np.ndarray>, a shared "
                             "parameter, a tuple or None (to create new param)" % (name,))
        setattr(self, name, p)
        if add_to_params:
            self.params[name] = p

    def _setup_params(self, w_sh, w, b, gamma, mean, std, dropout_rate,
                      pool_shape=None, gradnet_rate=None):
        """neural.py:146-256 (BN / dropout / gradnet branches are out of scope)."""
        from .. import config
        self.w = None
        w_init = dict(scale='glorot', mode='ortho' if config.use_ortho_init else 'normal',
                      pool=pool_shape, spatial_axes=self.spatial_axes)
        self._register_param(w, w_sh, 'w', init_kwargs=w_init, apply_train=True,
                             apply_reg=True)
        act = self.activation_func
        n_f = self.n_f
        self.b = None
        b_sh = (n_f,)
        if act == 'relu' or act.startswith("maxout"):
            norm = 1.0
            if len(w_sh) > 2:
                fov = 1
                for i in self.spatial_axes:
                    fov = fov * w_sh[i]
                norm = fov
            b_init = dict(scale=1.0 / norm, mode='const')
        elif act == 'sigmoid':
            b_init = dict(scale=0.5, mode='const')
        elif act == 'prelu':
            raise NotImplementedError("prelu is outside the HIP hot path")
        else:
            b_init = dict(scale=1e-6, mode='fix-uni')
        self._register_param(b, b_sh, 'b', init_kwargs=b_init, apply_train=True,
                             apply_reg=False)
        # batch normalisation (neural.py:206-241): gamma is trained and carries three times
        # the weight decay; mean / std hold the running statistics ('train') or the
        # statistics to apply ('predict')
        bn = self.batch_normalisation
        self.gamma = self.mean = self.std = None
        if bn == 'train':
            sh = (n_f,)
            self._register_param(gamma, sh, 'gamma', init_kwargs=dict(scale=1.0, mode='const'),
                                 apply_train=True, apply_reg=3.0)
            if mean is not None or std is not None:
                raise ValueError("Cannot pass mean and std for training, they "
                                 "are computed in the theano graph.")
            self._register_param(None, sh, 'mean', init_kwargs=dict(scale=0.0, mode='const'))
            self._register_param(None, sh, 'std', init_kwargs=dict(scale=1.0, mode='const'))
        elif bn == 'predict':
            sh = (n_f,)
            self._register_param(gamma, sh, 'gamma', init_kwargs=dict(scale=1.0, mode='const'))
            self._register_param(mean, sh, 'mean', init_kwargs=dict(scale=0.0, mode='const'))
            self._register_param(std, sh, 'std', init_kwargs=dict(scale=1.0, mode='const'))
        elif bn == 'fadeout':
            raise NotImplementedError("batch_normalisation='fadeout' needs gradnet, which is "
                                      "outside the HIP hot path")
        elif bn is not False and bn is not None:
            raise ValueError("Unknown value %s for batchnormalisation" % (bn,))
        self.dropout_rate = None
        if dropout_rate:
            raise NotImplementedError("dropout is outside the HIP hot path")
        self.gradnet_rate = None
        if gradnet_rate:
            raise NotImplementedError("gradnet is outside the HIP hot path")


class Conv(NeuralLayer):
    """Convolutional layer with subsequent pooling (neural.py:500-859).

    Op order (F3): conv(true convolution, F1) -> max-pool -> + b -> activation.
    """

    def __init__(self, parent, n_f, filter_shape, pool_shape=None,
                 conv_mode='valid', activation_func='relu',
                 mfp=False, batch_normalisation=False, dropout_rate=0,
                 name="conv", print_repr=True, w=None, b=None, gamma=None,
                 mean=None, std=None, gradnet_mode=None, invalidate_fov=False):
        Node.__init__(self, parent, name, print_repr)
        self.n_f = n_f
        self.filter_shape = tuple(filter_shape)
        self.conv_mode = conv_mode
        self.activation_func = activation_func
        self.batch_normalisation = batch_normalisation
        self.gradnet_mode = gradnet_mode
        self.mfp = mfp
        self.strides = parent.shape.strides
        self.mfp_offsets = parent.shape.mfp_offsets
        self.axis = parent.shape.tag2index('f')
        self.axis_order = None
        self.invalidate_fov = invalidate_fov
        if pool_shape is None:
            pool_shape = tuple([1 for _ in filter_shape])
        self.pool_shape = tuple(int(p) for p in pool_shape)
        self.spatial_axes = self.parent.shape.spatial_axes
        conv_dim = len(self.spatial_axes)
        x_dim = len(self.parent.shape)
        if len(self.spatial_axes) != len(filter_shape) or \
                len(filter_shape) != len(self.pool_shape):
            raise ValueError("The filter_shape dimensionality (%i), the number "
                             "of spatial dimensions in the input (%i) and "
                             "the dimensionality of pool_shape (%i) differ! "
                             "Use filter size 1 on axes which should not be "
                             "convolved." % (len(filter_shape), conv_dim,
                                             len(self.pool_shape)))
        n_in = parent.shape['f']
        fail = False
        w_sh = None
        if conv_dim == 1:
            raise NotImplementedError("1-D convolutions are outside the HIP hot path "
                                      "(no reference config uses them)")
        elif conv_dim == 2:
            # config 1 (examples/mnist.py): 'b,f,y,x' -- the 3-D kernels with a unit z axis
            if x_dim != 4:
                fail = True
            if self.spatial_axes == [2, 3]:
                self.axis_order = 'dnn'
                w_sh = [n_f, n_in] + list(filter_shape)
            else:
                fail = True
        elif conv_dim == 3:
            if x_dim != 5:
                fail = True
            if self.spatial_axes == [2, 3, 4]:
                self.axis_order = 'dnn'
                w_sh = [n_f, n_in] + list(filter_shape)
            else:
                fail = True
        else:
            fail = True
        if fail:
            raise NotImplementedError("Cannot convolve non-standard shapes / axis orders. "
                                      "Implement reshaping before conv and "
                                      "re-reshaping after!")
        if conv_mode != 'valid':
            raise NotImplementedError("conv_mode=%r: only 'valid' is on the HIP hot path"
                                      % (conv_mode,))
        if activation_func not in _HIP_ACTS:
            raise NotImplementedError("activation_func=%r: only %s are on the HIP hot path"
                                      % (activation_func, _HIP_ACTS))
        self.conv_dim = conv_dim
        self.w_sh = w_sh
        self._setup_params(w_sh, w, b, gamma, mean, std, dropout_rate, self.pool_shape,
                           1.0 if gradnet_mode else None)

    def _make_output(self):
        self.output = Sym(self, floatX)
        if self.mfp and not all(p == 1 for p in self.pool_shape):
            # max-fragment pooling (neural.py:666-675, computations.py:652-680): pool at
            # every offset inside the pooling window; the fragments go to the batch axis
            if self.input_nodes[0].shape['b'] != 1:
                raise ValueError("For MFP the batchsize of the raw image input must be 1.")
            from itertools import product
            strides = np.array(self.strides, np.int64)
            offsets_new = []
            for ix in product(*[range(p) for p in self.pool_shape]):
                for off in np.array(self.mfp_offsets, np.int64):
                    offsets_new.append(off + np.multiply(ix, strides))
            self.mfp_offsets = np.array(offsets_new)
        self.strides = np.multiply(self.pool_shape, self.strides)

    def _calc_shape(self):
        """neural.py:725-764."""
        sh = self.parent.shape
        for j, (i, f, p) in enumerate(zip(self.spatial_axes, self.filter_shape,
                                          self.pool_shape)):
            k = 1 - f
            s = (sh[i] + k) // p
            if self.mfp:
                if (sh[i] + k - p + 1) % p != 0:
                    raise ValueError("Cannot pool spatial axis '%s' of length %i "
                                     "by factor %i after convolving with "
                                     "kernel of size %i and using MFP."
                                     % (sh.tags[i], sh[i], p, f))
            elif (sh[i] + k) % p != 0:
                raise ValueError("Cannot pool spatial axis '%s' of length %i "
                                 "by factor %i after convolving with "
                                 "kernel of size %i." % (sh.tags[i], sh[i], p, f))
            if s <= 0:
                raise ValueError("Spatial axis '%s' of length %i is too small for "
                                 "kernel %i" % (sh.tags[i], sh[i], f))
            sh = sh.updateshape(i, s)
            if sh.fov[j] > 0 and not self.invalidate_fov:
                fov = sh.fov[j] + (f + p - 2) * sh.strides[j]
            else:
                fov = -1
            sh = sh.updatefov(j, fov)
        if self.mfp:
            sh = sh.updatemfp_offsets(self.mfp_offsets)
            sh = sh.updateshape('b', int(np.prod(self.pool_shape)), mode='mult')
        sh = sh.updatestrides(self.strides)
        sh = sh.updateshape('f', self.n_f)
        self.shape = sh

    def _calc_comp_cost(self):
        sh = self.parent.shape
        n_position = 1
        for i, f, p in zip(self.spatial_axes, self.filter_shape, self.pool_shape):
            n_position *= sh[i] + 1 - f
        b = 1 if sh['b'] is None else sh['b']
        self.computational_cost = int(np.prod(self.w_sh)) * n_position * b

    def __repr__(self):
        s = Node.__repr__(self) + "\n"
        s += "  n_f=%i, " % (self.n_f,)
        s += "%id conv, kernel=%s, pool=%s, " % (self.conv_dim, self.filter_shape,
                                                  self.pool_shape)
        s += "act='%s', " % (self.activation_func,)
        return s

    def make_dual(self, parent, share_w=False, mfp=False, **kwargs):
        """neural.py:780-846 (weight sharing is not supported here)."""
        if share_w or mfp:
            raise NotImplementedError("make_dual(share_w / mfp) is outside the hot path")
        defaults = dict(activation_func=self.activation_func, name=self.name + '.T',
                        print_repr=self._print_repr)
        defaults.update(kwargs)
        if self.w_sh[0] != parent.shape['f']:
            raise ValueError("Cannot make dual layer: input features mismatch")
        return UpConv(parent, self.parent.shape['f'], self.pool_shape, **defaults)

    # ---- device execution ------------------------------------------------------------
    @property
    def _k3(self):
        """kernel as (kz, kx, ky): 2-D layers get a unit z axis"""
        return (1,) * (3 - len(self.filter_shape)) + tuple(int(k) for k in self.filter_shape)

    @property
    def _p3(self):
        return (1,) * (3 - len(self.pool_shape)) + tuple(int(p) for p in self.pool_shape)

    @staticmethod
    def _sp3(shape):
        sp = tuple(int(v) for v in shape.spatial_shape)
        return (1,) * (3 - len(sp)) + sp

    @staticmethod
    def _w5(w):
        """device weights as (n_f, n_in, kz, kx, ky)"""
        return w if w.dim() == 5 else w.reshape(tuple(w.shape[:2]) + (1,) * (5 - w.dim()) +
                                                 tuple(w.shape[2:]))

    def _bn(self):
        return self.batch_normalisation in ('train', 'predict')

    def _fused_first(self, plan):
        """Cin = 1 first layer with a supported kernel/pool: fused conv+pool+bias+act
        kernels that never materialise the conv output (csrc/conv_first.hip)."""
        return (not self._bn() and self.parent.is_source and self.parent.shape['f'] == 1 and
                plan.ctx.conv1_supported(1, self._k3, self._p3) and
                not plan.needs_grad(self.parent) and not self._mfp_pool())

    def _mfp_pool(self):
        return bool(self.mfp) and not all(p == 1 for p in self._p3)

    def _fused_head(self, plan):
        """classifier head: this (1,1,1) 'lin' conv to <= 4 features feeds nothing but a
        Softmax of the same plan -> conv + softmax (+ NLL and all gradients) run in the
        fused head kernels (csrc/head.hip); returns that Softmax node or None."""
        key = (self, 'head')
        if key not in plan.scratch:
            sm = None
            kids = list(self.children.values())
            if (type(self) is Conv and not self._bn() and tuple(self._k3) == (1, 1, 1)
                    and all(p == 1 for p in self._p3) and self.activation_func == 'lin'
                    and len(kids) == 1 and type(kids[0]).__name__ == 'Softmax'
                    and kids[0].n_indep == 1
                    and any(n is kids[0] for n in plan.nodes)
                    and not any(n is self for n in plan.outputs)
                    and plan.ctx.head_supported(self.parent.shape['f'], self.n_f)
                    and not self._fused_first(plan)):
                sm = kids[0]
            plan.scratch[key] = sm
        return plan.scratch[key]

    def _tail(self, plan):
        """The tail of a net in ONE launch (csrc/tail.hip): this (1,1,1) relu conv feeds
        nothing but a fused classifier head under a MultinoulliNLL of a TRAINING plan -- both
        convs have one tap, so forward and backward of the pair (and of the loss) run per
        position tile out of LDS.  Returns (head conv, softmax, nll) or None."""
        key = (self, 'tail')
        if key not in plan.scratch:
            r = None
            kids = list(self.children.values())
            par = self.parent
            if (plan.opt['fuse_tail'] and plan.training
                    and type(self) is Conv and not self._bn() and not self.mfp
                    and tuple(self._k3) == (1, 1, 1) and all(p == 1 for p in self._p3)
                    and self.activation_func == 'relu' and len(kids) == 1
                    and type(kids[0]) is Conv and not isinstance(par, (list, tuple))
                    and not self._fused_first(plan)
                    and (getattr(plan.ctx, 'mfma_dtype', 'f32') == 'f32' or plan.opt['bf16_tail'])
                    and not any(n is self or n is kids[0] for n in plan.outputs)):
                head = kids[0]
                sm = head._fused_head(plan)
                nll = [c for c in (sm.children.values() if sm is not None else [])
                       if type(c).__name__ == 'MultinoulliNLL' and any(n is c for n in plan.nodes)]
                # the parent's output gradient has to come from this node alone (the launch
                # OVERWRITES it, during the forward half of the step)
                users = [c for c in par.children.values()
                         if id(c) in plan._loss_anc or c is plan.loss_node]
                x = plan.out.get(par)
                if (sm is not None and len(nll) == 1 and len(users) == 1 and users[0] is self
                        and x is not None and x.is_contiguous()
                        and plan.ctx.tail_supported(par.shape['f'], self.n_f, head.n_f)):
                    r = (head, sm, nll[0])
            plan.scratch[key] = r
        return plan.scratch[key]

    def _tail_gm(self, plan):
        """the tail launch of this node can carry the activation backward of its PARENT (which
        feeds only this node -- Conv._tail): a plain Conv without pooling / batch norm, relu or
        lin, with gradient buffers of its own.  Returns (mode, src tensor, bias) for
        e2_tail_fwd_bwd or None (the launch writes the parent's plain output gradient)."""
        par = self.parent
        if getattr(plan.ctx, 'mfma_dtype', 'f32') != 'f32':
            # bf16 mode: the parent's own activation backward also writes the bf16 operand images
            # of its two gradient GEMMs (bf16_ahead.py) -- the tail hands it the plain gradient
            return None
        if not (plan.opt['tail_gm'] and type(par) is Conv and not par._bn()
                and all(p == 1 for p in par._p3) and par.activation_func in ('relu', 'lin')
                and (par, 'dy') in plan.scratch and not par._fused_first(plan)
                and par._fused_head(plan) is None and par._tail(plan) is None
                and not par._mfp_pool() and plan.needs_grad(par)):
            return None
        if par.activation_func == 'lin':
            return (3, None, None)
        if par._fused_act(plan):
            return (1, plan.out[par], None)            # activated output, signed zeros
        y = plan.scratch.get((par, 'y'))
        if y is None or not y.is_contiguous():
            return None
        return (2, y, plan.param(par.b))               # pre-activation (+ bias)

    def _fused_act(self, plan):
        """no pooling and a specialised kernel width: bias + activation go into the
        conv kernel's epilogue, the pre-activation is never stored"""
        if not (all(p == 1 for p in self._p3) and self._k3[2] in (1, 3, 4, 5)
                and type(self) is Conv and not self._bn() and not self._fused_first(plan)):
            return False
        # the fused epilogue cannot split K: only where the output alone yields enough
        # work-groups to fill the chip (small late layers keep split-K + pointwise pass)
        osp = self._sp3(self.shape)
        tiles = plan.batch * osp[0] * (-(-(osp[1] * osp[2]) // 128)) * (-(-self.n_f // 112))
        return tiles >= 160

    _PART_WINDOWS = ((1, 1, 1), (1, 2, 2), (2, 1, 1), (2, 2, 2))

    def _parts_ok(self, plan):
        """this node's pre-activation / output gradient may arrive as split-K partial sums:
        its bias + activation (+ pooling) kernels add them up (e2hip.h, "split-K without
        atomics")"""
        return (type(self) is Conv and not self._bn() and not self._mfp_pool()
                and tuple(self._p3) in self._PART_WINDOWS and not self._fused_first(plan)
                and self._fused_head(plan) is None and not plan.fuse_actbwd)

    @staticmethod
    def _n_parts(shape):
        """slabs to provide for a tensor of this shape: 8, while they stay small (the layers
        that need split-K to fill the chip ARE small)"""
        nbytes = 4 * int(np.prod(shape))
        return 8 if 8 * nbytes <= (512 << 20) else 1

    def _plan_alloc(self, plan):
        N = plan.out_shape(self.parent)[0]     # (the fragments of MFP sit on the batch axis)
        psp = self._sp3(self.parent.shape)
        k = self._k3
        if self.mfp and plan.training:
            raise NotImplementedError("MFP is a prediction-time rewrite of the net "
                                      "(neural.py:531-533); train without it")
        if self._fused_head(plan) is not None:
            plan.out[self] = None             # the logits are never materialised
            return
        if self._fused_first(plan):
            plan.alloc_out(self)
            if plan.training:
                nb = plan.ctx.conv1_bwd_ws_bytes(plan.out_shape(self), k)
                plan.scratch[self, 'ws1'] = plan.empty_flat(nb // 4 + 16)
            return
        if self._tail(plan) is not None:
            # neither the activations nor their gradient are materialised; what the launch
            # leaves behind is `dy` (the pre-activation's gradient: this layer's weight
            # gradient reads it, 128 B of slack behind it as for any padded gradient) and the
            # parent's output gradient
            head = self._tail(plan)[0]
            plan.out[self] = None
            cin = self.parent.shape['f']
            nb = plan.ctx.conv_ws_bytes(self.n_f, cin, k)
            plan.scratch[self, 'wp_f'] = plan.zeros_flat(nb // 4 + 64)
            plan.pack_jobs.append((self.w, plan.scratch[self, 'wp_f'], 0))
            ysh = (N, self.n_f) + tuple(psp)
            flat = plan.zeros_flat(int(np.prod(ysh)) + 32)
            plan.scratch[self, 'dy'] = flat[:int(np.prod(ysh))].view(ysh)
            plan.scratch[self, 'dy_pad'] = plan.scratch[self, 'dy']
            if plan.needs_grad(self.parent):
                plan.scratch[self, 'wp_d'] = plan.zeros_flat(nb // 4 + 64)
                plan.pack_jobs.append((self.w, plan.scratch[self, 'wp_d'], 1))
            wsb = plan.ctx.tail_ws_bytes(plan.out_shape(self.parent), self.n_f, head.n_f)
            plan.scratch[self, 'tail_ws'] = plan.empty_flat(wsb // 4 + 16)
            return
        osp = [psp[i] - k[i] + 1 for i in range(3)]
        if not self._fused_act(plan):
            ysh = (N, self.n_f) + tuple(osp)
            yp = plan.empty(((self._n_parts(ysh) if self._parts_ok(plan) else 1),) + ysh)
            plan.scratch[self, 'y_parts'] = yp
            plan.scratch[self, 'y'] = yp[0]
        plan.alloc_out(self)
        if plan.training and self in plan.grad and self._parts_ok(plan):
            gsh = tuple(plan.grad[self].shape)
            if self._n_parts(gsh) > 1:         # the output gradient, as slabs for partial sums
                gp = plan.empty((self._n_parts(gsh),) + gsh)
                plan.scratch[self, 'grad_parts'] = gp
                plan.grad[self] = gp[0]
        if self._bn():
            pooled = any(p != 1 for p in self._p3)
            if pooled:        # batch norm acts on the POOLED conv output (neural.py:678-681)
                plan.scratch[self, 'lin'] = plan.empty(plan.out_shape(self))
                if plan.training:
                    plan.scratch[self, 'dlin'] = plan.empty(plan.out_shape(self))
            plan.scratch[self, 'bn_save'] = plan.zeros_flat(2 * self.n_f)
        cin = self.parent.shape['f']
        nb = plan.ctx.conv_ws_bytes(self.n_f, cin, k)
        plan.scratch[self, 'wp_f'] = plan.zeros_flat(nb // 4 + 64)   # padding stays zero
        plan.pack_jobs.append((self.w, plan.scratch[self, 'wp_f'], 0))
        plan.pack_nodes[id(plan.scratch[self, 'wp_f'])] = (self, 0)      # (Plan._refine_pack_rows)
        if plan.training:
            pad = [kk - 1 for kk in k]
            # zero-padded output gradient: dgrad runs a plain correlation over it and
            # wgrad fetches it 16 bytes per lane (128 B of slack behind the last element).
            # Rows OVERLAP: the row pitch is the INPUT's width Wo + kw - 1, not Wo + 2 (kw - 1)
            # -- the kw - 1 zeros behind a row's interior are its right border and the next
            # row's left border at once (nothing ever writes them).  The weight gradient's K
            # runs over the memory span of a plane (csrc/conv_wgrad_direct.hip): kw - 1 gap
            # columns per row instead of 2 (kw - 1), 5 % fewer MFMAs on the 37-wide layers.
            pshape = (N, self.n_f) + tuple(osp[i] + 2 * pad[i] for i in range(3))
            pitch = osp[2] + pad[2]
            plane = pshape[3] * pitch
            flat = plan.zeros_flat(N * self.n_f * pshape[2] * plane + pad[2] + 32)
            dyp = flat.as_strided(pshape, (self.n_f * pshape[2] * plane, pshape[2] * plane,
                                           plane, pitch, 1))
            plan.scratch[self, 'dy_pad'] = dyp
            plan.scratch[self, 'dy'] = dyp[:, :, pad[0]:pad[0] + osp[0],
                                           pad[1]:pad[1] + osp[1], pad[2]:pad[2] + osp[2]]
            if plan.needs_grad(self.parent):
                plan.scratch[self, 'wp_d'] = plan.zeros_flat(nb // 4 + 64)
                plan.pack_jobs.append((self.w, plan.scratch[self, 'wp_d'], 1))
                plan.pack_nodes[id(plan.scratch[self, 'wp_d'])] = (self, 1)

    # ---- the tuning keys of the node's three GEMM launches --------------------------------
    def _sig_fwd(self, plan):
        x = plan.out[self.parent]
        cin = self.parent.shape['f']
        if self._fused_act(plan):
            return (2, self.n_f, cin) + tuple(self._k3) + tuple(plan.out[self].shape[2:]) + (x.stride(3),)
        return (0, self.n_f, cin) + tuple(self._k3) + tuple(plan.scratch[self, 'y'].shape[2:]) + (x.stride(3),)

    def _sig_dgrad(self, plan):
        dst = plan.grad[self.parent]
        return (1, self.parent.shape['f'], self.n_f) + tuple(self._k3) + tuple(dst.shape[2:]) + \
            (plan.scratch[self, 'dy_pad'].stride(3),)

    def _sig_wgrad(self, plan):
        x, dy = plan.out[self.parent], plan.scratch[self, 'dy']
        return (self.n_f, self.parent.shape['f']) + tuple(self._k3) + tuple(dy.shape[2:]) + \
            (x.stride(3), dy.stride(3))

    def _need_f32_dy(self, plan):
        """a launch is about to read the f32 gradient of the pre-activation: it must exist"""
        b = plan.bf16a.get((self, 'dy'))
        if b is not None and not b['want_f32'] and plan._dy_ready.get(self):
            raise RuntimeError("bf16 mode: the tiling of a gradient launch of %s changed after its "
                               "operands were planned as bf16 images only (re-build the plan)" % self.name)

    def _bf16_fwd(self, plan, x, out, bias=None, act='lin'):
        """the forward launch as the kernel with bf16 operands in memory: with the operands made
        ahead (bf16_ahead.py) when this launch's tile is the one they were made for"""
        ctx = plan.ctx
        a = plan.bf16a.get((self, 'fwd'))
        w5 = self._w5(plan.param(self.w))
        if a is None or a['tile'] != ctx.current_tiling('igemm'):
            ctx.conv3d_fwd_bf16(x, w5, out, bias=bias, act=act, ws=plan.bf16_ws(self),
                                xkeep=plan.bf16_xkeep(self))
            return
        nx = bf16_ahead.next_image(plan, self) if bias is not None else None
        plan.ensure_wb()
        ctx.conv3d_fwd_bf16_ex(x, w5, out, bias=bias, act=act, ws=plan.bf16_ws(self),
                               xkeep=plan.bf16_xkeep(self), x_ready=plan._xb_ready.get(self, False),
                               wb=a['wb'] if plan._wb_ready else None,
                               next_xb=nx[0] if nx else None, next_kg=nx[1] if nx else 0)
        if nx:
            plan._xb_ready[nx[2]] = True

    def _plan_fwd(self, plan):
        ctx = plan.ctx
        if self._fused_head(plan) is not None:
            return                            # done by the Softmax / NLL node
        x = plan.out[self.parent]
        if self._fused_first(plan):
            nx = bf16_ahead.next_image(plan, self)
            if nx is not None:            # bf16 mode: + the next conv's channels-last input image
                ctx.conv1_pool_act_fwd_bf16(x, self._w5(plan.param(self.w)), plan.param(self.b),
                                            self._p3, self.activation_func, plan.out[self], nx[0], nx[1])
                plan._xb_ready[nx[2]] = True
            else:
                ctx.conv1_pool_act_fwd(x, self._w5(plan.param(self.w)), plan.param(self.b), self._p3,
                                       self.activation_func, plan.out[self])
            return
        if self._tail(plan) is not None:
            return                            # done by the NLL node (csrc/tail.hip)
        wp = plan.scratch[self, 'wp_f']       # packed by the plan's multi-pack launch
        plan.join_side()                      # ... which runs on the side stream
        cin = self.parent.shape['f']
        if self._fused_act(plan):
            out = plan.out[self]
            sig = (2, self.n_f, cin) + tuple(self._k3) + tuple(out.shape[2:]) + \
                (x.stride(3),)
            def fwd_act():
                if ctx.bf16_memory_form():
                    self._bf16_fwd(plan, x, out, bias=plan.param(self.b), act=self.activation_func)
                    plan.bf16_xkeep_valid(self, True)
                else:
                    plan.bf16_xkeep_valid(self, False)
                    with plan.image(wp):
                        ctx.conv3d_fwd_packed_act(x, wp, self.n_f, self._k3, plan.param(self.b),
                                                  self.activation_func, out)
            plan.tuned('igemm', sig,
                       autotune.igemm_candidates(self.n_f, cin, self._k3,
                                                 out.shape[2:], split_k=False) +
                       plan.bf16_cands(cin), fwd_act)
            return
        y = plan.scratch[self, 'y']
        sig = (0, self.n_f, cin) + tuple(self._k3) + tuple(y.shape[2:]) + \
            (x.stride(3),)
        yp = plan.scratch[self, 'y_parts']
        got = [1]                              # parts the (last) launch wrote

        def fwd_plain():
            plan.bf16_xkeep_valid(self, ctx.bf16_memory_form())
            if ctx.bf16_memory_form():
                self._bf16_fwd(plan, x, y)
                got[0] = 1
            elif yp.shape[0] > 1:
                with plan.image(wp):
                    got[0] = ctx.conv3d_fwd_packed_parts(x, wp, self.n_f, self._k3, yp)
            else:
                with plan.image(wp):
                    ctx.conv3d_fwd_packed(x, wp, self.n_f, self._k3, y)
        plan.tuned('igemm', sig,
                   autotune.igemm_candidates(self.n_f, cin, self._k3, y.shape[2:]) +
                   plan.bf16_cands(cin), fwd_plain, out=y)
        y_nparts = got[0]
        if self._bn():
            lin = y
            if any(p != 1 for p in self._p3):
                if self._mfp_pool():
                    raise NotImplementedError("MFP together with batch normalisation")
                lin = plan.scratch[self, 'lin']
                ctx.maxpool3d_fwd(y, self._p3, lin)
            train = self.batch_normalisation == 'train'
            # the running statistics are extra updates of the OPTIMISER's step function
            # (model.py:180-192), not of loss / prediction / gradient functions
            ctx.batchnorm_act_fwd(lin, plan.param(self.gamma), plan.param(self.b),
                                  plan.param(self.mean), plan.param(self.std), train,
                                  train and plan.step in ('SGD', 'Adam'), self.activation_func,
                                  plan.out[self], plan.scratch[self, 'bn_save'])
            return
        if self._mfp_pool():
            # fragment i = max-pool of the conv output shifted by the i-th offset inside the
            # pooling window (border ignored), stacked fragment-major on the batch axis
            from itertools import product
            out, n_in = plan.out[self], y.shape[0]
            pz, px, py = self._p3
            D, H, W = y.shape[2:]
            for i, (iz, ix, iy) in enumerate(product(range(pz), range(px), range(py))):
                src = y[:, :, iz:iz + D - pz + 1, ix:ix + H - px + 1, iy:iy + W - py + 1]
                ctx.pool_bias_act_fwd(src, plan.param(self.b), self._p3,
                                      self.activation_func, out[i * n_in:(i + 1) * n_in])
            return
        ndst = bf16_ahead.next_dst(plan, self)
        if ndst is not None:
            # bf16 mode: this pass also writes the next conv's channels-last input image
            ctx.pool_bias_act_fwd_bf16(yp[0], plan.param(self.b), self._p3, self.activation_func,
                                       plan.out[self], ndst, parts=y_nparts, part_stride=yp.stride(0))
            plan._xb_ready[plan.bf16a[self, 'next']['consumer']] = True
        elif y_nparts > 1:                      # split-K partial sums: added up on the way
            ctx.pool_bias_act_fwd_parts(yp, y_nparts, plan.param(self.b), self._p3,
                                        self.activation_func, plan.out[self])
        else:
            ctx.pool_bias_act_fwd(y, plan.param(self.b), self._p3, self.activation_func,
                                  plan.out[self])

    def _plan_bwd(self, plan):
        ctx = plan.ctx
        if self._fused_head(plan) is not None:
            return
        x = plan.out[self.parent]
        if self._fused_first(plan):
            if plan.opt['side_join_first']:
                # (the last launch of the backward chain: next to the side stream's weight
                # gradients it shares the CUs and takes 1.8x its time; behind them it runs alone)
                plan.join_side()
            ctx.conv1_pool_act_bwd(x, self._w5(plan.param(self.w)), plan.param(self.b), plan.grad[self],
                                   self._p3, self.activation_func, plan.pgrad(self.w),
                                   plan.pgrad(self.b), ws=plan.scratch[self, 'ws1'])
            return
        dy = plan.scratch[self, 'dy']
        tail = self._tail(plan) is not None
        if tail:
            # the tail launch wrote dy, the parent's output gradient (first and only writer)
            # and the slots the NLL node's reduction has added into the bias gradient
            if plan.needs_grad(self.parent):
                plan.grad_slot(self.parent)
                if self._tail_gm(plan) is not None:    # ... through its activation backward
                    plan.scratch[self.parent, 'dy_by_tail'] = True
        elif self._bn():
            train = self.batch_normalisation == 'train'
            pooled = any(p != 1 for p in self._p3)
            lin = plan.scratch[self, 'lin'] if pooled else plan.scratch[self, 'y']
            dlin = plan.scratch[self, 'dlin'] if pooled else dy
            ctx.batchnorm_act_bwd(plan.grad[self], lin, plan.param(self.gamma),
                                  plan.param(self.b), plan.scratch[self, 'bn_save'], train,
                                  self.activation_func, dlin,
                                  plan.pgrad(self.gamma) if self.gamma.apply_train else None,
                                  plan.pgrad(self.b))
            if pooled:
                ctx.maxpool3d_bwd(dlin, plan.scratch[self, 'y'], self._p3, dy)
        elif plan.scratch.get((self, 'dy_done')) or plan.scratch.get((self, 'dy_by_tail')):
            pass        # the consumer's data-gradient launch (below) or the tail launch
                        # (Conv._tail_gm) wrote dy and dbias
        elif (self, 'dy') in plan.bf16a:
            # bf16 mode: the pass that produces dy writes the operand images of this layer's two
            # gradient GEMMs (and the f32 tensor only if a launch still reads it)
            b = plan.bf16a[self, 'dy']
            gn = plan.scratch.get((self, 'grad_nparts'), 1)
            gp = plan.scratch.get((self, 'grad_parts'))
            dout = gp[0] if (gn > 1 and gp is not None) else plan.grad[self]
            fused = self._fused_act(plan)
            ctx.pool_bias_act_bwd_bf16(dout, plan.out[self] if fused else plan.scratch[self, 'y'],
                                       None if fused else plan.param(self.b), self._p3,
                                       self.activation_func, dy if b['want_f32'] else None,
                                       plan.pgrad(self.b), b['dst'], parts=gn,
                                       part_stride=gp.stride(0) if (gn > 1 and gp is not None) else 0)
            plan._dy_ready[self] = True
        elif self._fused_act(plan):
            gn = plan.scratch.get((self, 'grad_nparts'), 1)
            if gn > 1:
                ctx.bias_act_bwd_out_parts(plan.scratch[self, 'grad_parts'], gn, plan.out[self],
                                           self.activation_func, dy, plan.pgrad(self.b))
            else:
                ctx.bias_act_bwd_out(plan.grad[self], plan.out[self], self.activation_func, dy,
                                     plan.pgrad(self.b))
        else:
            gn = plan.scratch.get((self, 'grad_nparts'), 1)
            if gn > 1:
                ctx.pool_bias_act_bwd_parts(plan.scratch[self, 'grad_parts'], gn,
                                            plan.scratch[self, 'y'], plan.param(self.b),
                                            self._p3, self.activation_func, dy,
                                            plan.pgrad(self.b))
            else:
                ctx.pool_bias_act_bwd(plan.grad[self], plan.scratch[self, 'y'],
                                      plan.param(self.b), self._p3, self.activation_func, dy,
                                      plan.pgrad(self.b))
        cin = self.parent.shape['f']
        dw = self._w5(plan.pgrad(self.w))
        dyp = plan.scratch[self, 'dy_pad']
        sigw = (self.n_f, cin) + tuple(self._k3) + tuple(dy.shape[2:]) + \
            (x.stride(3), dy.stride(3))
        wcands = plan.bf16_wgrad_cands(cin, self._k3)
        wws = plan.bf16_wgrad_ws(self) if wcands else None

        def wgrad(accumulate):
            if ctx.bf16_memory_wgrad():     # "32,MB,NB,0,S": bf16 operands in memory
                # (the forward's channels-last bf16 copy of x, when this step's forward of
                # the layer was the memory form: one conversion pass less)
                aw = plan.bf16a.get((self, 'wgrad'))
                if aw is not None and plan._dy_ready.get(self) and aw['tile'] == ctx.current_tiling('wgrad'):
                    ctx.conv3d_wgrad_bf16_ex(x, dy, dw, accumulate=accumulate, ws=wws,
                                             xcl=plan.bf16_xkeep(self, only_valid=True),
                                             dyc=aw['dyc'], sums=aw['sums'])
                else:
                    self._need_f32_dy(plan)
                    ctx.conv3d_wgrad_bf16(x, dy, dw, accumulate=accumulate, ws=wws,
                                          xcl=plan.bf16_xkeep(self, only_valid=True))
            else:
                self._need_f32_dy(plan)
                ctx.conv3d_wgrad_pad(x, dyp, dw, accumulate=accumulate)
        # the weight gradient is independent of the data-gradient chain below: side stream.
        # (x is a tensor of the plan, or a view into one: Plan.SLACK zeroed floats behind it --
        # checked against the allocation, not assumed: Plan.slack_behind)
        slack = min(4 * plan.SLACK, plan.slack_behind(x))

        def wgrad_launch():
            ctx.set_input_slack(slack)
            try:
                plan.tuned('wgrad', sigw,
                           autotune.wgrad_candidates(self.n_f, cin, self._k3, dy.shape[2:]) + wcands,
                           lambda: wgrad(True), fn_tune=lambda: wgrad(False))
            finally:
                ctx.set_input_slack(0)
        plan.on_side(wgrad_launch, defer=True, force=plan.side_forced(self))
        if plan.needs_grad(self.parent) and not tail:
            wp = plan.scratch[self, 'wp_d']
            dyp = plan.scratch[self, 'dy_pad']
            par = self.parent
            if self._actbwd_into_parent(plan):
                # the parent conv does not pool and feeds only this conv: its activation
                # backward runs in this launch's epilogue, which writes the parent's
                # zero-padded gradient buffer and bias gradient directly
                pdyp = plan.scratch[par, 'dy_pad']
                ppad = [kk - 1 for kk in par._k3]
                if par._fused_act(plan):
                    src, pb = plan.out[par], None
                else:
                    src, pb = plan.scratch[par, 'y'], plan.param(par.b)
                osp = plan.out_shape(par)[2:]
                sig = (1, cin, self.n_f) + tuple(self._k3) + tuple(osp) + \
                    (dyp.stride(3),)
                plan._grad_written.add(id(par))
                plan.scratch[par, 'dy_done'] = True

                def launch(dbias):
                    with plan.image(wp):
                        ctx.conv3d_dgrad_packed_actbwd(dyp, wp, cin, self._k3, src,
                                                       par.activation_func, pdyp, ppad, dbias,
                                                       bias_prev=pb)
                plan.tuned('igemm', sig,
                           autotune.igemm_candidates(cin, self.n_f, self._k3, osp),
                           lambda: launch(plan.pgrad(par.b)), fn_tune=lambda: launch(None),
                           fn_once=lambda: launch(plan.pgrad(par.b)), out=pdyp)
                return
            dst, first = plan.grad_slot(self.parent)
            out = dst if first else plan.tmp_like(dst)
            # first writer of the parent's output gradient: split-K partial sums go to the
            # parent's slabs, its activation backward adds them up
            gparts = plan.scratch.get((self.parent, 'grad_parts')) if first else None
            got = [1]
            sig = (1, cin, self.n_f) + tuple(self._k3) + tuple(out.shape[2:]) + \
                (dyp.stride(3),)
            def dgrad():
                if ctx.bf16_memory_form():
                    ad = plan.bf16a.get((self, 'dgrad'))
                    if ad is not None and plan._dy_ready.get(self) and ad['tile'] == ctx.current_tiling('igemm'):
                        plan.ensure_wb()
                        ctx.conv3d_dgrad_bf16_ex(dyp, self._w5(plan.param(self.w)), out,
                                                 ws=plan.bf16_ws(self), dy_cl=ad['cl'],
                                                 wb=ad['wb'] if plan._wb_ready else None)
                    else:
                        self._need_f32_dy(plan)
                        ctx.conv3d_dgrad_bf16(dyp, self._w5(plan.param(self.w)), out,
                                              ws=plan.bf16_ws(self))
                    got[0] = 1
                elif gparts is not None:
                    self._need_f32_dy(plan)
                    with plan.image(wp):
                        got[0] = ctx.conv3d_dgrad_packed_parts(dyp, wp, cin, self._k3, gparts)
                else:
                    self._need_f32_dy(plan)
                    with plan.image(wp):
                        ctx.conv3d_dgrad_packed(dyp, wp, cin, self._k3, out)
            plan.tuned('igemm', sig,
                       autotune.igemm_candidates(cin, self.n_f, self._k3, out.shape[2:]) +
                       plan.bf16_cands(self.n_f), dgrad, out=out)
            if first:
                plan.scratch[self.parent, 'grad_nparts'] = got[0]
            if not first:
                ctx.copy5(out, dst, accumulate=True)

    def _actbwd_into_parent(self, plan):
        """this conv's data gradient can carry the activation backward of its parent: the
        parent is a plain Conv without pooling (relu / lin) whose output gradient comes
        from this node alone"""
        par = self.parent
        if not (plan.fuse_actbwd and type(par) is Conv and type(self) is Conv
                and not par._bn() and all(p == 1 for p in par._p3)
                and par.activation_func in ('relu', 'lin')
                and (par, 'dy_pad') in plan.scratch
                and not par._fused_first(plan) and par._fused_head(plan) is None
                and self._k3[2] in (1, 3, 4, 5)
                and plan.out_shape(par)[4] >= 4):
            return False
        users = [c for c in par.children.values()
                 if id(c) in plan._loss_anc or c is plan.loss_node]
        return len(users) == 1 and users[0] is self


class Perceptron(NeuralLayer):
    """Perceptron layer (neural.py:258-410): ``act((gamma / std) * dot(x, w) + b - ...)``;
    ``w`` has the reference's (n_in, n_f) shape; ``flatten=True`` joins every non-batch axis
    of the parent (C order).  Device side: csrc/dense_bn.hip."""

    def __init__(self, parent, n_f, activation_func='relu',
                 flatten=False, batch_normalisation=False, dropout_rate=0,
                 name="dot", print_repr=True, w=None, b=None, gamma=None,
                 mean=None, std=None, gradnet_mode=None):
        Node.__init__(self, parent, name, print_repr)
        self.n_f = n_f
        self.activation_func = activation_func
        self.batch_normalisation = batch_normalisation
        self.gradnet_mode = gradnet_mode
        self.axis = parent.shape.tag2index('f')
        self.flatten = flatten
        self.spatial_axes = parent.shape.spatial_axes
        if activation_func not in _HIP_ACTS:
            raise NotImplementedError("activation_func=%r: only %s are on the HIP hot path"
                                      % (activation_func, _HIP_ACTS))
        if flatten:
            if self.axis != 1:
                raise NotImplementedError("Cannot flatten tensor for "
                                          "Perceptron layer when batchsize is "
                                          "not on first axis")
            n_in = parent.shape.stripbatch_prod
        else:
            n_in = parent.shape['f']
            if any(int(s) != 1 for s in parent.shape.spatial_shape):
                raise NotImplementedError("a Perceptron over the feature axis of a spatial "
                                          "tensor is a (1,1,1) Conv on this path")
        w_sh = (n_in, n_f)
        self.w_sh = w_sh
        self._setup_params(w_sh, w, b, gamma, mean, std, dropout_rate)

    def _make_output(self):
        self.output = Sym(self, floatX)

    def _calc_shape(self):
        sh = self.parent.shape
        if self.flatten:
            self.shape = TaggedShape((sh['b'], self.n_f), 'b,f')
        else:
            self.shape = sh.updateshape('f', self.n_f)

    def _calc_comp_cost(self):
        self.computational_cost = self.parent.shape.stripnone_prod * self.n_f

    def _bn(self):
        return self.batch_normalisation in ('train', 'predict')

    def _plan_alloc(self, plan):
        plan.alloc_out(self)
        plan.scratch[self, 'lin'] = plan.empty(plan.out_shape(self))
        if plan.training:
            plan.scratch[self, 'dlin'] = plan.empty(plan.out_shape(self))
        if self._bn():
            plan.scratch[self, 'bn_save'] = plan.zeros_flat(2 * self.n_f)

    def _x2(self, plan, t):
        """the parent's buffer as the (batch, n_in) matrix"""
        if not t.is_contiguous():
            raise NotImplementedError("Perceptron input must be a dense buffer")
        return t.reshape(t.shape[0], -1)

    def _plan_fwd(self, plan):
        ctx = plan.ctx
        lin = plan.scratch[self, 'lin']
        ctx.dense_fwd(self._x2(plan, plan.out[self.parent]), plan.param(self.w),
                      lin.reshape(lin.shape[0], -1))
        if self._bn():
            train = self.batch_normalisation == 'train'
            ctx.batchnorm_act_fwd(lin, plan.param(self.gamma), plan.param(self.b),
                                  plan.param(self.mean), plan.param(self.std), train,
                                  train and plan.step in ('SGD', 'Adam'), self.activation_func,
                                  plan.out[self], plan.scratch[self, 'bn_save'])
        else:
            ctx.pool_bias_act_fwd(lin, plan.param(self.b), (1, 1, 1), self.activation_func,
                                  plan.out[self])

    def _plan_bwd(self, plan):
        ctx = plan.ctx
        lin, dlin = plan.scratch[self, 'lin'], plan.scratch[self, 'dlin']
        if self._bn():
            ctx.batchnorm_act_bwd(plan.grad[self], lin, plan.param(self.gamma),
                                  plan.param(self.b), plan.scratch[self, 'bn_save'],
                                  self.batch_normalisation == 'train', self.activation_func,
                                  dlin, plan.pgrad(self.gamma) if self.gamma.apply_train else None,
                                  plan.pgrad(self.b))
        else:
            ctx.pool_bias_act_bwd(plan.grad[self], lin, plan.param(self.b), (1, 1, 1),
                                  self.activation_func, dlin, plan.pgrad(self.b))
        d2 = dlin.reshape(dlin.shape[0], -1)
        ctx.dense_wgrad(self._x2(plan, plan.out[self.parent]), d2, plan.pgrad(self.w),
                        accumulate=True)
        if plan.needs_grad(self.parent):
            dst, first = plan.grad_slot(self.parent)
            ctx.dense_dgrad(d2, plan.param(self.w), self._x2(plan, dst), accumulate=not first)


class FragmentsToDense(Node):
    """neural.py:862-903: interleave the MFP fragments on the batch axis into ONE dense
    prediction: ``dense[..., off_i[k]::strides[k]] = fragment_i``."""

    def __init__(self, parent, name="to_dense", print_repr=True):
        super(FragmentsToDense, self).__init__(parent, name, print_repr)

    def _make_output(self):
        sh = self.parent.shape
        if sh['b'] != len(sh.mfp_offsets) or sh['b'] != np.prod(sh.strides):
            raise ValueError("Need %i fragments on the batch axis. "
                             "Is MFP active at all?" % np.prod(sh.strides))
        self.output = Sym(self, floatX)

    def _calc_shape(self):
        sh = self.parent.shape
        for ax, st in zip(sh.spatial_axes, sh.strides):
            sh = sh.updateshape(ax, int(st), mode='mult')
        sh = sh.updateshape('b', 1)
        n_sp = len(sh.spatial_axes)
        self.shape = TaggedShape(sh.shape, sh.tags, np.ones(n_sp, np.int64),
                                 np.zeros((1, n_sp), np.int64), sh.fov)

    def _calc_comp_cost(self):
        self.computational_cost = 0

    def _plan_alloc(self, plan):
        if plan.training:
            raise NotImplementedError("FragmentsToDense is a prediction-time node")
        plan.alloc_out(self)

    def _plan_fwd(self, plan):
        frag, dense = plan.out[self.parent], plan.out[self]
        sz, sx, sy = (int(v) for v in self.parent.shape.strides)
        for i, off in enumerate(np.asarray(self.parent.shape.mfp_offsets)):
            # strided scatter on the plan's stream (torch; captured with the graph)
            dense[0, :, int(off[0])::sz, int(off[1])::sx, int(off[2])::sy] = frag[i]

    def _plan_bwd(self, plan):
        raise NotImplementedError("FragmentsToDense is a prediction-time node")


class UpConv(Conv):
    """Upconvolution / transposed convolution with stride = kernel = pool_shape
    (neural.py:907-1129).  F2: y[n,co,p*i+r] = sum_ci w[co,ci,r] x[n,ci,i]."""

    def __init__(self, parent, n_f, pool_shape, activation_func='relu',
                 identity_init=True, batch_normalisation=False, dropout_rate=0,
                 name="upconv", print_repr=True, w=None, b=None, gamma=None,
                 mean=None, std=None, gradnet_mode=None):
        pool_shape = tuple(int(p) for p in pool_shape)
        Conv.__init__(self, parent, n_f, pool_shape, pool_shape, 'valid', activation_func,
                      mfp=False, batch_normalisation=batch_normalisation,
                      dropout_rate=dropout_rate, name=name, print_repr=print_repr,
                      w=w, b=b, gamma=gamma, mean=mean, std=std, gradnet_mode=gradnet_mode)
        if identity_init:          # neural.py:977-986
            w_val = self.w.get_value() * 0.1
            s = np.arange(np.minimum(w_val.shape[0], w_val.shape[1]))
            w_val[s, s] = 1.0
            self.w.set_value(w_val)
            self.b.set_value(self.b.get_value() * 0.0)

    def _make_output(self):
        self.output = Sym(self, floatX)

    def _calc_shape(self):
        """neural.py:1074-1097."""
        self.strides = np.divide(self.strides, self.pool_shape)
        sh = self.parent.shape
        for j, (i, f, p) in enumerate(zip(self.spatial_axes, self.filter_shape,
                                          self.pool_shape)):
            s = (sh[i] * p) + p - 1 + (1 - f)
            sh = sh.updateshape(i, s)
            sh = sh.updatefov(j, -1)
        sh = sh.updateshape('f', self.n_f)
        sh = sh.updatestrides(self.strides)
        self.shape = sh

    def _calc_comp_cost(self):
        sh = self.parent.shape
        n_position = 1
        for i, f, p in zip(self.spatial_axes, self.filter_shape, self.pool_shape):
            n_position *= (sh[i] * p) + 1 - f
        b = 1 if sh['b'] is None else sh['b']
        self.computational_cost = int(np.prod(self.w_sh)) * n_position * b

    def make_dual(self, *args, **kwargs):
        raise NotImplementedError("Use Conv instead?")

    def _plan_alloc(self, plan):
        plan.alloc_out(self)
        xs = plan.out_shape(self.parent)
        cin = self.parent.shape['f']
        nb = plan.ctx.upconv_ws_bytes(self.n_f, cin, self.pool_shape, xs)
        plan.scratch[self, 'ws'] = plan.empty_flat(nb // 4 + 64)
        # the two packed images, refreshed by the plan's one repack launch per step (they
        # used to be repacked inside every forward and backward call: two launches per node)
        w5 = plan._w5(plan.param(self.w))
        if w5.dim() == 5 and tuple(w5.shape[2:]) == tuple(self.pool_shape) \
                and plan.opt['upconv_packed']:
            ib = plan.ctx.upconv_image_bytes(self.n_f, cin, self.pool_shape)
            plan.scratch[self, 'wp_f'] = plan.zeros_flat(ib // 4 + 64)
            plan.pack_jobs.append((self.w, plan.scratch[self, 'wp_f'], 2))
            if plan.training and plan.needs_grad(self.parent):
                plan.scratch[self, 'wp_d'] = plan.zeros_flat(ib // 4 + 64)
                plan.pack_jobs.append((self.w, plan.scratch[self, 'wp_d'], 3))

    def _tune_sigs(self, plan):
        """the GEMMs inside e2_upconv3d_fwd / _bwd: forward M = Cout*R, K = Cin over the INPUT
        positions; data gradient the transpose; weight gradient over the dense space-to-depth
        image (1x1x1 taps: it is its own padded form, so the direct kernel applies)"""
        x = plan.out[self.parent]
        cin, R = self.parent.shape['f'], int(np.prod(self.pool_shape))
        sp = tuple(x.shape[2:])
        key = (self.n_f, cin) + tuple(self.pool_shape) + sp + (x.stride(3),)
        cw = autotune.wgrad_candidates(self.n_f * R, cin, (1, 1, 1), sp)
        return dict(fwd=((3,) + key, autotune.igemm_candidates(self.n_f * R, cin, (1, 1, 1), sp)),
                    dgrad=((4,) + key, autotune.igemm_candidates(cin, self.n_f * R, (1, 1, 1), sp)),
                    wgrad=((5,) + key, cw))

    def _plan_fwd(self, plan):
        sig, cands = self._tune_sigs(plan)['fwd']
        wp = plan.scratch.get((self, 'wp_f'))
        if wp is not None:
            plan.join_side()                  # (the repack may run on the side stream)

        def run():
            if wp is not None:
                plan.ctx.upconv3d_fwd_packed(plan.out[self.parent], wp, plan.param(self.b),
                                             self.n_f, self.pool_shape, self.activation_func,
                                             plan.out[self])
            else:
                plan.ctx.upconv3d_fwd(plan.out[self.parent], plan.param(self.w),
                                      plan.param(self.b), self.pool_shape,
                                      self.activation_func, plan.out[self],
                                      plan.scratch[self, 'ws'])
        plan.tuned('igemm', sig, cands, run)

    def _plan_bwd(self, plan):
        ctx = plan.ctx
        dx = None
        first = True
        if plan.needs_grad(self.parent):
            dst, first = plan.grad_slot(self.parent)
            dx = dst if first else plan.tmp_like(dst)
        # one call = space-to-depth + data-gradient GEMM + weight-gradient GEMM.  The eager
        # form overwrites every output (idempotent: the tuner may run it many times, each
        # GEMM's tiling tuned with the other at its best known one); the captured form adds
        # dw and dbias to the gradient arena the plan has zeroed (two fill launches less)
        sigs = self._tune_sigs(plan)
        wp_d = plan.scratch.get((self, 'wp_d'))
        packed = (self, 'wp_f') in plan.scratch and (dx is None or wp_d is not None)

        def run():
            if packed:
                ctx.upconv3d_bwd_packed(plan.out[self.parent], wp_d, plan.out[self],
                                        plan.grad[self], self.pool_shape, self.activation_func,
                                        dx, plan.pgrad(self.w), plan.pgrad(self.b),
                                        plan.scratch[self, 'ws'], accumulate=plan._capturing)
            else:
                ctx.upconv3d_bwd(plan.out[self.parent], plan.param(self.w), plan.out[self],
                                 plan.grad[self], self.pool_shape, self.activation_func, dx,
                                 plan.pgrad(self.w), plan.pgrad(self.b), plan.scratch[self, 'ws'])
        plan.tuned('igemm', sigs['dgrad'][0], sigs['dgrad'][1] if dx is not None else [],
                   lambda: plan.tuned('wgrad', sigs['wgrad'][0], sigs['wgrad'][1], run))
        if dx is not None and not first:
            ctx.copy5(dx, dst, accumulate=True)


class Crop(Node):
    """Symmetric spatial crop (neural.py:1132-1190): a zero-copy view."""

    def __init__(self, parent, crop, name="crop", print_repr=True):
        super(Crop, self).__init__(parent, name, print_repr)
        self.crop = [int(c) for c in crop]

    def _calc_shape(self):
        sh = self.parent.shape.copy()
        k = 0
        for i, s in enumerate(self.parent.shape):
            if i in self.parent.shape.spatial_axes:
                sh = sh.updateshape(i, s - 2 * self.crop[k])
                k += 1
        self.shape = sh

    def _calc_comp_cost(self):
        self.computational_cost = 0

    def _slicer(self):
        sl = []
        k = 0
        for i, s in enumerate(self.parent.shape):
            if i in self.parent.shape.spatial_axes:
                off = self.crop[k]
                sl.append(slice(off, s - off))
                k += 1
            else:
                sl.append(slice(None))
        return tuple(sl)

    def _plan_alloc(self, plan):
        plan.out[self] = plan.out[self.parent][self._slicer()]
        if plan.training and plan.needs_grad(self):
            plan.alloc_grad(self)

    def _plan_fwd(self, plan):
        pass

    def _plan_bwd(self, plan):
        plan.add_grad_region(self.parent, self._slicer(), plan.grad[self])


def AutoMerge(parent1, parent2, upconv_n_f=None, merge_mode='concat',
              disable_upconv=False, upconv_kwargs=None, name='merge', print_repr=True):
    """neural.py:1282-1405: align a low-res and a high-res branch by UpConv +
    Crop, then Concat((lo_res, hi_res)) or Add."""
    assert len(parent1.shape) == len(parent2.shape)
    assert parent1.shape.spatial_axes == parent2.shape.spatial_axes
    if any(np.array(parent2.shape.strides) // np.array(parent1.shape.strides) < 1):
        lo_res, hi_res = parent1, parent2
    else:
        hi_res, lo_res = parent1, parent2
    unpool = (np.array(lo_res.shape.strides) // np.array(hi_res.shape.strides)).astype(int)
    if np.any(unpool > 1) and not disable_upconv:
        if upconv_n_f is None:
            raise ValueError('AutoMerge is trying to insert an UpConv node, but '
                             'upconv_n_f is not defined. Please set it to the '
                             'desired number of features to be used for UpConv.')
        if upconv_kwargs is None:
            upconv_kwargs = {}
        lo_res = UpConv(lo_res, upconv_n_f, tuple(int(u) for u in unpool), **upconv_kwargs)
    sh_hi = hi_res.shape.spatial_shape
    sh_lo = lo_res.shape.spatial_shape
    crop_lo, crop_hi = [], []
    for i in range(len(sh_hi)):
        diff = sh_hi[i] - sh_lo[i]
        if diff % 2 != 0:
            raise ValueError("hi_res and lo_res maps cannot be aligned with "
                             "shapes:\n%s\n%s" % (sh_hi, sh_lo))
        if diff > 0:
            crop_hi.append(diff // 2)
            crop_lo.append(0)
        else:
            crop_lo.append(-diff // 2)
            crop_hi.append(0)
    if np.any(crop_lo):
        lo_res = Crop(lo_res, crop_lo, print_repr=print_repr)
    if np.any(crop_hi):
        hi_res = Crop(hi_res, crop_hi, print_repr=print_repr)
    if merge_mode == 'concat':
        return Concat((lo_res, hi_res), axis='f', name=name, print_repr=print_repr)
    elif merge_mode == 'add':
        return Add(lo_res, hi_res, name=name, print_repr=print_repr)
    raise ValueError('Invalid "merge_mode". Should be "add" or "concat".')


UpConvMerge = AutoMerge


class Pool(Node):
    """Max-pooling node (neural.py:1409-1559)."""

    def __init__(self, parent, pool_shape, stride=None, mfp=False, mode='max',
                 name="pool", print_repr=True):
        super(Pool, self).__init__(parent, name, print_repr)
        if mfp:
            # the reference refuses it as well (neural.py:1535-1536, in _calc_shape)
            raise NotImplementedError("Check this first before use")
        if stride is not None and tuple(stride) != tuple(pool_shape):
            raise NotImplementedError("Stride!=Pool using 3d pooling")   # computations.py:612
        if mode != 'max':
            raise NotImplementedError("Pooling mode %r needs cuDNN in the reference; "
                                      "only 'max' is on the hot path" % (mode,))
        self.pool_shape = tuple(int(p) for p in pool_shape)
        self.pool_stride = self.pool_shape
        self.mfp = False
        self.mode = mode
        self.strides = parent.shape.strides
        self.axis = parent.shape.tag2index('f')
        spatial_axes = self.parent.shape.spatial_axes
        if len(pool_shape) != 3 or len(self.parent.shape) != 5 or spatial_axes != [2, 3, 4]:
            raise NotImplementedError("Cannot pool non-standard shapes / axis orders "
                                      "on the HIP hot path.")
        self.spatial_axes = spatial_axes
        self.conv_dim = 3

    def _make_output(self):
        self.output = Sym(self, floatX)
        self.strides = np.multiply(self.pool_stride, self.strides)

    def _calc_shape(self):
        sh = self.parent.shape
        for j, (i, p, st) in enumerate(zip(self.spatial_axes, self.pool_shape,
                                           self.pool_stride)):
            tmp = sh[i] - p + st - 1
            s = tmp // st + 1
            if (tmp + 1) % st != 0:
                raise ValueError("Cannot downsample spatial axis '%s' of length %i "
                                 "by factor %i with pool %i." % (sh.tags[i], sh[i], st, p))
            sh = sh.updateshape(i, s)
            fov = sh.fov[j] + (p - 1) * sh.strides[j] if sh.fov[j] > 0 else -1
            sh = sh.updatefov(j, fov)
        self.shape = sh.updatestrides(self.strides)

    def _plan_fwd(self, plan):
        plan.ctx.maxpool3d_fwd(plan.out[self.parent], self.pool_shape, plan.out[self])

    def _plan_bwd(self, plan):
        if not plan.needs_grad(self.parent):
            return
        dst, first = plan.grad_slot(self.parent)
        plan.ctx.maxpool3d_bwd(plan.grad[self], plan.out[self.parent], self.pool_shape, dst,
                               accumulate=not first)
