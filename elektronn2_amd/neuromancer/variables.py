"""Parameter objects with the reference's protocol
(elektronn2/neuromancer/variables.py:25-155): ``get_value()`` (copy),
``set_value(ndarray)`` with shape check and optional float downcast, ``name``,
``apply_train``, ``apply_reg``, ``constant``, ``updates``; and ``initweights``
(variables.py:205-266).

Storage: a parameter starts as a host numpy array.  When a plan is built the
model binds it to a slice of the flat device arena (``bind``); from then on the
device copy is authoritative and ``get_value`` reads it back.
"""
from __future__ import annotations

import numpy as np

from .graphutils import floatX, as_floatX

__all__ = ['VariableParam', 'VariableWeight', 'ConstantParam', 'initweights']


class VariableParam(object):
    def __init__(self, value=None, name=None, apply_train=True, apply_reg=True,
                 dtype=None, **unused):
        self.apply_reg = apply_reg
        self.apply_train = apply_train
        self._updates = None
        self.constant = False
        if not apply_train:
            name = (name or '') + "_noTrain"
        self.name = name
        if isinstance(value, (int, float)):
            value = np.array(value, dtype=dtype)
        value = np.array(value, copy=True)
        if dtype is not None:
            value = value.astype(dtype)
        self._host = value
        self._dev = None          # torch view into the model's arena
        self._owner = None        # the model that owns the arena

    # -- device binding -----------------------------------------------------
    def bind(self, dev_view):
        import torch
        dev_view.copy_(torch.from_numpy(
            np.ascontiguousarray(self._host, np.float32)).reshape(dev_view.shape))
        self._dev = dev_view

    @property
    def dtype(self):
        return self._host.dtype

    @property
    def shape(self):
        return self._host.shape

    def get_value(self, borrow=False):
        if self._dev is not None:
            self._host = self._dev.detach().cpu().numpy().reshape(
                self._host.shape).astype(self._host.dtype)
        return self._host.copy()

    def set_value(self, new_value, borrow=False):
        new_value = np.asarray(new_value)
        if new_value.dtype != self._host.dtype:
            new_value = new_value.astype(self._host.dtype)   # allow_floatX_downcast
        if new_value.shape != self._host.shape:
            new_value = np.broadcast_to(new_value, self._host.shape)
        self._host = np.array(new_value, copy=True)
        if self._dev is not None:
            import torch
            self._dev.copy_(torch.from_numpy(
                np.ascontiguousarray(self._host, np.float32)).reshape(self._dev.shape))
            if self._owner is not None:
                # (a training plan whose optimiser launch keeps the packed weight images current
                # must repack before its next step: Plan._plan_fused_update)
                self._owner._img_owner = None

    @property
    def updates(self):
        return self._updates

    @updates.setter
    def updates(self, up):
        if self.apply_train or self.apply_reg:
            raise ValueError("Cannot register extra updates for trainable "
                             "parameter %s" % (repr(self),))
        self._updates = up

    def __repr__(self):
        return "<%s %s %s>" % (self.__class__.__name__, self.name, self._host.shape)


class VariableWeight(VariableParam):
    def __init__(self, shape=None, init_kwargs=None, value=None, name=None,
                 apply_train=True, apply_reg=True, dtype=None, **unused):
        if value is None:
            if (shape is None) or (init_kwargs is None):
                raise ValueError("shape and init_kwargs are required if value is None")
            value = initweights(shape, **init_kwargs)
        elif shape is not None:
            if np.array(value).ndim > 1:
                raise ValueError("If value and shape are specified, value must be scalar.")
            value = np.ones(shape) * value
        super(VariableWeight, self).__init__(value, name, apply_train, apply_reg, dtype)

    def set_value(self, new_value, borrow=False):
        sh = self._host.shape
        if isinstance(new_value, np.ndarray):
            if not (sh == new_value.shape):
                raise NotImplementedError(
                    "given shape: %s, required shape: %s Crop value or extend "
                    "with similar numbers" % (new_value.shape, sh))
        elif isinstance(new_value, (float, int)):
            pass
        else:
            raise ValueError("Value/type not understood")
        super(VariableWeight, self).set_value(as_floatX(new_value), borrow)


class ConstantParam(object):
    def __init__(self, value, name=None, dtype=None, **unused):
        self.name = (name or '') + "_const"
        if isinstance(value, (int, float)):
            value = np.array(value, dtype=dtype)
        if dtype is not None:
            value = value.astype(dtype)
        self.value = value
        self.apply_train = False
        self.apply_reg = False
        self.constant = True
        self._dev = None

    def set_value(self, new_value, borrow=False):
        raise RuntimeError("Cannot set value for ConstantParam")

    def get_value(self, borrow=False):
        return self.value

    @property
    def updates(self):
        return None


def initweights(shape, dtype=floatX, scale='glorot', mode='normal', pool=None,
                spatial_axes=None):
    """Initial values of a weight / bias tensor, drawn from the global ``np.random`` in the
    reference's call order (variables.py:205-266; F8: nothing is seeded -- tests seed
    ``np.random`` or inject ``w=`` / ``b=``).  The three forms the BASELINE configs reach:

    * ``mode='const'``: every element = ``scale`` (relu biases: 1 / prod(kernel));
    * ``mode='fix-uni'``: U(-scale, scale) (biases of 'lin' layers: scale 1e-6);
    * ``scale='glorot', mode='normal'``: N(0, sqrt(2 / fan)) with fan = n_in + n_out for a
      2-D matrix and (n_in + n_out / prod(pool)) * prod(kernel) for a conv tensor whose
      non-spatial axes are (n_out, n_in).

    * ``scale='glorot', mode='uni'``: U(-sqrt(2 / fan), +sqrt(2 / fan));
    * ``scale='glorot', mode='ortho'`` (``config.use_ortho_init``): rows of the (n_out, rest)
      matrix = right singular vectors of a normal draw, each scaled to the glorot std.

    'prelu' belongs to an activation the hot path rejects and is not built."""
    if mode == 'const':
        return np.ascontiguousarray(np.full(shape, scale), dtype=dtype)
    if mode == 'fix-uni':
        return np.ascontiguousarray(np.random.uniform(-scale, scale, shape), dtype=dtype)
    if scale != 'glorot' or mode not in ('normal', 'uni', 'ortho'):
        raise NotImplementedError("initweights(scale=%r, mode=%r): only 'const', 'fix-uni' and "
                                  "glorot with 'normal' / 'uni' / 'ortho' are built "
                                  "(SURVEY.md 8a-9)" % (scale, mode))
    if len(shape) == 2:
        n_out, n_in = shape[1], shape[0]
        fan = shape[0] + shape[1]
    else:
        if spatial_axes is None:
            raise ValueError("initweights: spatial_axes are needed for a conv tensor")
        kernel = [s for i, s in enumerate(shape) if i in spatial_axes]
        n_out, n_in = [s for i, s in enumerate(shape) if i not in spatial_axes]
        fan = (n_in + float(n_out) / np.prod(pool)) * np.prod(kernel)
    std = np.sqrt(2.0 / fan)
    if mode == 'uni':
        return np.ascontiguousarray(np.random.uniform(-std, std, shape), dtype=dtype)
    if mode == 'ortho':
        return np.ascontiguousarray(_ortho_rows(shape, n_out, std), dtype=dtype)
    return np.ascontiguousarray(np.random.normal(0, std, shape), dtype=dtype)


def _ortho_rows(shape, n_out, std):
    """variables.py:246-262: the first draw always has the tensor's shape (it fixes the RNG
    stream); with more rows than columns a square matrix is drawn instead and its surplus
    columns are dropped after the decomposition."""
    m = np.random.normal(0, std, size=shape).reshape(n_out, -1)
    cols = m.shape[1]
    if n_out > cols:
        m = np.random.normal(0, std, size=(n_out, n_out))
    vt = np.linalg.svd(m, full_matrices=False)[2]
    rows = vt * (std / vt.std(axis=1, keepdims=True))
    return rows[:, :cols].reshape(shape)
