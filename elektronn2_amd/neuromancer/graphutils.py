"""Shapes and the compile seam.

``TaggedShape`` restates the bookkeeping of the reference
(elektronn2/neuromancer/graphutils.py:27-310): shape + tags + strides + fov +
mfp_offsets, which ``BatchCreatorImage`` reads from the input/target nodes
(data/cnndata.py:134-140).

``make_func`` is the seam the reference wraps around ``theano.function``
(graphutils.py:314-387).  Here "compile" builds a static launch plan over the
C ABI of libe2hip.so (see ``plan.py``); the call protocol is kept: lazy compile
on first call, zero-arg call = compile only, single value or list returned,
``last_exec_time`` in seconds, ``func = None`` forces a rebuild.
"""
from __future__ import annotations

import time

import numpy as np

floatX = 'float32'


def as_floatX(x):
    if not hasattr(x, '__len__'):
        return np.array(x, dtype=floatX)
    return np.ascontiguousarray(x, dtype=floatX)


class TaggedShape(object):
    """Shape with per-axis tags, spatial strides, fov and MFP offsets."""

    def __init__(self, shape, tags, strides=None, mfp_offsets=None, fov=None):
        self._shape = list(shape)
        self._tags = self._check_tags(tags)
        if len(self._shape) != len(self._tags):
            raise ValueError("Shape %s and tags %s must have same length"
                             % (self._shape, self._tags))
        n_sp = len(self.spatial_axes)
        self._strides = (np.ones(n_sp, np.int64) if strides is None
                         else np.array(strides))
        self._mfp_offsets = (np.zeros((1, n_sp), np.int64) if mfp_offsets is None
                             else np.atleast_2d(np.array(mfp_offsets, np.int64)))
        self._fov = (np.ones(n_sp, np.int64) if fov is None
                     else np.array(fov, np.int64))

    def __repr__(self):
        return "[" + ", ".join("(%s,%s)" % (s, t)
                               for s, t in zip(self._shape, self._tags)) + "]"

    @property
    def ext_repr(self):
        return repr(self) + '\nfov=%s, offsets=%s, strides=%s, spatial shape=%s' % (
            self.fov, self.offsets, self.strides, self.spatial_shape)

    @staticmethod
    def _check_tags(tags):
        if tags is None:
            return None
        if not isinstance(tags, (list, tuple)):
            if isinstance(tags, str):
                tags = [x.strip() for x in tags.split(',')]
            else:
                raise ValueError("Tags must be either list/tuple of "
                                 "comma-separated string, not %s" % (tags,))
        for t in tags:
            if t not in ['r', 'b', 'z', 'f', 'x', 'y', 's']:
                raise ValueError("Unknown tag %s" % (t,))
        return list(tags)

    def __getitem__(self, sl):
        if isinstance(sl, str):
            return self._shape[self.tag2index(sl)]
        return self._shape[sl]

    def __len__(self):
        return len(self._shape)

    def __iter__(self):
        return iter(self._shape)

    shape = property(lambda self: self._shape)
    tags = property(lambda self: self._tags)
    strides = property(lambda self: self._strides)
    mfp_offsets = property(lambda self: self._mfp_offsets)

    @property
    def fov(self):
        return [int(f) for f in self._fov]

    @property
    def fov_all_centered(self):
        return bool(np.all(np.mod(self.fov, 2) == 1))

    @property
    def offsets(self):
        return [int(i) // 2 for i in self.fov]

    @property
    def spatial_axes(self):
        ax = [self.tag2index(t) for t in ['z', 'x', 'y'] if self.hastag(t)]
        return sorted(ax)

    @property
    def ndim(self):
        return len(self.spatial_axes)

    @property
    def spatial_shape(self):
        return [self._shape[i] for i in self.spatial_axes]

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    @property
    def stripnone(self):
        return [s for s in self._shape if s is not None]

    @property
    def stripbatch_prod(self):
        return np.prod([s for s, t in zip(self._shape, self._tags) if t != 'b'])

    @property
    def stripnone_prod(self):
        return np.prod(self.stripnone)

    def tag2index(self, target_tag):
        try:
            return self._tags.index(target_tag)
        except ValueError:
            raise ValueError("Shape does not have tag %s, only tags %s"
                             % (target_tag, self._tags))

    def hastag(self, tag):
        return tag in self._tags

    def updateshape(self, axis, new_size, mode=None):
        i = axis if isinstance(axis, (int, np.integer)) else self.tag2index(axis)
        sh = list(self._shape)
        if mode is None:
            sh[i] = new_size
        elif sh[i] is not None:
            if mode == 'add':
                sh[i] += new_size
            elif mode == 'mult':
                sh[i] *= new_size
        ret = self.copy()
        ret._shape = sh
        return ret

    def updatefov(self, axis, new_fov):
        ret = self.copy()
        ret._fov[axis] = new_fov
        return ret

    def updatestrides(self, strides):
        ret = self.copy()
        ret._strides = np.array(strides)
        return ret

    def updatemfp_offsets(self, mfp_offsets):
        ret = self.copy()
        ret._mfp_offsets = mfp_offsets
        return ret

    def addaxis(self, axis, size, tag):
        i = axis if isinstance(axis, int) else self.tag2index(axis) + 1
        sh, tags = list(self._shape), list(self._tags)
        sh.insert(i, size)
        tags.insert(i, tag)
        return TaggedShape(sh, tags, self._strides, self._mfp_offsets, self._fov)

    def delaxis(self, axis):
        i = axis if isinstance(axis, int) else self.tag2index(axis) + 1
        sh, tags = list(self._shape), list(self._tags)
        sh.pop(i)
        tags.pop(i)
        return TaggedShape(sh, tags, self._strides, self._mfp_offsets, self._fov)

    def copy(self):
        return TaggedShape(self._shape, self._tags, self._strides,
                           self._mfp_offsets, self.fov)


class make_func(object):
    """Lazy "compiled function": builds a ``plan.Plan`` on first use.

    Parameters mirror graphutils.py:314-347: ``tt_input`` = list of source
    nodes, ``tt_output`` = node or list of nodes whose outputs are returned.
    ``step`` (optional) = an optimiser name: the plan then also runs backward
    and the update (what ``updates=`` did in the reference).
    """

    def __init__(self, tt_input, tt_output, updates=None, name='Unnamed Function',
                 borrow_inp=False, borrow_out=False, profile_execution=False,
                 model=None, step=None):
        self.tt_input = list(tt_input)
        self.single_return = not isinstance(tt_output, (list, tuple))
        self.tt_output = [tt_output] if self.single_return else list(tt_output)
        self.updates = updates
        self.name = name
        self.func = None
        self.last_exec_time = None
        self.profile_execution = profile_execution
        self._model = model
        self._step = step

    def compile(self, profile=False):
        from . import plan
        self.func = plan.Plan(self.tt_input, self.tt_output, model=self._model,
                              step=self._step, name=self.name)

    def __call__(self, *args):
        if self.func is None:
            self.compile()
        if len(args) == 0:          # graphutils.py:353-354: compile only
            return None
        t0 = time.time()
        ret = self.func(*args)
        self.last_exec_time = time.time() - t0
        if self.func.last_device_time is not None:
            self.last_exec_time = self.func.last_device_time
        return ret[0] if self.single_return else ret
