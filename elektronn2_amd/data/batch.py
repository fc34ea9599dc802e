"""Batch creation from volumes resident on the GPU: the part of
``BatchCreatorImage.getbatch`` (data/cnndata.py:214-402) that is on the training path --
draw a cube, cut a (warped) patch and its centred target, grey-augment the image,
sub-sample the target by the net's output strides -- without the file / HDF5 / KNOSSOS
plumbing, lazy labels, affinities and non-geometric blob augmentations."""
from __future__ import annotations

import numpy as np
import torch

from . import transformations
from .transformations import WarpingOOBError


def grey_augment(d, channels, rng, ctx=None):
    """cnndata.py:42-60 on a device tensor (f, z, x, y):
    ``d[ch] = clip(d[ch] * alpha + c, 0, 1) ** gamma`` with alpha ~ 1 +- 0.15,
    c ~ +- 0.15, gamma ~ 2**U(-1, 1), one triple per listed channel; returns a new
    tensor (the patch is a fresh buffer anyway, so it is modified in place)."""
    if not channels:
        return d
    if ctx is None:
        from ..neuromancer.plan import get_ctx
        ctx = get_ctx()
    k = len(channels)
    alpha = 1 + (rng.rand(k) - 0.5) * 0.3
    c = (rng.rand(k) - 0.5) * 0.3
    gamma = 2.0 ** (rng.rand(k) * 2 - 1)
    for i, ch in enumerate(channels):
        ctx.grey_augment(d[ch], alpha[i], c[i], gamma[i])
    return d


class PatchSampler(object):
    """``getbatch(batch_size, source, grey_augment_channels, warp, warp_args)`` ->
    ``(data (b, f, z, x, y), target (b, f_t, z', x', y'))`` device tensors.

    data / targets: lists of (f, z, x, y) arrays (uploaded once); targets must be
    centred in their images.  ``patch_size`` = the input node's spatial shape,
    ``strides`` / ``offsets`` = the target node's (cnndata.py:134-140)."""

    @classmethod
    def from_nodes(cls, input_node, target_node=None, data=None, targets=None,
                   cube_prios=None, valid_cubes=None, aniso_factor=2, target_vec_ix=None,
                   target_discrete_ix=None, seed=None, border_mode='crop', d_path=None,
                   l_path=None, d_files=None, l_files=None, h5stream=False, zxy=True):
        """The reference constructor protocol (``BatchCreatorImage(input_node,
        target_node, **data_init_kwargs)``, cnndata.py:110-176): the geometry is read from
        the nodes -- ``patch_size = input_node.shape.spatial_shape``, ``strides`` /
        ``offsets`` of ``target_node.shape`` (cnndata.py:134-140) -- so a reference
        config's ``data_init_kwargs`` dict drives it unchanged.  The cubes are given as
        in-memory arrays (``data`` / ``targets``: lists of (f, z, x, y)); the file keys of
        the dict (``d_path``, ``l_path``, ``d_files``, ``l_files``, ``h5stream``, ``zxy``,
        ``border_mode``) belong to the HDF5 loader, which is outside the hot path, and are
        accepted and ignored."""
        if target_node is None:
            raise ValueError("from_nodes: a target node is required (img-img mode)")
        if data is None or targets is None:
            raise ValueError("from_nodes: pass the cubes as data=[...], targets=[...] "
                             "(HDF5 loading is not part of this build)")
        if len(data) != len(targets):
            raise ValueError("d_files and l_files must be lists of same length!")
        if input_node.shape.ndim != target_node.shape.ndim:
            raise ValueError("img-scalar mode is outside the hot path")
        if target_vec_ix is not None:
            raise NotImplementedError("vector targets (target_vec_ix) are outside the hot path")
        return cls(data, targets, input_node.shape.spatial_shape, target_node.shape.strides,
                   target_node.shape.offsets, aniso_factor=aniso_factor,
                   target_discrete_ix=target_discrete_ix, seed=seed,
                   valid=valid_cubes if valid_cubes is not None else (),
                   cube_prios=cube_prios)

    def __init__(self, data, targets, patch_size, strides, offsets, aniso_factor=2,
                 target_discrete_ix=None, seed=None, valid=(), cube_prios=None):
        from ..neuromancer.plan import get_ctx
        self.ctx = get_ctx()
        dev = self.ctx.device
        self.d = [torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev) for a in data]
        self.t = [torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev) for a in targets]
        self.valid = list(valid)
        self.train = [i for i in range(len(self.d)) if i not in self.valid]
        self.patch_size = tuple(int(p) for p in patch_size)
        self.strides = tuple(int(s) for s in strides)
        self.offsets = tuple(int(o) for o in offsets)
        self.target_ps = tuple(p - 2 * o for p, o in zip(self.patch_size, self.offsets))
        self.aniso_factor = aniso_factor
        self.target_discrete_ix = target_discrete_ix
        self.rng = np.random.RandomState(seed)
        if cube_prios is None:               # proportional to the cube sizes, cnndata.py:535-541
            w = np.array([self.t[i][0].numel() for i in self.train], np.float64)
        else:
            w = np.array([cube_prios[i] for i in self.train], np.float64)
        self._sampling_weight = np.hstack((0, np.cumsum(w / w.sum())))   # cnndata.py:576-583
        self.n_failed_warp = 0
        self.n_successful_warp = 0

    def _getcube(self, source):
        if source == 'train':
            p = self.rng.rand()
            i = self.train[int(np.flatnonzero(self._sampling_weight <= p)[-1])]
        elif source == 'valid':
            if not self.valid:
                raise ValueError("No validation set")
            i = self.valid[self.rng.randint(0, len(self.valid))]
        else:
            raise ValueError("Unknown data source")
        return self.d[i], self.t[i]

    def getbatch(self, batch_size=1, source='train', grey_augment_channels=None, warp=False,
                 warp_args=None, force_dense=False, affinities=False, nhood=None):
        """cnndata.py:214-402.  ``affinities='malis'``: the batch is
        ``(images, aff, seg)`` for a MalisNLL loss -- affinity graph and relabelled IDs of
        the first target channel (cnndata.py:377-382; computed on the host, the ID
        volumes are small); ``'affinity'``: ``(images, aff)``."""
        grey_augment_channels = grey_augment_channels or []
        warp_args = dict(warp_args or {})
        n_f, n_t = self.d[0].shape[0], self.t[0].shape[0]
        dev = self.ctx.device
        images = torch.empty((batch_size, n_f) + self.patch_size, dtype=torch.float32, device=dev)
        target = torch.empty((batch_size, n_t) + self.target_ps, dtype=torch.float32, device=dev)
        count = 0
        while count < batch_size:
            d, t = self._getcube(source)
            if warp is True or warp == 1:
                do_warp = True
            elif 0 < warp < 1:
                do_warp = bool(self.rng.rand() < warp)
            else:
                do_warp = False
            args = dict(warp_args)
            if not do_warp:
                args['warp_amount'] = 0
            try:
                transformations.get_warped_slice(
                    d, self.patch_size, aniso_factor=self.aniso_factor, target=t,
                    target_ps=self.target_ps, target_discrete_ix=self.target_discrete_ix,
                    rng=self.rng, out=images[count], target_out=target[count], **args)
                self.n_successful_warp += 1
            except WarpingOOBError:
                self.n_failed_warp += 1
                continue
            if source == 'train':
                grey_augment(images[count], grey_augment_channels, self.rng, self.ctx)
            count += 1
        if not (force_dense or all(s == 1 for s in self.strides)):
            target = target[:, :, ::self.strides[0], ::self.strides[1], ::self.strides[2]]
        if affinities in ('malis', 'affinity'):
            from .image import make_affinities
            ids = np.rint(target[:, 0].cpu().numpy()).astype(np.int32)
            aff, seg = make_affinities(ids, nhood)
            aff = torch.from_numpy(aff.astype(np.float32)).to(dev)
            if affinities == 'affinity':
                return images, aff
            return images, aff, torch.from_numpy(seg[:, None].astype(np.float32)).to(dev)
        return images, target


class RingFeeder(object):
    """Keeps the input ring of a training plan (Plan.set_input_ring) filled ``k`` batches at a time,
    one launch ahead of ``Model.trainingsteps(k, ring=feeder.ring, sync=False)``:

        feeder = RingFeeder(sampler, model, 'Adam', k, grey_augment_channels=[0], warp=0.5)
        feeder.fill()                                   # the first k slots
        for b in range(n_launches):
            losses, t = model.trainingsteps(k, optimiser='Adam', ring=feeder.ring, sync=False)
            feeder.fill()                               # the other half, while the launch runs

    The ring has 2 k slots; ``fill`` writes the k slots the NEXT launch will read.  They were last
    read by the launch before the one just submitted, whose losses ``trainingsteps(sync=False)``
    has just waited for -- so the slots are free without any further synchronisation.  The
    reference's counterpart is the BackgroundProc queue in front of trainingstep
    (training/trainer.py:174-186)."""

    def __init__(self, sampler, model, optimiser, k, **getbatch_kwargs):
        opt = model.optimisers[optimiser]
        plan = opt.step.func
        if plan is None or not plan._built:
            raise RuntimeError("RingFeeder: call trainingstep once first (it builds the plan)")
        self.sampler, self.plan, self.k, self.kw = sampler, plan, int(k), getbatch_kwargs
        self.ring = torch.zeros(2 * self.k, plan.input_arena.numel(), device=plan.ctx.device)
        plan.set_input_ring(self.ring)
        self.next = plan.ring_position()        # the step that reads the next slot to fill
        self.batch = plan.batch

    def fill(self):
        n = self.ring.shape[0]
        for j in range(self.k):
            batch = self.sampler.getbatch(self.batch, 'train', **self.kw)
            row = self.ring[(self.next + j) % n]
            for node, src in zip(self.plan.inputs, batch):
                o, cnt = self.plan.input_slices[node]
                row[o:o + cnt].copy_(src.reshape(-1), non_blocking=True)
        self.next += self.k

