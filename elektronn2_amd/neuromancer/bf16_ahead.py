"""bf16 mode: the operands of the GEMM launches made AHEAD of them ("the
producers' epilogues"; csrc/bf16_prod.hip, e2_*_bf16_ex in include/e2hip.h).

The kernels with bf16 operands in memory (csrc/conv_bf16.hip, csrc/wgrad_bf16.hip) read three
kinds of images: channels-last pixels of the input (forward: x; data gradient: the zero-padded
dy), channel-major planes of dy (weight gradient), and packed filter rows.  Until round 4 every
launch converted its f32 operands first (prep_bf16_kernel, wgrad_bf16_cvt_kernel).  Once the
tilings of a Conv node's launches are known (after the first eager step, or at once from the
shipped table) this module gives the node buffers of its own and from then on
  * ONE launch at the start of the step packs the filter rows of every layer and direction,
  * the kernel that produces a layer's input (the bias / activation / pooling pass of the layer
    before it, or that layer's conv epilogue) also writes its channels-last bf16 image,
  * the kernel that produces a layer's pre-activation gradient (the activation / pooling
    backward) also writes -- or only writes -- the padded channels-last image and the planes,
and the GEMM launches find everything ready.  A node or direction whose tiling is not a
"32,..." form, or whose neighbours are not plain Conv nodes, keeps the converting calls.

Reference seam: unchanged -- the tensors are those of neural.py:662-712 (conv -> pool -> bias ->
activation) and of T.grad of that chain (model.py:182)."""
import numpy as np
import torch

from .. import autotune, backend



def _known(plan, kind, sig):
    return autotune.known(plan.ctx, kind, sig)


def _zeros_u8(plan, nbytes):
    return torch.zeros(int(nbytes), dtype=torch.uint8, device=plan.ctx.device)


def conv_nodes(plan):
    from .neural import Conv
    for n in plan.nodes:
        if type(n) is Conv and hasattr(n, '_k3') and n.parent is not None:
            yield n


def eligible(plan, node):
    """a Conv node that runs the generic conv path of the plan (not the fused first layer, the
    fused head or the tail), without batch normalisation / MFP, relu or lin"""
    return (not node._fused_first(plan) and node._fused_head(plan) is None
            and node._tail(plan) is None and not node._bn() and not node._mfp_pool()
            and node.activation_func in ('relu', 'lin')
            and tuple(node._p3) in node._PART_WINDOWS)


def prepare(plan):
    """idempotent; called before a run while no capture is in progress"""
    ctx = plan.ctx
    if getattr(ctx, 'mfma_dtype', 'f32') != 'bf16' or not plan._bf16_ahead_on:
        return
    a = plan.bf16a
    if not a:
        # engage once every launch's tiling is known -- and, with option bf16_ahead_min > 0, only
        # where the kernels with bf16 operands in memory carry that share of the GEMM launches
        # (round 4's rule, 0.6; since the side stream runs on a queue of its own the producers pay
        # on neuro3d@185 too, whose small late layers run the operand-rounding kernels: default 0)
        mem = tot = 0
        for node in conv_nodes(plan):
            if not eligible(plan, node) or plan.out[node.parent] is None:
                continue
            sigs = [('igemm', node._sig_fwd(plan))]
            if plan.training and (node, 'dy') in plan.scratch and (node, 'dy_pad') in plan.scratch:
                sigs.append(('wgrad', node._sig_wgrad(plan)))
                if plan.needs_grad(node.parent):
                    sigs.append(('igemm', node._sig_dgrad(plan)))
            for kind, sig in sigs:
                t = _known(plan, kind, sig)
                if t is None:
                    return                    # (not tuned yet: after the first eager step)
                tot += 1
                mem += t.startswith('32,')
        if tot == 0 or mem < float(plan.opt['bf16_ahead_min']) * tot:
            plan._bf16_ahead_on = False
            return
    new_jobs = False
    for node in conv_nodes(plan):
        if not eligible(plan, node):
            continue
        x = plan.out[node.parent]
        if x is None or not x.is_contiguous():
            continue
        cin = node.parent.shape['f']
        k = tuple(node._k3)
        w5 = node._w5(plan.param(node.w))
        fused = node._fused_act(plan)
        out = plan.out[node] if fused else plan.scratch[node, 'y']
        # ---- forward ---------------------------------------------------------------------
        if (node, 'fwd') not in a:
            t = _known(plan, 'igemm', node._sig_fwd(plan))
            tile = backend.Context.bf16_tile(t)
            if tile is not None and plan.bf16_xkeep(node) is not None:
                wb = _zeros_u8(plan, ctx.conv_bf16_wb_bytes(node.n_f, cin, k, x.shape[4], out.shape[4], tile))
                a[node, 'fwd'] = dict(tile=t, wb=wb, job=(w5, 0, x.shape[4], out.shape[4], tile, wb))
                new_jobs = True
        if not plan.training:
            continue
        dy = plan.scratch.get((node, 'dy'))
        dyp = plan.scratch.get((node, 'dy_pad'))
        if dy is None or dyp is None:
            continue
        # ---- data gradient -----------------------------------------------------------------
        want_d = plan.needs_grad(node.parent)
        if want_d and (node, 'dgrad') not in a:
            t = _known(plan, 'igemm', node._sig_dgrad(plan))
            tile = backend.Context.bf16_tile(t)
            if tile is not None:
                wb = _zeros_u8(plan, ctx.conv_bf16_wb_bytes(cin, node.n_f, k, dyp.shape[4], x.shape[4], tile))
                kg = (node.n_f + 15) // 16 * 2
                N, _, Dp, Hp, Wp = dyp.shape
                cl = _zeros_u8(plan, N * Dp * kg * Hp * Wp * 16 + 256)
                a[node, 'dgrad'] = dict(tile=t, wb=wb, cl=cl, cl_dims=(kg, Dp, Hp, Wp),
                                        job=(w5, 1, dyp.shape[4], x.shape[4], tile, wb))
                new_jobs = True
        # ---- weight gradient ---------------------------------------------------------------
        if (node, 'wgrad') not in a:
            t = _known(plan, 'wgrad', node._sig_wgrad(plan))
            if t and t.startswith('32,'):
                plane, dbytes, sbytes = ctx.wgrad_bf16_geometry(x.shape, node.n_f, k)
                a[node, 'wgrad'] = dict(tile=t, dyc=_zeros_u8(plan, dbytes),
                                        sums=torch.zeros(sbytes // 4, device=ctx.device),
                                        plane=plane, pitch=x.shape[4])
        # ---- the kernel that produces dy writes the images -----------------------------------
        # ... where that takes a conversion pass off the data-gradient chain (the chain the step
        # waits for); a layer whose data gradient reads f32 keeps the f32 pass + the weight
        # gradient's own conversion on the side stream (measured: neuro3d@185 1.22 -> 1.15 ms)
        if (node, 'dy') not in a and ((node, 'dgrad') in a or
                                      ((node, 'wgrad') in a and (not want_d or plan.opt['bf16_ahead_wgrad_only']))):
            d, wg = a.get((node, 'dgrad')), a.get((node, 'wgrad'))
            pad = [kk - 1 for kk in k]
            dst = backend.bf16_dst(cl=d['cl'] if d else None, cl_dims=d['cl_dims'] if d else None,
                                   cl_off=pad, pl=wg['dyc'] if wg else None,
                                   pl_plane=wg['plane'] if wg else 0, pl_pitch=wg['pitch'] if wg else 0)
            a[node, 'dy'] = dict(dst=dst, want_f32=not (wg is not None and (d is not None or not want_d)))
    # ---- who produces a layer's input image ----------------------------------------------------
    for node in conv_nodes(plan):
        f = a.get((node, 'fwd'))
        if f is None or 'producer' in f:
            continue
        f['producer'] = None
        par = node.parent
        from .neural import Conv
        if type(par) is not Conv or not hasattr(par, '_k3') or par.parent is None:
            continue
        if par._fused_first(plan):
            if par.n_f > 32:              # (the matrix-core form of the fused first layer only)
                continue
        elif not eligible(plan, par):
            continue
        if tuple(plan.out[par].shape) != tuple(plan.out_shape(par)) or (par, 'next') in a:
            continue                      # (a second consumer of the same tensor converts by itself)
        kgn = (par.n_f + 15) // 16 * 2
        if not par._fused_first(plan) and par._fused_act(plan):
            # the parent's conv epilogue writes it -- when that launch is a memory form too
            pf = a.get((par, 'fwd'))
            if pf is None:
                continue
        a[par, 'next'] = dict(consumer=node, buf=plan.bf16_xkeep(node), kg=kgn)
        f['producer'] = par
    if new_jobs:
        jobs = [v['job'] for (n, kind), v in a.items() if kind in ('fwd', 'dgrad')]
        plan._bf16_wjobs = ctx.make_bf16_wjobs(jobs) if jobs else None


def next_image(plan, node):
    """(buffer, channel groups, consumer) of the next layer's input image that `node` is to
    write, or None"""
    nx = plan.bf16a.get((node, 'next'))
    if nx is None:
        return None
    return nx['buf'], nx['kg'], nx['consumer']


def next_dst(plan, node):
    """the e2_bf16_dst of the next layer's input image for node's pooling / activation pass"""
    nx = plan.bf16a.get((node, 'next'))
    if nx is None:
        return None
    if 'dst' not in nx:
        osh = plan.out_shape(node)
        nx['dst'] = backend.bf16_dst(cl=nx['buf'], cl_dims=(nx['kg'], osh[2], osh[3], osh[4]))
    return nx['dst']
