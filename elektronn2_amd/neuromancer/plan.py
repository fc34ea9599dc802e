"""The "compiled function": a static launch plan over libe2hip.so.

This replaces what ``theano.function`` did behind
elektronn2/neuromancer/graphutils.py:376-387 and -- for training plans -- what
``T.grad`` + the optimiser ``updates`` did (model.py:182-199,
optimiser.py:301-334).  Shapes are static once the batch size is known, so a
plan is: one device buffer per node output (views for Crop), one zero-padded
gradient buffer per Conv, two flat arenas (parameters, gradients) owned by the
model, and a fixed kernel sequence.  After one eager warm-up call the sequence
is captured into a hipGraph (``e2_graph_*``) and replayed; inputs are copied
into static buffers before each replay.

Data-parallel training (new; the reference has none): when
``torch.distributed`` is initialised the gradient arena is all-reduced (mean)
between the backward graph and the optimiser graph -- one RCCL collective per
step over the single flat buffer (3.5 MB for neuro3d_lite, 11 MB for neuro3d).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import backend

_ctx = None


def get_ctx():
    """The process-wide HIP context (device from LOCAL_RANK, default 0)."""
    global _ctx
    if _ctx is None:
        import os
        from .. import parallel
        dev = parallel.local_device()     # raises for a LOCAL_RANK without a GPU under RCCL
        _ctx = backend.Context(dev)
        if os.environ.get("E2_MFMA_DTYPE", "f32") not in ("f32", "float32"):
            _ctx.set_mfma_dtype(os.environ["E2_MFMA_DTYPE"])
    return _ctx


def set_mfma_dtype(dtype):
    """'f32' (default; exact f32 MFMA) or 'bf16': the convolution GEMMs round their
    operands to bf16 on the way into the matrix core and accumulate in f32 (SURVEY.md
    8f-3; the reference has no counterpart -- Theano computes in float32).  Process-wide;
    plans captured earlier keep the arithmetic they were captured with, so set it before
    the first call of a model.  Also: environment ``E2_MFMA_DTYPE=bf16``."""
    get_ctx().set_mfma_dtype(dtype)


class Plan(object):
    def __init__(self, inputs, outputs, model=None, step=None, name='', use_graph=True, options=None):
        # the host switches, SNAPSHOT at construction (options.py): what is set later -- in the
        # environment or through set_plan_options -- never reaches a plan that exists
        from . import options as _options
        self.opt = _options.snapshot(options)
        self.inputs = list(inputs)
        self.outputs = list(outputs)
        self.step = step                       # None | 'SGD' | 'Adam' | 'grad'
        self.training = step is not None
        self.name = name
        self.model = model if model is not None else getattr(self.outputs[0], '_model', None)
        if self.model is None:
            raise RuntimeError("plan: nodes are not registered in a model")
        self.use_graph = use_graph and self.opt['graph']
        self.ctx = None
        self.batch = None
        self.last_device_time = None
        self._built = False
        # topological node list = union of all ancestors of the outputs
        seen, order = set(), []
        for o in self.outputs:
            for n in o.all_parents.values():
                if id(n) not in seen:
                    seen.add(id(n))
                    order.append(n)
        self.nodes = order
        self.loss_node = self.outputs[0] if self.training else None
        self._loss_anc = (set(id(n) for n in self.loss_node.all_parents.values())
                          if self.training else set())
        self._ng_cache = {}
        self._dp_cut_cache = False

    # ---- allocation helpers ---------------------------------------------------
    SLACK = 32       # floats of ZEROED slack behind every tensor of the plan (e2_set_input_slack)

    def empty(self, shape):
        """an uninitialised tensor followed by SLACK zeros: the position-split weight gradient
        (csrc/conv_pw_wgrad.hip, "MT,NT,9,0,S") reads up to 31 floats behind its input"""
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape)) if shape else 1
        flat = torch.empty(n + self.SLACK, dtype=torch.float32, device=self.ctx.device)
        flat[n:].zero_()
        if self.opt['debug_poison']:
            flat[:n].fill_(float('nan'))       # (a read before the first write shows up as NaN)
        return flat[:n].view(shape)

    def zeros(self, shape):
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape)) if shape else 1
        return torch.zeros(n + self.SLACK, dtype=torch.float32, device=self.ctx.device)[:n].view(shape)

    def full(self, shape, v):
        return torch.full(tuple(int(s) for s in shape), float(v), dtype=torch.float32,
                          device=self.ctx.device)

    def slack_behind(self, t):
        """bytes of the tensor's own allocation behind the view's last element: what the plan may
        promise to e2_set_input_slack.  Tensors from empty / zeros / empty_flat / the input arena
        end in SLACK zeroed floats; a view into one of them (Crop, a concat slice) is followed by
        more of the written buffer.  Anything else (torch.* allocations without slack) yields its
        true remainder, possibly 0 -- the position-split weight gradient is then not offered and
        the library's own tiling runs."""
        last = t.storage_offset() + sum((int(n) - 1) * int(st) for n, st in zip(t.shape, t.stride()))
        return int(t.untyped_storage().nbytes()) - 4 * (last + 1)

    def empty_flat(self, n):
        flat = torch.empty(int(n) + self.SLACK, dtype=torch.float32, device=self.ctx.device)
        flat[int(n):].zero_()
        if self.opt['debug_poison']:
            flat[:int(n)].fill_(float('nan'))
        return flat[:int(n)]

    def zeros_flat(self, n):
        return torch.zeros(int(n), dtype=torch.float32, device=self.ctx.device)

    def out_shape(self, node):
        """DEVICE shape of a node's output: always (b, f, z, x, y).  Nodes with fewer spatial
        axes (config 1: 'b,f,y,x' images, 'b,f' Perceptron outputs) get unit axes behind
        'f', so every kernel sees the 5-D layout it is written for."""
        sh = tuple(self.batch if s is None else int(s) for s in node.shape.shape)
        if len(sh) == 5:
            return sh
        tags = tuple(node.shape.tags)
        if len(sh) < 2 or len(sh) > 5 or tags[0] != 'b' or tags[1] != 'f':
            raise NotImplementedError("axis order %s: the device layout is (b, f, spatial...)"
                                      % (",".join(tags),))
        return sh[:2] + (1,) * (5 - len(sh)) + sh[2:]

    @staticmethod
    def user_view(node, t):
        """a device tensor in the rank the node declares (drops the unit axes of out_shape)"""
        n = len(node.shape.shape)
        if n == 5 or t.dim() != 5:
            return t
        return t.reshape(tuple(t.shape[:2]) + tuple(t.shape[5 - (n - 2):]))

    def alloc_out(self, node):
        self.out[node] = self.empty(self.out_shape(node))
        if self.training and self.needs_grad(node):
            self.alloc_grad(node)

    def alloc_grad(self, node):
        self.grad[node] = self.empty(self.out_shape(node))

    def tmp_like(self, t):
        key = ('tmp', tuple(t.shape))
        if key not in self.scratch:
            self.scratch[key] = self.empty(t.shape)
        return self.scratch[key]

    def needs_grad(self, node):
        """gradient wrt this node's output is required: it is on the loss path
        and some trainable parameter lies upstream of (or in) it."""
        if not self.training or id(node) not in self._loss_anc:
            return False
        r = self._ng_cache.get(id(node))
        if r is None:
            r = len(node.all_trainable_params) > 0
            self._ng_cache[id(node)] = r
        return r

    def param(self, p):
        return self.model.device_param(p)

    # ---- bf16 mode: the kernel with bf16 operands in memory (csrc/conv_bf16.hip) ----------
    def bf16_cands(self, k_channels):
        """extra tuner candidates of a conv launch in bf16 mode"""
        from .. import autotune
        if getattr(self.ctx, 'mfma_dtype', 'f32') != 'bf16':
            return []
        return autotune.bf16_memory_candidates(k_channels)

    def bf16_wgrad_cands(self, cin, k):
        """extra tuner candidates of a weight-gradient launch in bf16 mode"""
        from .. import autotune
        if getattr(self.ctx, 'mfma_dtype', 'f32') != 'bf16':
            return []
        return autotune.bf16_wgrad_candidates(cin, k)

    def _bf16_ws_alloc(self):
        """both bf16 scratch buffers, sized ONCE for the largest Conv node of the plan before
        any launch (they used to grow on demand during the first eager step: the old block
        went back to the caching allocator while a side-stream kernel could still read it)"""
        need_w = need_c = 0
        for n in self.nodes:
            if type(n).__name__ != 'Conv' or not hasattr(n, '_k3') or n.parent is None:
                continue
            try:
                psh = self.out_shape(n.parent)
                need_w = max(need_w, self.ctx.wgrad_bf16_ws_bytes(psh, n.n_f, n._k3))
                need_c = max(need_c, self.ctx.conv_bf16_ws_bytes(psh, n.n_f, n._k3))
            except Exception:
                continue
        if need_w:
            self.scratch['bf16_wgrad_ws'] = torch.empty(need_w, dtype=torch.uint8, device=self.ctx.device)
        if need_c:
            self.scratch['bf16_ws'] = torch.empty(need_c, dtype=torch.uint8, device=self.ctx.device)

    def bf16_xkeep(self, node, only_valid=False):
        """the per-layer buffer that keeps the forward's channels-last bf16 copy of the layer's
        input for its weight gradient (zero-filled once: the pixels behind the last plane are
        never written); ``only_valid``: None unless this step's forward filled it"""
        if not self.opt['bf16_xkeep']:
            return None
        if only_valid and not self._xkeep_valid.get(node, False):
            return None
        buf = self.scratch.get((node, 'xb_keep'))
        if buf is None:
            if only_valid:
                return None
            if self._capturing:
                raise RuntimeError("bf16 kept-copy buffer must exist before capture")
            need = self.ctx.conv_bf16_xkeep_bytes(self.out_shape(node.parent), node._k3)
            buf = torch.zeros(need, dtype=torch.uint8, device=self.ctx.device)
            self.scratch[node, 'xb_keep'] = buf
        return buf

    def bf16_xkeep_valid(self, node, ok):
        self._xkeep_valid[node] = bool(ok)

    def bf16_wgrad_ws(self, node):
        """the scratch of csrc/wgrad_bf16.hip (bf16 copies of x and dy, the f32 sums): its own
        buffer, because the weight gradient runs on the side stream next to the data
        gradient that uses ``bf16_ws``; one for the plan, sized for the largest layer"""
        need = self.ctx.wgrad_bf16_ws_bytes(self.out_shape(node.parent), node.n_f, node._k3)
        ws = self.scratch.get('bf16_wgrad_ws')
        if ws is None or ws.numel() < need:
            if self._capturing:
                raise RuntimeError("bf16 weight-gradient workspace must exist before capture")
            if ws is not None:
                ws.record_stream(self.side)       # (a side-stream kernel may still use it)
            ws = torch.empty(need, dtype=torch.uint8, device=self.ctx.device)
            self.scratch['bf16_wgrad_ws'] = ws
        return ws

    def bf16_ws(self, node):
        """ONE scratch buffer for the bf16 planes / filter rows of every conv of the plan (the
        launches are ordered on one stream), sized for the largest layer"""
        need = self.ctx.conv_bf16_ws_bytes(self.out_shape(node.parent), node.n_f, node._k3)
        ws = self.scratch.get('bf16_ws')
        if ws is None or ws.numel() < need:
            if self._capturing:
                raise RuntimeError("bf16 workspace must exist before capture")
            ws = torch.empty(need, dtype=torch.uint8, device=self.ctx.device)
            self.scratch['bf16_ws'] = ws
        return ws

    @staticmethod
    def _w5(w):
        """conv weights as (n_f, n_in, kz, kx, ky): 2-D layers get a unit z axis"""
        return w if w.dim() == 5 else w.reshape(tuple(w.shape[:2]) + (1,) * (5 - w.dim()) +
                                                 tuple(w.shape[2:]))

    def pgrad(self, p):
        return self.model.device_grad(p)

    def tuned(self, kind, sig, cands, fn, fn_tune=None, out=None, fn_once=None):
        """launch a conv kernel with its autotuned tiling (tunes on first sight,
        never while a hipGraph capture is in progress).

        ``out``: the launch's output tensor.  A split-K launch zero-fills it first (a ~5 us
        kernel each); the eager run notes which outputs those are, the captured step
        zeroes them all with ONE launch at the start of their segment and tells the
        library to skip its own fill (``_zero_jobs``)."""
        from .. import autotune
        skip = (self._capturing and out is not None
                and out.data_ptr() in self._zero_batched.get(self._seg_idx, ()))
        if skip:
            self.ctx.set_skip_zero_fill(True)
        try:
            autotune.tuned_call(self.ctx, kind, sig, cands, fn, allow_tune=not self._capturing,
                                fn_tune=fn_tune, fn_once=fn_once)
        finally:
            if skip:
                self.ctx.set_skip_zero_fill(False)
        if out is not None and not self._capturing:
            ptr, n = self.ctx.conv_last_zero_fill()
            if n and ptr == out.data_ptr():
                self._zero_seen.append((self._seg_idx, ptr, n))

    def zero_early(self, t):
        """zero a flat tensor that nothing touches between the start of the current
        segment and this point (gradient arena, loss statistics): in the captured step it
        joins the segment's one multi-fill instead of costing a launch of its own"""
        if self._capturing and t.data_ptr() in self._zero_batched.get(self._seg_idx, ()):
            return
        self.ctx.fill(t, 0.0)
        if not self._capturing:
            self._zero_seen.append((self._seg_idx, t.data_ptr(), t.numel()))

    def _zero_jobs(self):
        """per segment: (device pointers, device counts, number) of the split-K outputs the
        last eager run zero-filled -- each exactly once in its segment (a scratch buffer
        that two launches share keeps its in-line fills)"""
        jobs, self._zero_batched = {}, {}
        by_seg = {}
        for seg, ptr, n in self._zero_seen:
            by_seg.setdefault(seg, []).append((ptr, n))
        for seg, lst in by_seg.items():
            ptrs = [p for p, _ in lst]
            uniq = [(p, n) for p, n in lst if ptrs.count(p) == 1]
            if len(uniq) < 2:
                continue
            dev = self.ctx.device
            jobs[seg] = (torch.tensor([p for p, _ in uniq], dtype=torch.int64, device=dev),
                         torch.tensor([n for _, n in uniq], dtype=torch.int64, device=dev),
                         len(uniq))
            self._zero_batched[seg] = set(p for p, _ in uniq)
        return jobs

    # ---- second stream ----------------------------------------------------------------
    def on_side(self, fn, always=False, defer=False, force=False):
        """run the launches of ``fn`` on the side stream, ordered after everything
        issued on the main stream so far.  ``always``: also when the general side-stream
        switch is off (the weight repack: it overlaps the first layer's HBM-bound kernel).
        ``defer``: see plan option side_defer"""
        if not (self.use_side or force or (always and self.use_side_pack)):
            return fn()
        self._flush_side()                 # (the side stream keeps the order of the on_side calls)
        self.ctx.stream_fork(self.side)    # the dependency is fixed HERE: main's launches so far
        self._side_dirty = True
        if defer and self.opt['side_defer']:
            # issue the launches later (at the next on_side / join_side), i.e. BEHIND the next
            # launches of the main stream: what they wait for is unchanged, but in the captured
            # graph the main chain's next kernel becomes the fork node's FIRST child -- the
            # runtime keeps first children on their parent's queue, so the main chain stays on
            # one queue instead of hopping (a hop costs 6-10 us, DESIGN finding 54)
            self._side_pending = fn
            return
        self._run_side(fn)

    def side_rank(self, node):
        """position of a Conv node among the conv layers whose weight gradient is a launch of its
        own (forward order: 0 = the one the backward pass reaches last)"""
        if self._side_order is None:
            from .neural import Conv
            self._side_order = [n for n in self.nodes if type(n) is Conv and hasattr(n, '_k3')
                                and n.parent is not None and not n._fused_first(self)
                                and n._fused_head(self) is None]
        try:
            return self._side_order.index(node)
        except ValueError:
            return 1 << 30

    def side_forced(self, node):
        """side stream off (f32 mode): does this conv's weight gradient run there all the same?
        Option side_mask (experiments: bit r = the conv of side_rank r), else the tuning table's
        per-problem flag (autotune.side_flag; option side_table)"""
        if self.use_side or self.opt['side_stream'] is False:
            return False
        m = int(self.opt['side_mask'])
        if m:
            r = self.side_rank(node)
            return r < 62 and bool((m >> r) & 1)
        if not self.opt['side_table']:
            return False
        from .. import autotune
        try:
            return autotune.side_flag(self.ctx, node._sig_wgrad(self))
        except Exception:
            return False

    def _run_side(self, fn):
        ctx = self.ctx
        main = ctx.stream
        ctx.set_stream(self.side)
        try:
            with torch.cuda.stream(self.side):
                fn()
        finally:
            ctx.set_stream(main)

    def _flush_side(self):
        fn, self._side_pending = self._side_pending, None
        if fn is not None:
            self._run_side(fn)

    def join_side(self):
        """main stream waits for the side stream (no-op when nothing ran there)"""
        self._flush_side()
        if self._side_dirty:
            self.ctx.stream_join(self.side)
            self._side_dirty = False

    # ---- gradient routing ----------------------------------------------------------
    def grad_slot(self, node):
        """(buffer, first): first == True -> the caller must OVERWRITE it."""
        first = id(node) not in self._grad_written
        self._grad_written.add(id(node))
        return self.grad[node], first

    def add_grad(self, node, src):
        if not self.needs_grad(node):
            return
        dst, first = self.grad_slot(node)
        self.ctx.copy5(src, dst, accumulate=not first)

    def add_grad_region(self, node, slicer, src):
        if not self.needs_grad(node):
            return
        dst, first = self.grad_slot(node)
        if first:
            self.zero_early(dst)      # (a gradient buffer: nothing touches it before its first writer)
        self.ctx.copy5(src, dst[slicer], accumulate=True)

    # ---- build ------------------------------------------------------------------------
    def build(self, batch):
        self.ctx = get_ctx()
        for g in list(getattr(self, '_graphs', None) or []) + list(getattr(self, '_multi', {}).values()):
            self.ctx.graph_destroy(g)        # batch size changed: drop the graphs captured for the old one
        self._graphs = None
        self.batch = int(batch)
        self.stream = torch.cuda.Stream(device=self.ctx.device)
        # second stream: weight gradients (and the weight repack) run next to the
        # data-gradient chain; inside the captured graph they are parallel branches
        self.side = torch.cuda.Stream(device=self.ctx.device)
        self._side_dirty = False
        self._side_pending = None
        self._side_order = None
        # measured: two MFMA-bound f32 kernels sharing the chip finish no sooner than
        # back to back (lite183 1.69 / 1.70 ms, neuro3d 2.04 / 2.15 with the branch), so f32 keeps
        # one stream; the bf16 kernels leave the matrix pipe idle most of the time and the
        # branch pays there (lite183 0.884 -> 0.855 ms, neuro3d 1.223 -> 1.198, unet3d 7.23 ->
        # 7.05).  Option side_stream = True / False overrides.
        self.use_side = self.opt['side_stream'] if self.opt['side_stream'] is not None else \
            (getattr(self.ctx, 'mfma_dtype', 'f32') == 'bf16')
        # the repack of the weight images as a parallel branch of the graph next to the first
        # layer (the first kernel that reads a packed image joins it, Conv._plan_fwd): measured
        # on one box, interleaved (tools/ab.sh), the fork / join costs more than the overlap
        # saves -- lite183 1.715 vs 1.699 ms, full185 2.120 vs 2.095 ms -- so it stays off
        self.use_side_pack = bool(self.opt['side_pack'])
        # activation backward of an un-pooled conv inside its consumer's data-gradient launch
        # (e2_conv3d_dgrad_packed_actbwd).  Measured (tools/ab.sh, DESIGN.md finding 17): the
        # mask loads in the GEMM epilogue cost what the removed 10 us kernel cost, and the
        # bias-gradient atomics of split-K grids cost more (neuro3d@185 +0.7 ms) -- off
        self.fuse_actbwd = int(self.opt['fuse_actbwd'])
        self.out, self.grad, self.scratch = {}, {}, {}
        # bf16 mode: operands of the GEMM launches made ahead of them (bf16_ahead.py)
        self.bf16a, self._bf16_wjobs = {}, None
        self._xb_ready, self._dy_ready, self._wb_ready = {}, {}, False
        self._bf16_ahead_on = bool(self.opt['bf16_ahead'])
        self._xkeep_valid = {}       # Conv node -> its kept bf16 input copy is this step's
        self.pack_jobs = []          # (param, packed image, mode) of every Conv node
        self.pack_nodes = {}         # id(packed image) -> (Conv node, mode): whose launch reads it
        self._pack_rows_done = False
        self._img_stride = {}        # id(packed image) -> floats per row it is packed with (absent: formula)
        self.model.ensure_arena(self.ctx)
        with torch.cuda.stream(self.stream):
            # the static input buffers of all Input nodes are slices of ONE allocation (each
            # 16-byte aligned): a producer that stages a batch in the same layout fills them
            # with a single copy (input_arena; bench.py does)
            offs, tot = [], 0
            for n in self.inputs:
                offs.append(tot)
                tot += (int(np.prod(self.out_shape(n))) + 3) // 4 * 4
            self.input_arena = self.empty_flat(max(tot, 4))
            self.input_slices = {}
            for n, o in zip(self.inputs, offs):
                sh = self.out_shape(n)
                self.out[n] = self.input_arena[o:o + int(np.prod(sh))].view(sh)
                self.input_slices[n] = (o, int(np.prod(sh)))
            for n in self.nodes:
                n._plan_alloc(self)
            if getattr(self.ctx, 'mfma_dtype', 'f32') == 'bf16':
                self._bf16_ws_alloc()
            self._pack_dev = self._pack_dev_up = self._upd = None
            if self.pack_jobs:
                jobs = [(self._w5(self.param(w)), wp, mode) for (w, wp, mode) in self.pack_jobs]
                self._pack_dev = self.ctx.make_pack_jobs(jobs)
                self._plan_fused_update(jobs)
        self._graphs = None
        self._segs = None
        self._segs_key = None
        self._dp_scale_mode = False          # (False = not decided yet; _segments decides)
        self._seg_idx = 0
        self._zero_seen, self._zero_batched, self._zero_keep = [], {}, None
        self._capturing = False
        self._calls = 0
        self._grad_written = set()
        # two pairs of timing events, used alternately: a deferred fetch (fetch_async) reads
        # the pair of the step BEFORE the one just submitted
        self._ev_pairs = [(self.ctx.event(), self.ctx.event()), (self.ctx.event(), self.ctx.event())]
        self._ev_idx = 0
        self._n_runs = 0                     # run() calls so far (fetch_async pairs slots with runs)
        self._async = None                   # state of fetch_async: pinned loss slots, events
        # several steps in one graph (run_steps): batches out of a device-side ring, losses into one
        self._ring = None                    # (n_slots, arena floats): the batches
        self._hist = None                    # (n_slots, 1): the losses
        self._step_state = None              # int64[2] on the device: steps started, arrivals (e2_step_prologue)
        self._multi = {}                     # k -> graph of k steps
        self._built = True

    @property
    def _ev0(self):
        return self._ev_pairs[self._ev_idx][0]

    @property
    def _ev1(self):
        return self._ev_pairs[self._ev_idx][1]

    # ---- kernel sequences ----------------------------------------------------------------
    def _plan_fused_update(self, jobs):
        """Adam plans: the optimiser launch itself writes the conv weight images
        (e2_adam_pack_step, csrc/update_pack.hip) -- the repack launch at the head of the step
        and its second pass over the weights are gone; what stays in `_pack_dev_up` are the UpConv
        images (other layouts).  The images are then current from step to step as long as nobody
        else writes the parameters: Model._img_owner names the plan whose last update wrote them,
        every other writer (set_value, load, broadcast, another optimiser) clears it, and
        _run_device repacks eagerly, outside the graphs, when it is not this plan."""
        if self.step != 'Adam' or not self.opt['adam_pack']:
            return
        m = self.model
        by_param, order, up = {}, [], []
        for (w, wp, mode), (w5, _, _) in zip(self.pack_jobs, jobs):
            if mode in (0, 1):
                if id(w) not in by_param:
                    if not getattr(w, 'apply_train', False) or id(w) not in m._slots:
                        return                       # (a frozen conv weight: keep the two launches)
                    by_param[id(w)] = [w, tuple(int(v) for v in w5.shape), None, None]
                    order.append(id(w))
                by_param[id(w)][2 + mode] = wp
            else:
                up.append((w5, wp, mode))
        if not order:
            return

        def mult(p):
            r = p.apply_reg
            return float(r) if (r and r is not True) else (1.0 if r else 0.0)
        ujobs = [(m._slots[k][0], by_param[k][2], by_param[k][3], by_param[k][1], mult(by_param[k][0]))
                 for k in order]
        rests = [(m._slots[id(p)][0], m._slots[id(p)][1], mult(p)) for p in m._param_list
                 if p.apply_train and id(p) not in by_param]
        self._upd = self.ctx.make_upd_jobs(ujobs, rests)
        self._pack_dev_up = self.ctx.make_pack_jobs(up) if up else None

    def image(self, wp):
        """context manager around the launches that read the packed image ``wp``: announces the
        row length the image was packed with (e2_set_image_rows), the formula otherwise"""
        plan, rows = self, self._img_stride.get(id(wp), 0)

        class _I(object):
            def __enter__(self_):
                if rows:
                    plan.ctx.set_image_rows(rows)

            def __exit__(self_, *a):
                if rows:
                    plan.ctx.set_image_rows(0)
        return _I()

    @staticmethod
    def _tile_rows(tiling, m):
        """how far the M tiles of a conv GEMM launch with this tiling string reach for m output
        rows, or 0 (= the repack's default) when the string is unknown / of another kernel family"""
        if not tiling:
            return 0
        try:
            v = [int(t) for t in tiling.split(",")]
        except ValueError:
            return 0
        if len(v) == 4:                                   # "MT,NT,CC,SK": 16x16x4 kernel
            per = 16 * v[0]
        elif len(v) == 8 and v[0] == 4:                   # "4,MG,NT,CC,SK,WM,WN,G": 4x4x1 kernel
            per = 4 * v[1] * v[5]
        elif len(v) in (3, 5) and v[0] == 1:              # "1,MT,NT[,KC,0]": pointwise GEMM
            per = 16 * v[1]
        elif v[0] == 32:                                  # bf16 operands in memory: image not read
            per = 16
        else:
            return 0
        # (whole 128-byte lines: a line the repack writes only in part is merged with memory when the
        # GEMM fetches it -- rows cut at 80 made the 5 x 2 tiles 6 % slower)
        return -(-(-(-m // per) * per) // 32) * 32

    def _refine_pack_rows(self):
        """after the first eager step the tiling of every conv launch is known (tuned or from the
        shipped table): the repack then rewrites, per image, only the padding rows THAT tiling
        fetches instead of the worst case over all tilings -- 1.5-3x fewer rows for the layers
        with few channels (the rows beyond stay zero from the one-time fill: correct for any
        tiling, only not refreshed in the memory-side cache if the tiling changes later)"""
        self._pack_rows_done = True
        want_rows, want_stride = self.opt['pack_rows'], self.opt['image_stride'] and self._upd is None
        if not (want_rows or want_stride) or self._pack_dev is None:
            return
        from .. import autotune
        rows, strides, any_ = [], [], False
        self._img_stride = {}
        for (w, wp, mode) in self.pack_jobs:
            r = st = 0
            ent = self.pack_nodes.get(id(wp))
            if ent is not None:
                node, md = ent
                try:
                    if md == 0:
                        t, m = autotune.known(self.ctx, 'igemm', node._sig_fwd(self)), node.n_f
                    elif self.needs_grad(node.parent):
                        t, m = autotune.known(self.ctx, 'igemm', node._sig_dgrad(self)), node.parent.shape['f']
                    else:
                        t, m = None, 0
                    r = self._tile_rows(t, m)
                    # a row length of its own only for the 16x16x4 / pointwise kernels (the 4x4x1
                    # kernel's lanes read 64 consecutive floats per row; a memory-form tiling does
                    # not read the image, but the launch that replaces it under another pin would)
                    v = [int(q) for q in t.split(",")] if t else []
                    # (measured for the 4x4x1 kernel too -- rows of 64 floats for 20 channels --:
                    # neuro3d_lite +5 us, neuro3d -5 us, not of one sign: its images keep the formula)
                    if want_stride and r and (len(v) == 4 or (len(v) in (3, 5) and v[0] == 1)):
                        # exactly as long as the tiles reach: measured against + 16 floats and against
                        # rows rounded to whole 128-byte lines, the densest form wins (finding 52)
                        st = -(-max(r, -(-m // 16) * 16) // 16) * 16
                except Exception:
                    r = st = 0
            rows.append(r if want_rows else 0)
            strides.append(st)
            if st:
                self._img_stride[id(wp)] = st
            any_ = any_ or st > 0 or (want_rows and r > 0)
        if any_:
            jobs = [(self._w5(self.param(w)), wp, mode) for (w, wp, mode) in self.pack_jobs]
            # (an image whose row length changes is re-zeroed first: its old rows sit elsewhere)
            for (w5, wp, mode), st in zip(jobs, strides):
                if st:
                    wp.zero_()
            self._pack_dev = self.ctx.make_pack_jobs(jobs, rows, strides)

    def _emit_forward(self):
        if self._upd is not None:
            # (conv images: written by this plan's optimiser launch, or by the eager repack of
            # _run_device when someone else touched the parameters)
            if self._pack_dev_up is not None:
                self.on_side(lambda: self.ctx.conv3d_pack_multi(*self._pack_dev_up), always=True)
        self._xb_ready, self._dy_ready = {}, {}
        self._wb_ready = False               # bf16 mode: see ensure_wb
        if self._upd is None and self._pack_dev is not None:     # all packed weight images, one launch
            def packs():
                self.ctx.conv3d_pack_multi(*self._pack_dev)
                if self.use_side and self.opt['wb_on_side']:
                    # bf16 mode: the filter rows of every layer too, beside the fused first layer
                    # (which reads neither) instead of behind it
                    self.ensure_wb()
            self.on_side(packs, always=True)
        for n in self.nodes:
            n._plan_fwd(self)
        self.join_side()

    def ensure_wb(self):
        """bf16 mode: every layer's packed filter rows by ONE launch per step, issued in front of
        the first launch that reads them (behind the fused first layer, which does not).
        Option bf16_wpack='call': each launch's own conversion pass packs its rows instead (A/B switch)"""
        if not self._wb_ready and self._bf16_wjobs is not None and \
                self.opt['bf16_wpack'] == "step":
            self.ctx.conv3d_bf16_pack_w_multi(*self._bf16_wjobs)
            self._wb_ready = True
        return self._wb_ready

    def _bwd_nodes(self):
        return [n for n in reversed(self.nodes)
                if id(n) in self._loss_anc and (self.needs_grad(n) or n is self.loss_node)]

    def _emit_backward(self, part=None):
        """part None: the whole backward pass; 0 / 1: the nodes before / after the
        data-parallel cut (see _dp_cut)"""
        nodes = self._bwd_nodes()
        if part is not None:
            k = self._dp_cut()[0]
            nodes = nodes[:k] if part == 0 else nodes[k:]
        if part in (None, 0):
            if not self._upd_zeroes_g():       # (else: the optimiser kernel left it zero)
                self.zero_early(self.model.G)
            self._grad_written = set()
        for n in nodes:
            n._plan_bwd(self)
        self.join_side()

    def _dp_cut(self):
        """(k, lo) or None: after the first k nodes of the backward pass every gradient
        in G[lo:] is final, so that slice can be exchanged while the other nodes run.
        The arena is laid out in node order, so the late layers (most of the parameters,
        least of the backward time) form its tail."""
        if self._dp_cut_cache is not False:
            return self._dp_cut_cache
        self._dp_cut_cache = None
        m = self.model
        if not self.opt['dp_overlap'] or m.n_train < (1 << 16):
            return None
        nodes = self._bwd_nodes()
        last = {}                                  # id(param) -> last backward index using it
        for i, n in enumerate(nodes):
            for p in n.params.values():
                if getattr(p, 'apply_train', False) and id(p) in m._slots:
                    last[id(p)] = i
        slots = sorted(((m._slots[q][0], q) for q in last), reverse=True)
        for k in range(1, len(nodes) - 1):
            lo = m.n_train
            for off, q in slots:
                if last[q] >= k:
                    break
                lo = off
            if m.n_train - lo >= 0.75 * m.n_train:
                self._dp_cut_cache = (k, lo)
                break
        return self._dp_cut_cache

    def _labelled_count(self):
        """one-element device view: the number of labelled target voxels of this rank's
        batch (written by the forward pass of the single MultinoulliNLL under the loss), or
        None when the loss has another structure"""
        ln = self.loss_node
        kids = ln.parent if isinstance(ln.parent, (list, tuple)) else [ln.parent]
        if len(kids) != 1 or type(kids[0]).__name__ != 'MultinoulliNLL':
            return None
        st = self.scratch.get((kids[0].pred, 'stats'))
        return None if st is None else st[1:2]

    def _emit_update(self):
        if self.step in ('Adam', 'SGD'):
            self.model.optimisers[self.step].device_update(self)

    def _upd_zeroes_g(self):
        """the optimiser launch of this plan clears the gradient arena for the next step
        (e2_adam_step_ex zero_g): no fill launch at the head of the backward pass.  G is
        zero on entry because the last step left it so -- or _run_device fills it first
        (Model._g_clean)."""
        return (self.training and self.step in ('Adam', 'SGD') and bool(self.opt['zero_in_update']))

    def _dp_key(self):
        """everything outside the plan that the segmentation and the captured graphs depend on:
        the data-parallel state of the model.  A change re-plans and re-captures (_run_device);
        between two changes the plan uses the values STORED by _segments -- never a fresh read
        (a flag flipped after capture made the host exchange scale by the count while
        the captured optimiser still divided by it)."""
        m = self.model
        if not self.training:
            return (1,)
        return (m.dp_world(), bool(getattr(m, '_dp_weighted', False)), bool(getattr(m, '_dp_force', False)),
                id(getattr(m, '_dp_group', None)))

    def _dp_scale_now(self):
        """how the data-parallel step normalises: None = the exchange object scales with
        elementwise launches (parallel.BucketedMean); ('sum',) = the loss kernels leave the
        gradients unnormalised and put the labelled count into the slot behind the arena, the
        collectives only SUM, the optimiser kernel divides by the summed count; ('mean', 1/w)
        = plain mean, the factor applied by the optimiser kernel."""
        if not (self.training and self.step in ('Adam', 'SGD') and self.model.dp_world() > 1
                and self.opt['dp_fused_scale'] and self.model.G.is_cuda):
            return None
        if getattr(self.model, '_dp_weighted', False):
            return ('sum',) if self._labelled_count() is not None else None
        import torch.distributed as dist
        return ('mean', 1.0 / dist.get_world_size(self.model._dp_group))

    def _dp_scale(self):
        """the mode decided when the step was last segmented (see _dp_key)"""
        if self._dp_scale_mode is False:
            self._dp_scale_mode = self._dp_scale_now()
        return self._dp_scale_mode

    def loss_grad_mode(self):
        """context manager around the NLL backward launches of a loss node"""
        plan = self

        class _M(object):
            def __enter__(self_):
                self_.on = plan._dp_scale() == ('sum',)
                if self_.on:
                    plan.ctx.set_loss_grad_mode(True, plan.model.G_spare)

            def __exit__(self_, *a):
                if self_.on:
                    plan.ctx.set_loss_grad_mode(False, None)
        return _M()

    def _segments(self):
        """The step as a list of (emit, after): ``emit`` issues launches on the plan's
        stream (each segment is captured into its own hipGraph), ``after`` is host code
        run between two segments -- the gradient exchange of the data-parallel step, or
        the host part of a node (MALIS: Kruskal on the CPU between forward and loss)."""
        dp = self.training and self.step in ('Adam', 'SGD') and self.model.dp_world() > 1
        self._dp_scale_mode = self._dp_scale_now()       # decided ONCE per segmentation
        host = [n for n in self.nodes if hasattr(n, '_plan_host')]
        upd = self.training and self.step in ('Adam', 'SGD')
        segs = []
        if host:
            def host_steps():
                for n in host:
                    n._plan_host(self)

            def rest():
                for n in host:
                    n._plan_fwd_post(self)
                if self.training:
                    self._emit_backward()
                    if upd and not dp:
                        self._emit_update()
            segs.append((self._emit_forward, host_steps))
            segs.append((rest, (lambda: self.model.allreduce_grads(
                self._labelled_count(), raw=self._dp_scale() is not None)) if dp else None))
            if dp:
                segs.append((self._emit_update, None))
            return segs
        cut = self._dp_cut() if dp else None
        if cut:
            # the late layers' gradients travel while the early layers' are computed
            n_train = self.model.G.numel()

            def seg0():
                self._emit_forward()
                self._emit_backward(0)

            def start_tail():
                self._ex = self.model.grad_exchange(self._labelled_count(),
                                                    raw=self._dp_scale() is not None)
                self._ex.start(cut[1], n_train)

            def finish():
                self._ex.start(0, cut[1])
                self._ex.finish()
            return [(seg0, start_tail), (lambda: self._emit_backward(1), finish),
                    (self._emit_update, None)]

        def whole():
            self._emit_forward()
            if self.training:
                self._emit_backward()
                if upd and not dp:
                    self._emit_update()
        segs.append((whole, (lambda: self.model.allreduce_grads(
            self._labelled_count(), raw=self._dp_scale() is not None)) if dp else None))
        if dp:
            segs.append((self._emit_update, None))
        return segs

    def _run_device(self):
        """fwd (+ bwd + all-reduce + update) on self.stream, graph-replayed."""
        ctx = self.ctx
        key = self._dp_key()
        if self._segs is None or self._segs_key != key:
            # (data parallelism switched on / its weighting changed after earlier steps: re-plan
            # and re-capture)
            self._segs, self._segs_key = self._segments(), key
            for g in list(self._graphs or []) + list(self._multi.values()):
                ctx.graph_destroy(g)
            self._multi = {}
            if self._graphs is not None:
                self._calls = 0          # one eager run of the new segmentation first
            self._graphs = None
            self._dp_cut_cache = False
        if self.training:
            if self._upd_zeroes_g() and not self.model._g_clean:
                ctx.fill(self.model.G, 0.0)            # (someone else wrote G: eager, outside the graphs)
            self.model._g_clean = False
        if self._upd is not None and getattr(self.model, '_img_owner', None) is not self:
            # (someone else wrote P since this plan's last update -- or this is its first step:
            # the conv images are packed eagerly, outside the graphs)
            ctx.conv3d_pack_multi(*self._pack_dev)
        if self._graphs is None:
            from . import bf16_ahead
            bf16_ahead.prepare(self)                   # (allocates: never during a capture)
            if self._calls >= 1 and not self._pack_rows_done:
                self._refine_pack_rows()               # (tilings are known after the eager step)
            elif self._calls == 0 and self._img_stride:
                # (the plan was reset to re-tune: back to images any tiling can read)
                self._img_stride, self._pack_rows_done = {}, False
                jobs = [(self._w5(self.param(w)), wp, mode) for (w, wp, mode) in self.pack_jobs]
                for (_, wp, _) in jobs:
                    wp.zero_()
                self._pack_dev = self.ctx.make_pack_jobs(jobs)
        capture = self.use_graph and self._calls >= 1
        if capture and self._graphs is None:
            # the first captured call runs segment by segment: a host step between two
            # segments needs the results of the one before it
            graphs = []
            zero_jobs = self._zero_jobs()
            self._zero_keep = zero_jobs                  # the graphs read these tensors
            ctx.record(self._ev0)
            try:
                for i, (emit, after) in enumerate(self._segs):
                    self._seg_idx = i
                    self._capturing = True
                    ctx.graph_begin()
                    try:
                        self._emit_seg(i, emit, zero_jobs)
                    except BaseException:
                        # leave capture mode (a stream stuck in capture poisons every later
                        # launch of the context) and drop the partial graph
                        self._capturing = False
                        try:
                            ctx.graph_destroy(ctx.graph_end())
                        except Exception:
                            pass
                        raise
                    g = ctx.graph_end()
                    self._capturing = False
                    graphs.append(g)
                    ctx.graph_launch(g)
                    if after is not None:
                        after()
            except BaseException:
                for g in graphs:
                    ctx.graph_destroy(g)
                raise
            self._graphs = graphs
            from .. import autotune
            autotune.save()
            ctx.record(self._ev1)
            self._calls += 1
            self.model._g_clean = self._upd_zeroes_g()
            self.model._img_owner = self if self._upd is not None else None
            return
        ctx.record(self._ev0)
        if not capture:
            self._zero_seen = []
        for i, (emit, after) in enumerate(self._segs):
            self._seg_idx = i
            if capture:
                ctx.graph_launch(self._graphs[i])
            else:
                self._emit_seg(i, emit, None)
            if after is not None:
                after()
        ctx.record(self._ev1)
        self._calls += 1
        if self.training:
            self.model._g_clean = self._upd_zeroes_g()
            if self.step in ('Adam', 'SGD'):       # (this step wrote P: whose images are current?)
                self.model._img_owner = self if self._upd is not None else None

    def _emit_seg(self, i, emit, zero_jobs):
        """the launches of segment i: (the step's prologue) + (batched zero fills) + the segment"""
        ctx = self.ctx
        if i == 0 and (self._ring is not None or self._hist is not None):
            # ONE launch: this step's batch out of the ring, the previous step's loss into the history
            ctx.step_prologue(self._step_state, ring=self._ring, dst=self.input_arena,
                              src=self._loss_dev() if self._hist is not None else None, hist=self._hist)
        if zero_jobs and i in zero_jobs:
            ctx.fill_multi(*zero_jobs[i])
        emit()

    def _loss_dev(self):
        nll = self.loss_node.parent[0] if isinstance(self.loss_node.parent, (list, tuple)) \
            else self.loss_node.parent
        return self.scratch[nll, 'loss']

    def _drop_graphs(self):
        """the captured graphs no longer describe the step (a ring was attached ...): capture again
        at the next run (no eager run in between: tilings and buffers are unchanged)"""
        if self._graphs or self._multi:
            self.stream.synchronize()        # (a launch of them may still be running)
        for g in list(self._graphs or []) + list(self._multi.values()):
            self.ctx.graph_destroy(g)
        self._graphs, self._multi = None, {}

    # ---- several steps per launch --------------------------------------------------------------
    def set_input_ring(self, ring):
        """The step takes its batch out of ``ring[(steps so far) % n_slots]`` -- a float32 device
        tensor (n_slots, input_arena.numel()), every slot in the layout of ``input_arena``
        (``input_slices``: image | target, 16-byte aligned slices) -- by a launch of its own graph
        (e2_step_prologue) instead of waiting for ``set_inputs``.  ``None`` detaches the ring.  The
        producer (data/batch.py's sampler on its stream, or host copies) fills slots AHEAD of the
        steps that read them and orders itself against the plan's stream; the reference's
        counterpart is the BackgroundProc queue of training/trainer.py:174-186."""
        if ring is not None:
            if not self._built:
                raise RuntimeError("set_input_ring: build the plan first (set_inputs / one step)")
            if not (isinstance(ring, torch.Tensor) and ring.is_cuda and ring.dtype == torch.float32
                    and ring.dim() == 2 and ring.is_contiguous()
                    and ring.shape[1] == self.input_arena.numel() and ring.data_ptr() % 16 == 0):
                raise ValueError("set_input_ring: a contiguous float32 device tensor (n_slots, %d) is needed"
                                 % self.input_arena.numel())
            self._ring = ring
        else:
            self._ring = None
        self._state_alloc()
        self._drop_graphs()

    def _state_alloc(self):
        # (the count goes on across attachments: step L reads slot L % n_slots; ring_position() tells L)
        if self._step_state is None:
            self._step_state = torch.zeros(2, dtype=torch.int64, device=self.ctx.device)

    def ring_position(self):
        """steps that have run their prologue (e2_step_prologue: the next step reads ring slot
        ring_position() % n_slots) since a ring or the history was first attached; reads the
        device count: waits"""
        if self._step_state is None:
            return 0
        self.stream.synchronize()
        return int(self._step_state[0].item())

    def keep_loss_history(self, n_slots=256):
        """every step's loss is kept in a device-side ring of ``n_slots`` entries (written by the
        NEXT step's prologue; the newest loss is the plan's loss scalar); ``loss_history(n)``
        reads the last n."""
        if not self.training:
            raise RuntimeError("keep_loss_history: a training plan is needed")
        if self._hist is None or self._hist.shape[0] != int(n_slots):
            self._hist = torch.zeros(int(n_slots), 1, device=self.ctx.device)
            self._state_alloc()
            self._hist_from = self.ring_position()   # steps before this one are not in the history
            self._drop_graphs()

    def loss_history(self, n):
        """the losses of the last n steps, oldest first (waits for the plan's stream)"""
        if self._hist is None:
            raise RuntimeError("loss_history: call keep_loss_history() before the steps")
        t = self.ring_position()
        ns = self._hist.shape[0]
        if n < 1 or n > min(t - self._hist_from, ns + 1):
            raise ValueError("loss_history: %d steps asked for, %d kept" % (n, min(t - self._hist_from, ns + 1)))
        h = self._hist[:, 0].cpu().numpy()
        out = [h[(t - n + j) % ns] for j in range(n - 1)] + [float(self._loss_dev().item())]
        return np.array(out, np.float32)

    def run_steps(self, k):
        """``k`` training steps on the device with ONE graph launch where the step is a single
        graph (no data-parallel exchange, no host part between its launches); otherwise k calls
        of run().  Batches: the input ring if one is attached, else every step re-reads the
        static input buffers.  Losses: keep_loss_history() is switched on; read them with
        loss_history(k).  Why: between two graph launches the device idles ~19 us (DESIGN
        findings 54, 55)."""
        k = int(k)
        if k < 1:
            return
        if not self.training:
            raise RuntimeError("run_steps: a training plan is needed")
        if self._hist is None:
            self.keep_loss_history()
        # (the first calls of a plan are the eager run and the capture of the single step)
        while k > 0 and (self._graphs is None or not self.use_graph or len(self._segs) != 1
                         or self._segs[0][1] is not None or self._segs_key != self._dp_key()):
            self.run()
            k -= 1
            if not self.use_graph or (self._segs is not None and
                                      (len(self._segs) != 1 or self._segs[0][1] is not None)):
                for _ in range(k):
                    self.run()
                return
        if k == 0:
            return
        if k == 1:
            return self.run()
        self._ev_idx ^= 1
        ctx = self.ctx
        old = ctx.stream
        ctx.set_stream(self.stream)
        try:
            with torch.cuda.stream(self.stream):
                if self._upd_zeroes_g() and not self.model._g_clean:
                    ctx.fill(self.model.G, 0.0)
                self.model._g_clean = False
                if self._upd is not None and getattr(self.model, '_img_owner', None) is not self:
                    ctx.conv3d_pack_multi(*self._pack_dev)
                g = self._multi.get(k)
                if g is None:
                    emit = self._segs[0][0]
                    self._seg_idx = 0
                    self._capturing = True
                    ctx.graph_begin()
                    try:
                        for _ in range(k):
                            self._emit_seg(0, emit, self._zero_keep)
                    except BaseException:
                        self._capturing = False
                        try:
                            ctx.graph_destroy(ctx.graph_end())
                        except Exception:
                            pass
                        raise
                    g = ctx.graph_end()
                    self._capturing = False
                    self._multi[k] = g
                ctx.record(self._ev0)
                ctx.graph_launch(g)
                ctx.record(self._ev1)
                self._calls += k
                self._n_runs += k
                self.model._g_clean = self._upd_zeroes_g()
                if self.step in ('Adam', 'SGD'):
                    self.model._img_owner = self if self._upd is not None else None
        finally:
            ctx.set_stream(old)

    # ---- call ----------------------------------------------------------------------------------
    def set_inputs(self, args):
        if len(args) != len(self.inputs):
            raise TypeError("%s: %i inputs required, %i were given."
                            % (self.name, len(self.inputs), len(args)))
        batch = None
        for node, a in zip(self.inputs, args):
            decl = node.shape.shape
            if len(a.shape) != len(decl):
                raise TypeError("input '%s': rank %i given, %i required"
                                % (node.name, len(a.shape), len(decl)))
            for d, (s, g) in enumerate(zip(decl, a.shape)):
                if s is None:
                    if node.shape.tags[d] == 'b':
                        batch = g if batch is None else batch
                elif int(s) != int(g):
                    raise TypeError("input '%s': shape %s given, %s required"
                                    % (node.name, tuple(a.shape), tuple(decl)))
            if decl[node.shape.tag2index('b')] is not None:
                batch = decl[node.shape.tag2index('b')] if batch is None else batch
        if batch is None:
            batch = 1
        if not self._built or batch != self.batch:
            self.build(batch)
        # the plan's stream is non-blocking: device inputs were produced on the caller's
        # current stream (PatchSampler's warp kernels, a broadcast, a fill) -- order the
        # copies behind it, and keep the source alive until they have run
        self.stream.wait_stream(torch.cuda.current_stream(self.ctx.device))
        with torch.cuda.stream(self.stream):
            for node, a in zip(self.inputs, args):
                dst = self.out[node]
                if isinstance(a, torch.Tensor):
                    if a.data_ptr() == dst.data_ptr():
                        continue          # written in place (input_buffer): nothing to copy
                    if a.is_cuda:
                        a.record_stream(self.stream)
                    dst.copy_(a.to(torch.float32).reshape(dst.shape), non_blocking=True)
                else:
                    h = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
                    dst.copy_(h.reshape(dst.shape), non_blocking=False)

    def run(self):
        """launch the plan on the already-set inputs; returns nothing (async)."""
        self._ev_idx ^= 1
        self._n_runs += 1
        old = self.ctx.stream
        self.ctx.set_stream(self.stream)
        try:
            with torch.cuda.stream(self.stream):
                self._run_device()
        finally:
            self.ctx.set_stream(old)

    def fetch(self):
        self.stream.synchronize()
        self.last_device_time = self.ctx.elapsed_ms(self._ev0, self._ev1) * 1e-3
        rets = []
        with torch.cuda.stream(self.stream):
            for o in self.outputs:
                if self.training and o is self.loss_node:
                    nll = o.parent[0] if isinstance(o.parent, (list, tuple)) else o.parent
                    rets.append(np.float32(self.scratch[nll, 'loss'].item()))
                elif hasattr(o, 'host_value'):
                    rets.append(o.host_value(self))
                elif self.out.get(o) is None:
                    raise NotImplementedError("output of node %s is not materialised"
                                              % (o.name,))
                else:
                    rets.append(self.user_view(o, self.out[o]).detach().cpu().numpy().copy())
        if self.step == 'grad':
            return [g.detach().cpu().numpy().copy()
                    for g in self.model.device_grads_list()]
        return rets

    def fetch_async(self):
        """Deferred read-back of a TRAINING plan's loss: queues a copy of this step's loss
        into pinned host memory behind the step and returns ``(loss, device seconds)`` of
        the step submitted BEFORE it (None, None on the first call) -- the host never waits
        for the step it has just submitted, so the next one is queued while this one runs.
        The reference's trainingstep is synchronous (model.py:548-600: the loss is an output
        of the compiled function); this is the opt-in form of Model.trainingstep(sync=False)."""
        if not self.training:
            raise RuntimeError("fetch_async: a training plan is needed")
        nll = self.loss_node.parent[0] if isinstance(self.loss_node.parent, (list, tuple)) \
            else self.loss_node.parent
        dev_loss = self.scratch[nll, 'loss']
        if self._async is None:
            self._async = dict(pin=[torch.empty(1, dtype=torch.float32).pin_memory() for _ in range(2)],
                               ev=[torch.cuda.Event(), torch.cuda.Event()], n=0, run=[-1, -1])
        a = self._async
        slot = a['n'] & 1
        with torch.cuda.stream(self.stream):
            a['pin'][slot].copy_(dev_loss.reshape(1), non_blocking=True)
            a['ev'][slot].record(self.stream)
        a['run'][slot] = self._n_runs
        prev = None, None
        # the other slot belongs to the step before this one only if no synchronous call
        # (fetch) ran in between: the timing pair `_ev_idx ^ 1` and the pinned value are then
        # those of run n - 1.  Mixed sync / async use gets (None, None) for the step after a
        # synchronous one rather than an older step's loss paired with another step's time.
        if a['n'] > 0 and a['run'][slot ^ 1] == self._n_runs - 1:
            a['ev'][slot ^ 1].synchronize()          # the step before the one just queued
            e0, e1 = self._ev_pairs[self._ev_idx ^ 1]
            prev = np.float32(a['pin'][slot ^ 1].item()), self.ctx.elapsed_ms(e0, e1) * 1e-3
        a['n'] += 1
        return prev

    def __call__(self, *args):
        self.set_inputs(args)
        self.run()
        return self.fetch()

    # Input nodes own the static input buffers
    def input_buffer(self, node, batch=1):
        """the plan's static device buffer of an Input node.  A producer (PatchSampler,
        bench.py) may write it in place on the current stream and pass it to the call: the
        copy is then skipped (set_inputs orders the plan's stream behind the producer)."""
        if not self._built or (batch is not None and int(batch) != self.batch):
            self.build(batch)
        return self.out[node]


def _input_alloc(self, plan):
    if self not in plan.out:                  # (inputs of the plan are slices of its input arena)
        plan.out[self] = plan.empty(plan.out_shape(self))


from .node_basic import Input as _Input   # noqa: E402
_Input._plan_alloc = _input_alloc
