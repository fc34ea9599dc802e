"""ctypes bridge to ``libe2hip.so`` (the C ABI of ``include/e2hip.h``).

PyTorch-ROCm is used here for device memory and streams only: every tensor
handed to the library is a ``torch.float32`` CUDA(=HIP) tensor whose
``data_ptr()`` and element strides are passed through ``e2_tensor5``.

There is NO fallback: if the shared library is missing or does not load, the
import of this module raises, and so does every op.  (The CPU oracle under
``oracle/`` is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# E2HIP_LIB: explicit path of another build of the library (debug-switch / ablation builds
# made by `make BUILD=... OUT=...`; the product build is never overwritten by experiments)
LIB_PATH = os.environ.get("E2HIP_LIB") or os.path.join(_HERE, "libe2hip.so")

ACT = {"lin": 0, "relu": 1}


class E2Error(RuntimeError):
    pass


class Tensor5(C.Structure):
    _fields_ = [("ptr", C.c_void_p),
                ("n", C.c_int32), ("c", C.c_int32), ("d", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32),
                ("sn", C.c_int64), ("sc", C.c_int64), ("sd", C.c_int64),
                ("sh", C.c_int64)]


class Bf16Dst(C.Structure):
    """e2_bf16_dst (include/e2hip.h): where a producer kernel puts the bf16 copies of its output"""
    _fields_ = [("cl", C.c_void_p), ("cl_kg", C.c_int32), ("cl_d", C.c_int32), ("cl_h", C.c_int32),
                ("cl_w", C.c_int32), ("cl_oz", C.c_int32), ("cl_oy", C.c_int32), ("cl_ox", C.c_int32),
                ("pl", C.c_void_p), ("pl_plane", C.c_int64), ("pl_pitch", C.c_int32)]


def bf16_dst(cl=None, cl_dims=None, cl_off=(0, 0, 0), pl=None, pl_plane=0, pl_pitch=0):
    """cl: uint8 device tensor holding [n][d][kg][h][w][8] bf16 (cl_dims = (kg, d, h, w)); pl: uint8
    device tensor holding [n][c][d][pl_plane] bf16"""
    d = Bf16Dst()
    d.cl = cl.data_ptr() if cl is not None else None
    if cl is not None:
        d.cl_kg, d.cl_d, d.cl_h, d.cl_w = (int(v) for v in cl_dims)
        d.cl_oz, d.cl_oy, d.cl_ox = (int(v) for v in cl_off)
    d.pl = pl.data_ptr() if pl is not None else None
    d.pl_plane, d.pl_pitch = int(pl_plane), int(pl_pitch)
    return d


def t5_shape(shape):
    """a Tensor5 that carries extents only (null pointer): the shape argument of the entry points
    whose operand was made ahead of the call"""
    n, c, d, h, w = (int(v) for v in shape)
    return Tensor5(None, n, c, d, h, w, c * d * h * w, d * h * w, h * w, w)


def _load():
    if not os.path.exists(LIB_PATH):
        raise E2Error(
            "libe2hip.so not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C elektronn2_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    P5 = C.POINTER(Tensor5)
    vp, i, sz, fp = C.c_void_p, C.c_int, C.c_size_t, C.c_void_p
    sig = {
        "e2_ctx_create": (C.c_int, [i, C.POINTER(vp)]),
        "e2_ctx_destroy": (C.c_int, [vp]),
        "e2_ctx_set_stream": (C.c_int, [vp, vp]),
        "e2_last_error": (C.c_char_p, []),
        "e2_version": (C.c_int, []),
        "e2_conv3d_workspace_bytes": (sz, [i, i, i, i, i]),
        "e2_conv3d_fwd": (C.c_int, [vp, P5, fp, i, i, i, i, P5, vp, sz]),
        "e2_conv3d_dgrad": (C.c_int, [vp, P5, fp, i, i, i, i, P5, vp, sz]),
        "e2_conv3d_pack": (C.c_int, [vp, fp, i, i, i, i, i, i, vp, sz]),
        "e2_conv3d_fwd_packed": (C.c_int, [vp, P5, vp, i, i, i, i, P5]),
        "e2_conv3d_dgrad_packed": (C.c_int, [vp, P5, vp, i, i, i, i, P5]),
        "e2_conv3d_fwd_packed_act": (C.c_int, [vp, P5, vp, i, i, i, i, fp, i, P5]),
        "e2_bias_act_bwd_out": (C.c_int, [vp, P5, P5, i, P5, fp]),
        "e2_conv3d_dgrad_packed_actbwd": (C.c_int, [vp, P5, vp, i, i, i, i, P5, i, fp, P5, i, i, i, fp]),
        "e2_conv3d_wgrad": (C.c_int, [vp, P5, P5, fp, i, i, i]),
        "e2_conv3d_wgrad_acc": (C.c_int, [vp, P5, P5, fp, i, i, i]),
        "e2_conv3d_wgrad_pad": (C.c_int, [vp, P5, P5, fp, i, i, i, i]),
        "e2_pack_job_bytes": (sz, []),
        "e2_pack_job_fill": (C.c_int, [vp, fp, vp, i, i, i, i, i, i]),
        "e2_pack_job_set_rows": (C.c_int, [vp, i]),
        "e2_pack_job_set_stride": (C.c_int, [vp, i]),
        "e2_set_image_rows": (C.c_int, [vp, i]),
        "e2_conv3d_pack_multi": (C.c_int, [vp, vp, i]),
        "e2_conv3d_pack_multi_ex": (C.c_int, [vp, vp, i, i]),
        "e2_head_supported": (C.c_int, [i, i]),
        "e2_head_fwd": (C.c_int, [vp, P5, fp, fp, i, P5, P5, fp]),
        "e2_head_bwd_workspace_bytes": (C.c_size_t, [i, i, i, i, i, i]),
        "e2_tail_supported": (C.c_int, [i, i, i]),
        "e2_tail_workspace_bytes": (C.c_size_t, [i, i, i, i, i, i, i]),
        "e2_tail_fwd_bwd": (C.c_int, [vp, P5, fp, fp, fp, i, fp, fp, i, P5, P5, P5, P5, i, P5, fp, fp,
                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
        "e2_tail_reduce": (C.c_int, [vp, C.c_void_p, i, i, i, fp, fp, fp, fp, fp, i, fp]),
        "e2_head_bwd": (C.c_int, [vp, P5, fp, P5, P5, fp, P5, i, fp, fp, fp, C.c_void_p,
                                  C.c_size_t]),
        "e2_malis_loss_weights": (C.c_int, [i, C.c_void_p, i, C.c_void_p, C.c_void_p,
                                            C.c_void_p, i, C.c_void_p]),
        "e2_malis_connected_components": (C.c_int, [i, i, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    i, C.c_void_p]),
        "e2_warp_slice": (C.c_int, [vp, P5, C.POINTER(C.c_float), i, C.c_uint,
                                    C.POINTER(C.c_int), C.POINTER(C.c_float), P5]),
        "e2_grey_augment": (C.c_int, [vp, fp, C.c_size_t, C.c_float, C.c_float, C.c_float]),
        "e2_stream_fork": (C.c_int, [vp, C.c_void_p]),
        "e2_stream_join": (C.c_int, [vp, C.c_void_p]),
        "e2_conv1_supported": (C.c_int, [i, i, i, i, i, i, i]),
        "e2_conv1_pool_act_fwd": (C.c_int, [vp, P5, fp, fp, i, i, i, i, i, i, P5]),
        "e2_conv1_bwd_workspace_bytes": (C.c_size_t, [i, i, i, i, i, i, i]),
        "e2_conv1_pool_act_bwd": (C.c_int, [vp, P5, fp, fp, P5, i, i, i, i, i, fp, fp,
                                            C.c_void_p, C.c_size_t]),
        "e2_pool_bias_act_fwd": (C.c_int, [vp, P5, fp, i, i, i, i, P5]),
        "e2_conv3d_fwd_packed_parts": (C.c_int, [vp, P5, vp, i, i, i, i, P5, C.c_int64, i,
                                                 C.POINTER(C.c_int)]),
        "e2_conv3d_dgrad_packed_parts": (C.c_int, [vp, P5, vp, i, i, i, i, P5, C.c_int64, i,
                                                   C.POINTER(C.c_int)]),
        "e2_pool_bias_act_fwd_parts": (C.c_int, [vp, P5, C.c_int64, i, fp, i, i, i, i, P5]),
        "e2_pool_bias_act_bwd_parts": (C.c_int, [vp, P5, C.c_int64, i, P5, fp, i, i, i, i, P5, fp]),
        "e2_bias_act_bwd_out_parts": (C.c_int, [vp, P5, C.c_int64, i, P5, i, P5, fp]),
        "e2_pool_bias_act_bwd": (C.c_int, [vp, P5, P5, fp, i, i, i, i, P5, fp]),
        "e2_maxpool3d_fwd": (C.c_int, [vp, P5, i, i, i, P5]),
        "e2_maxpool3d_bwd": (C.c_int, [vp, P5, P5, i, i, i, P5, i]),
        "e2_upconv3d_workspace_bytes": (sz, [i, i, i, i, i, i, i, i, i]),
        "e2_upconv3d_fwd": (C.c_int, [vp, P5, fp, fp, i, i, i, i, i, P5, vp, sz]),
        "e2_upconv3d_bwd": (C.c_int, [vp, P5, fp, P5, P5, i, i, i, i, P5, fp, fp, vp, sz]),
        "e2_upconv3d_image_bytes": (sz, [i, i, i, i, i]),
        "e2_upconv3d_fwd_packed": (C.c_int, [vp, P5, fp, fp, i, i, i, i, i, P5]),
        "e2_upconv3d_bwd_packed": (C.c_int, [vp, P5, fp, P5, P5, i, i, i, i, P5, fp, fp, vp, sz, i]),
        "e2_transpose_ncdhw_to_ndhwc": (C.c_int, [vp, P5, fp]),
        "e2_transpose_ndhwc_to_ncdhw": (C.c_int, [vp, fp, P5]),
        "e2_copy5": (C.c_int, [vp, P5, P5, i]),
        "e2_fill": (C.c_int, [vp, fp, sz, C.c_float]),
        "e2_softmax_nll_fwd": (C.c_int, [vp, P5, P5, P5, fp]),
        "e2_softmax_nll_bwd": (C.c_int, [vp, P5, P5, fp, P5, fp]),
        "e2_malis_nll": (C.c_int, [vp, P5, fp, fp, fp, P5, fp]),
        "e2_fill_multi": (C.c_int, [vp, vp, vp, C.c_int, C.c_float]),
        "e2_set_skip_zero_fill": (C.c_int, [vp, C.c_int]),
        "e2_set_input_slack": (C.c_int, [vp, C.c_int]),
        "e2_conv_last_zero_fill": (C.c_int, [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
        "e2_set_mfma_dtype": (C.c_int, [vp, C.c_int]),
        "e2_set_tiling": (C.c_int, [vp, C.c_int, C.c_char_p]),
        "e2_last_launch": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "e2_tiling_fallbacks": (C.c_uint, [vp]),
        "e2_conv3d_bf16_workspace_bytes": (sz, [i, i, i, i, i, i, i, i, i]),
        "e2_conv3d_fwd_bf16": (C.c_int, [vp, P5, fp, i, i, i, i, fp, i, P5, vp, sz]),
        "e2_conv3d_dgrad_bf16": (C.c_int, [vp, P5, fp, i, i, i, i, P5, vp, sz]),
        "e2_conv3d_wgrad_bf16_workspace_bytes": (sz, [i, i, i, i, i, i, i, i, i]),
        "e2_conv3d_wgrad_bf16": (C.c_int, [vp, P5, P5, fp, i, i, i, i, vp, sz]),
        "e2_conv3d_bf16_xkeep_bytes": (sz, [i, i, i, i, i, i, i]),
        "e2_conv3d_fwd_bf16_keep": (C.c_int, [vp, P5, fp, i, i, i, i, fp, i, P5, vp, sz, vp, sz]),
        "e2_conv3d_wgrad_bf16_xcl": (C.c_int, [vp, P5, vp, i, P5, fp, i, i, i, i, vp, sz]),
        "e2_pool_bias_act_bwd_bf16": (C.c_int, [vp, P5, C.c_int64, i, P5, fp, i, i, i, i, P5, fp, C.POINTER(Bf16Dst)]),
        "e2_pool_bias_act_fwd_bf16": (C.c_int, [vp, P5, C.c_int64, i, fp, i, i, i, i, P5, C.POINTER(Bf16Dst)]),
        "e2_conv1_pool_act_fwd_bf16": (C.c_int, [vp, P5, fp, fp, i, i, i, i, i, i, P5, vp, i]),
        "e2_conv3d_bf16_wb_bytes": (sz, [i, i, i, i, i, i, i, i, i]),
        "e2_bf16_wjob_bytes": (sz, []),
        "e2_bf16_wjob_fill": (C.c_int, [vp, fp, i, i, i, i, i, i, i, i, i, i, vp, sz]),
        "e2_conv3d_bf16_pack_w_multi": (C.c_int, [vp, vp, i]),
        "e2_conv3d_fwd_bf16_ex": (C.c_int, [vp, P5, fp, i, i, i, i, fp, i, P5, vp, sz, vp, sz, i, vp, vp, i]),
        "e2_conv3d_dgrad_bf16_ex": (C.c_int, [vp, P5, fp, i, i, i, i, P5, vp, sz, vp, vp]),
        "e2_conv3d_wgrad_bf16_geometry": (C.c_int, [i, i, i, i, i, i, i, i, i, C.POINTER(C.c_int64),
                                                    C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "e2_conv3d_wgrad_bf16_ex": (C.c_int, [vp, P5, vp, i, P5, vp, fp, fp, i, i, i, i, vp, sz]),
        "e2_dense_fwd": (C.c_int, [vp, fp, fp, fp, i, i, i]),
        "e2_dense_dgrad": (C.c_int, [vp, fp, fp, fp, i, i, i, i]),
        "e2_dense_wgrad": (C.c_int, [vp, fp, fp, fp, i, i, i, i]),
        "e2_batchnorm_act_fwd": (C.c_int, [vp, P5, fp, fp, fp, fp, i, i, i, P5, fp]),
        "e2_batchnorm_act_bwd": (C.c_int, [vp, P5, P5, fp, fp, fp, i, i, P5, fp, fp]),
        "e2_get_mfma_dtype": (C.c_int, [vp]),
        "e2_adam_step": (C.c_int, [vp, fp, fp, fp, fp, sz, vp, fp, i, fp]),
        "e2_sgd_step": (C.c_int, [vp, fp, fp, fp, sz, vp, fp, i, fp]),
        "e2_set_loss_grad_mode": (C.c_int, [vp, i, fp]),
        "e2_upd_job_bytes": (sz, []),
        "e2_upd_rest_bytes": (sz, []),
        "e2_upd_job_fill": (C.c_int, [vp, C.c_long, vp, vp, i, i, i, i, i, C.c_float, i,
                                      C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
        "e2_upd_rest_fill": (C.c_int, [vp, C.c_long, C.c_long, C.c_float]),
        "e2_adam_pack_step": (C.c_int, [vp, fp, fp, fp, fp, vp, i, i, vp, i, fp, fp, C.c_float, i, sz]),
        "e2_adam_step_ex": (C.c_int, [vp, fp, fp, fp, fp, sz, vp, fp, i, fp, fp, C.c_float, i]),
        "e2_sgd_step_ex": (C.c_int, [vp, fp, fp, fp, sz, vp, fp, i, fp, fp, C.c_float, i]),
        "e2_step_prologue": (C.c_int, [vp, fp, i, C.c_size_t, fp, fp, i, fp, i, vp]),
        "e2_graph_begin": (C.c_int, [vp]),
        "e2_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
        "e2_graph_launch": (C.c_int, [vp, vp]),
        "e2_graph_destroy": (C.c_int, [vp]),
        "e2_graph_debug_dot": (C.c_int, [vp, C.c_char_p, i]),
        "e2_event_create": (C.c_int, [C.POINTER(vp)]),
        "e2_event_record": (C.c_int, [vp, vp]),
        "e2_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
        "e2_event_destroy": (C.c_int, [vp]),
        "e2_stream_synchronize": (C.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


_lib, EXPORTED_SYMBOLS = _load()


def lib():
    return _lib


def _chk(rc: int, what: str):
    if rc != 0:
        msg = _lib.e2_last_error()
        raise E2Error("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def t5(t: torch.Tensor) -> Tensor5:
    """Describe a 5-D float32 device tensor (any view with unit W stride)."""
    if t.dim() != 5:
        raise TypeError("expected a 5-D (b,f,z,x,y) tensor, got %s" % (tuple(t.shape),))
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    if not t.is_cuda:
        raise E2Error("tensor is on %s; the HIP path needs a GPU tensor (no CPU fallback)"
                      % t.device)
    s = t.stride()
    if t.shape[4] > 1 and s[4] != 1:
        raise TypeError("innermost stride must be 1, got %s" % (s,))
    return Tensor5(t.data_ptr(), t.shape[0], t.shape[1], t.shape[2], t.shape[3],
                   t.shape[4], s[0], s[1], s[2], s[3])


def _fp(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise TypeError("expected a contiguous float32 GPU tensor")
    return C.c_void_p(t.data_ptr())


class Context:
    """One HIP stream + the library handle (one per process / device)."""

    def __init__(self, device: int = 0, stream: Optional[torch.cuda.Stream] = None):
        if not torch.cuda.is_available():
            raise E2Error("no GPU visible: the elektronn2_amd hot path is HIP-only")
        torch.cuda.set_device(device)
        h = C.c_void_p()
        _chk(_lib.e2_ctx_create(device, C.byref(h)), "e2_ctx_create")
        self.h = h
        self.device = torch.device("cuda", device)
        self.stream = None
        self.set_stream(stream if stream is not None else torch.cuda.current_stream(device))
        self._ws = {}

    def set_stream(self, stream: torch.cuda.Stream):
        self.stream = stream
        _chk(_lib.e2_ctx_set_stream(self.h, C.c_void_p(stream.cuda_stream)),
             "e2_ctx_set_stream")

    def stream_fork(self, side: torch.cuda.Stream):
        _chk(_lib.e2_stream_fork(self.h, C.c_void_p(side.cuda_stream)), "e2_stream_fork")

    def stream_join(self, side: torch.cuda.Stream):
        _chk(_lib.e2_stream_join(self.h, C.c_void_p(side.cuda_stream)), "e2_stream_join")

    def synchronize(self):
        _chk(_lib.e2_stream_synchronize(self.h), "e2_stream_synchronize")

    # ---- workspace -----------------------------------------------------
    def workspace(self, key, nbytes: int) -> torch.Tensor:
        w = self._ws.get(key)
        if w is None or w.numel() * 4 < nbytes:
            w = torch.empty((nbytes + 3) // 4 + 64, dtype=torch.float32, device=self.device)
            self._ws[key] = w
        return w

    # ---- conv ------------------------------------------------------------
    def conv_ws_bytes(self, cout, cin, k):
        return int(_lib.e2_conv3d_workspace_bytes(cout, cin, k[0], k[1], k[2]))

    def conv3d_fwd(self, x, w, y, ws=None):
        cout, cin, kd, kh, kw = w.shape
        nb = self.conv_ws_bytes(cout, cin, (kd, kh, kw))
        ws = ws if ws is not None else self.workspace("conv", nb)
        _chk(_lib.e2_conv3d_fwd(self.h, C.byref(t5(x)), _fp(w), cout, kd, kh, kw,
                                C.byref(t5(y)), C.c_void_p(ws.data_ptr()), ws.numel() * 4),
             "e2_conv3d_fwd")

    def conv3d_dgrad(self, dy_pad, w, dx, ws=None):
        cout, cin, kd, kh, kw = w.shape
        nb = self.conv_ws_bytes(cout, cin, (kd, kh, kw))
        ws = ws if ws is not None else self.workspace("conv", nb)
        _chk(_lib.e2_conv3d_dgrad(self.h, C.byref(t5(dy_pad)), _fp(w), cin, kd, kh, kw,
                                  C.byref(t5(dx)), C.c_void_p(ws.data_ptr()),
                                  ws.numel() * 4), "e2_conv3d_dgrad")

    def conv3d_pack(self, w, mode, ws):
        cout, cin, kd, kh, kw = w.shape
        _chk(_lib.e2_conv3d_pack(self.h, _fp(w), cout, cin, kd, kh, kw, mode,
                                 C.c_void_p(ws.data_ptr()), ws.numel() * 4),
             "e2_conv3d_pack")

    def conv3d_fwd_packed(self, x, wp, cout, k, y):
        _chk(_lib.e2_conv3d_fwd_packed(self.h, C.byref(t5(x)), C.c_void_p(wp.data_ptr()),
                                       cout, k[0], k[1], k[2], C.byref(t5(y))),
             "e2_conv3d_fwd_packed")

    def conv3d_dgrad_packed(self, dy_pad, wp, cin, k, dx):
        _chk(_lib.e2_conv3d_dgrad_packed(self.h, C.byref(t5(dy_pad)),
                                         C.c_void_p(wp.data_ptr()), cin, k[0], k[1], k[2],
                                         C.byref(t5(dx))), "e2_conv3d_dgrad_packed")

    # ---- split-K with partial-sum stores (e2hip.h "split-K without atomics") ----------------
    def conv3d_fwd_packed_parts(self, x, wp, cout, k, y_parts):
        """y_parts: (P, n, cout, d, h, w) dense slabs; returns the number of parts written
        (1: y_parts[0] is the complete result)"""
        n = C.c_int(0)
        _chk(_lib.e2_conv3d_fwd_packed_parts(self.h, C.byref(t5(x)), C.c_void_p(wp.data_ptr()),
                                             cout, k[0], k[1], k[2], C.byref(t5(y_parts[0])),
                                             y_parts.stride(0), y_parts.shape[0], C.byref(n)),
             "e2_conv3d_fwd_packed_parts")
        return n.value

    def conv3d_dgrad_packed_parts(self, dy_pad, wp, cin, k, dx_parts):
        n = C.c_int(0)
        _chk(_lib.e2_conv3d_dgrad_packed_parts(self.h, C.byref(t5(dy_pad)),
                                               C.c_void_p(wp.data_ptr()), cin, k[0], k[1], k[2],
                                               C.byref(t5(dx_parts[0])), dx_parts.stride(0),
                                               dx_parts.shape[0], C.byref(n)),
             "e2_conv3d_dgrad_packed_parts")
        return n.value

    def pool_bias_act_fwd_parts(self, y_parts, nparts, bias, pool, act, out):
        _chk(_lib.e2_pool_bias_act_fwd_parts(self.h, C.byref(t5(y_parts[0])), y_parts.stride(0),
                                             nparts, _fp(bias), pool[0], pool[1], pool[2],
                                             ACT[act], C.byref(t5(out))),
             "e2_pool_bias_act_fwd_parts")

    def pool_bias_act_bwd_parts(self, dout_parts, nparts, y, bias, pool, act, dy, dbias):
        _chk(_lib.e2_pool_bias_act_bwd_parts(self.h, C.byref(t5(dout_parts[0])),
                                             dout_parts.stride(0), nparts, C.byref(t5(y)),
                                             _fp(bias), pool[0], pool[1], pool[2], ACT[act],
                                             C.byref(t5(dy)), _fp(dbias)),
             "e2_pool_bias_act_bwd_parts")

    def bias_act_bwd_out_parts(self, dout_parts, nparts, out, act, dy, dbias):
        _chk(_lib.e2_bias_act_bwd_out_parts(self.h, C.byref(t5(dout_parts[0])),
                                            dout_parts.stride(0), nparts, C.byref(t5(out)),
                                            ACT[act], C.byref(t5(dy)), _fp(dbias)),
             "e2_bias_act_bwd_out_parts")

    def conv3d_fwd_packed_act(self, x, wp, cout, k, bias, act, out):
        """conv + bias + activation in one launch (layers that do not pool)"""
        _chk(_lib.e2_conv3d_fwd_packed_act(self.h, C.byref(t5(x)), C.c_void_p(wp.data_ptr()),
                                           cout, k[0], k[1], k[2], _fp(bias), ACT[act],
                                           C.byref(t5(out))), "e2_conv3d_fwd_packed_act")

    def conv3d_dgrad_packed_actbwd(self, dy_pad, wp, cin, k, out_prev, act_prev, dy_pad_prev,
                                   pad_prev, dbias_prev, bias_prev=None):
        """data gradient + activation backward of the producing layer, written into the
        interior of that layer's zero-padded gradient buffer (e2hip.h)"""
        _chk(_lib.e2_conv3d_dgrad_packed_actbwd(
            self.h, C.byref(t5(dy_pad)), C.c_void_p(wp.data_ptr()), int(cin), int(k[0]),
            int(k[1]), int(k[2]), C.byref(t5(out_prev)), ACT[act_prev],
            _fp(bias_prev) if bias_prev is not None else None, C.byref(t5(dy_pad_prev)),
            int(pad_prev[0]), int(pad_prev[1]), int(pad_prev[2]),
            _fp(dbias_prev) if dbias_prev is not None else None),
            "e2_conv3d_dgrad_packed_actbwd")

    def bias_act_bwd_out(self, dout, out, act, dy, dbias):
        _chk(_lib.e2_bias_act_bwd_out(self.h, C.byref(t5(dout)), C.byref(t5(out)), ACT[act],
                                      C.byref(t5(dy)), _fp(dbias)), "e2_bias_act_bwd_out")

    # ---- patch extraction / augmentation ---------------------------------------
    def warp_slice(self, src, minv, perspective, nearest_mask, dest_off, src_off, dst):
        """src (F,Z,X,Y) view, dst (F,pz,px,py) dense device tensors; minv 4x4 host array"""
        m = (C.c_float * 16)(*[float(v) for v in np.asarray(minv, np.float32).ravel()])
        do = (C.c_int * 3)(*[int(v) for v in dest_off])
        so = (C.c_float * 3)(*[float(v) for v in src_off])
        _chk(_lib.e2_warp_slice(self.h, C.byref(t5(src[None])), m, 1 if perspective else 0,
                                int(nearest_mask), do, so, C.byref(t5(dst[None]))),
             "e2_warp_slice")

    def grey_augment(self, chan, alpha, c, gamma):
        """in place on one dense channel tensor"""
        if not chan.is_contiguous():
            raise TypeError("grey_augment needs a dense channel")
        _chk(_lib.e2_grey_augment(self.h, _fp(chan), chan.numel(), float(alpha), float(c),
                                  float(gamma)), "e2_grey_augment")

    # ---- fused classifier head -------------------------------------------------
    @staticmethod
    def head_supported(cin, ncls):
        return bool(_lib.e2_head_supported(int(cin), int(ncls)))

    def head_fwd(self, x, w, bias, target, probs, stats):
        _chk(_lib.e2_head_fwd(self.h, C.byref(t5(x)), _fp(w), _fp(bias), probs.shape[1],
                              C.byref(t5(target)) if target is not None else None,
                              C.byref(t5(probs)), _fp(stats) if stats is not None else None),
             "e2_head_fwd")

    @staticmethod
    def head_bwd_ws_bytes(x_shape, ncls):
        n, cin, d, h, w = (int(v) for v in x_shape)
        return int(_lib.e2_head_bwd_workspace_bytes(n, cin, int(ncls), d, h, w))

    def head_bwd(self, x, w, probs, target, stats, dx, accumulate_dx, dw, dbias, loss_out,
                 ws=None):
        if ws is None:
            ws = torch.empty(self.head_bwd_ws_bytes(x.shape, probs.shape[1]) // 4 + 16,
                             dtype=torch.float32, device=self.device)
        _chk(_lib.e2_head_bwd(self.h, C.byref(t5(x)), _fp(w), C.byref(t5(probs)),
                              C.byref(t5(target)), _fp(stats),
                              C.byref(t5(dx)) if dx is not None else None,
                              1 if accumulate_dx else 0, _fp(dw), _fp(dbias),
                              _fp(loss_out) if loss_out is not None else None,
                              C.c_void_p(ws.data_ptr()), ws.numel() * 4), "e2_head_bwd")

    # ---- the tail of the neuro3d nets: 1x1x1 conv + head, forward and backward --------
    @staticmethod
    def tail_supported(c1, c2, ncls):
        return bool(_lib.e2_tail_supported(int(c1), int(c2), int(ncls)))

    @staticmethod
    def tail_ws_bytes(x_shape, c2, ncls):
        n, c1, d, h, w = (int(v) for v in x_shape)
        return int(_lib.e2_tail_workspace_bytes(n, c1, int(c2), int(ncls), d, h, w))

    def tail_fwd_bwd(self, x, wp_fwd, wp_dgrad, bias1, w_head, b_head, target, probs, dpre, dx,
                     stats, ws, gm_mode=0, gm_src=None, gm_bias=None):
        """csrc/tail.hip: forward and backward of [1x1x1 conv + bias + relu] -> [classifier
        head] in one launch; returns the number of partial-sum slots written to ``ws``.
        ``gm_mode``: dx goes through the activation backward of the layer that produced x
        (1: slope from ``gm_src`` = its activated output, 2: from ``gm_src`` = its
        pre-activation + ``gm_bias``, 3: linear) and that layer's bias gradient joins the slots"""
        n_slots = C.c_int(0)
        _chk(_lib.e2_tail_fwd_bwd(self.h, C.byref(t5(x)), _fp(wp_fwd), _fp(wp_dgrad), _fp(bias1),
                                  dpre.shape[1], _fp(w_head), _fp(b_head), probs.shape[1],
                                  C.byref(t5(target)), C.byref(t5(probs)), C.byref(t5(dpre)),
                                  C.byref(t5(dx)) if dx is not None else None, int(gm_mode),
                                  C.byref(t5(gm_src)) if gm_src is not None else None,
                                  _fp(gm_bias), _fp(stats),
                                  C.c_void_p(ws.data_ptr()), ws.numel() * 4, C.byref(n_slots)),
             "e2_tail_fwd_bwd")
        return int(n_slots.value)

    def tail_reduce(self, ws, n_slots, c2, ncls, dw_head, db_head, db1, stats, loss_out,
                    db_parent=None):
        _chk(_lib.e2_tail_reduce(self.h, C.c_void_p(ws.data_ptr()), int(n_slots), int(c2),
                                 int(ncls), _fp(dw_head), _fp(db_head), _fp(db1), _fp(stats),
                                 _fp(loss_out) if loss_out is not None else None,
                                 0 if db_parent is None else int(db_parent.numel()),
                                 _fp(db_parent)), "e2_tail_reduce")

    def conv3d_wgrad(self, x, dy, dw, accumulate=False):
        kd, kh, kw = dw.shape[2:]
        fn = _lib.e2_conv3d_wgrad_acc if accumulate else _lib.e2_conv3d_wgrad
        _chk(fn(self.h, C.byref(t5(x)), C.byref(t5(dy)), _fp(dw), kd, kh, kw),
             "e2_conv3d_wgrad")

    def conv3d_wgrad_pad(self, x, dy_pad, dw, accumulate=False):
        """wgrad from the zero-padded gradient buffer (borders zero, >= 128 readable
        bytes after its last element): the direct kernel."""
        kd, kh, kw = dw.shape[2:]
        _chk(_lib.e2_conv3d_wgrad_pad(self.h, C.byref(t5(x)), C.byref(t5(dy_pad)), _fp(dw),
                                      kd, kh, kw, 1 if accumulate else 0),
             "e2_conv3d_wgrad_pad")

    def set_image_rows(self, rows):
        """floats per k-row of the packed conv weight images the next conv3d_pack /
        conv3d_{fwd,dgrad}_packed* calls read (0 / None = the library's formula)"""
        _chk(_lib.e2_set_image_rows(self.h, int(rows or 0)), "e2_set_image_rows")

    def make_pack_jobs(self, jobs, rows=None, strides=None):
        """jobs: list of (w tensor, wp tensor, mode).  Returns a device byte tensor
        of job records for conv3d_pack_multi.  ``rows`` (optional, one entry per job, None / 0 =
        default): how far the reading launch's M tiles reach (e2_pack_job_set_rows).  ``strides``
        (optional, likewise): row length of the image (e2_pack_job_set_stride; the launches that
        read it need set_image_rows with the same value)."""
        rec = int(_lib.e2_pack_job_bytes())
        buf = (C.c_char * (rec * len(jobs)))()
        for n, (w, wp, mode) in enumerate(jobs):
            cout, cin, kd, kh, kw = w.shape
            _chk(_lib.e2_pack_job_fill(C.byref(buf, n * rec), _fp(w), C.c_void_p(wp.data_ptr()),
                                       cout, cin, kd, kh, kw, mode), "e2_pack_job_fill")
            if strides is not None and strides[n]:
                _chk(_lib.e2_pack_job_set_stride(C.byref(buf, n * rec), int(strides[n])), "e2_pack_job_set_stride")
            if rows is not None and rows[n]:
                _chk(_lib.e2_pack_job_set_rows(C.byref(buf, n * rec), int(rows[n])), "e2_pack_job_set_rows")
        host = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
        dev = host.to(self.device)
        dev.max_taps = max(int(w.shape[2] * w.shape[3] * w.shape[4]) for w, _, _ in jobs)
        dev.keep = [(w, wp) for w, wp, _ in jobs]
        return dev, len(jobs)

    def conv3d_pack_multi(self, jobs_dev, njobs):
        _chk(_lib.e2_conv3d_pack_multi_ex(self.h, C.c_void_p(jobs_dev.data_ptr()), njobs,
                                          int(getattr(jobs_dev, 'max_taps', 248))),
             "e2_conv3d_pack_multi_ex")

    # ---- fused first layer ----------------------------------------------------
    @staticmethod
    def conv1_supported(cin, k, pool):
        return bool(_lib.e2_conv1_supported(cin, k[0], k[1], k[2], pool[0], pool[1], pool[2]))

    def conv1_pool_act_fwd(self, x, w, bias, pool, act, out):
        _chk(_lib.e2_conv1_pool_act_fwd(self.h, C.byref(t5(x)), _fp(w), _fp(bias), w.shape[0],
                                        w.shape[3], w.shape[4], pool[1], pool[2], ACT[act],
                                        C.byref(t5(out))), "e2_conv1_pool_act_fwd")

    def conv1_pool_act_fwd_bf16(self, x, w, bias, pool, act, out, next_xb, next_kg):
        """conv1_pool_act_fwd + the next conv layer's channels-last bf16 input image"""
        _chk(_lib.e2_conv1_pool_act_fwd_bf16(self.h, C.byref(t5(x)), _fp(w), _fp(bias), w.shape[0],
                                             w.shape[3], w.shape[4], pool[1], pool[2], ACT[act],
                                             C.byref(t5(out)), C.c_void_p(next_xb.data_ptr()),
                                             int(next_kg)), "e2_conv1_pool_act_fwd_bf16")

    @staticmethod
    def conv1_bwd_ws_bytes(dout_shape, k):
        n, cout, d, ho, wo = (int(v) for v in dout_shape)
        return int(_lib.e2_conv1_bwd_workspace_bytes(n, cout, d, ho, wo, k[1], k[2]))

    def conv1_pool_act_bwd(self, x, w, bias, dout, pool, act, dw, dbias, ws=None):
        if ws is None:
            ws = torch.empty(self.conv1_bwd_ws_bytes(dout.shape, w.shape[2:]) // 4 + 16,
                             dtype=torch.float32, device=self.device)
        _chk(_lib.e2_conv1_pool_act_bwd(self.h, C.byref(t5(x)), _fp(w), _fp(bias),
                                        C.byref(t5(dout)), w.shape[3], w.shape[4], pool[1],
                                        pool[2], ACT[act], _fp(dw), _fp(dbias),
                                        C.c_void_p(ws.data_ptr()), ws.numel() * 4),
             "e2_conv1_pool_act_bwd")

    # ---- pool / bias / act ---------------------------------------------------
    def pool_bias_act_fwd(self, y, bias, pool, act, out):
        _chk(_lib.e2_pool_bias_act_fwd(self.h, C.byref(t5(y)), _fp(bias), pool[0], pool[1],
                                       pool[2], ACT[act], C.byref(t5(out))),
             "e2_pool_bias_act_fwd")

    def pool_bias_act_bwd(self, dout, y, bias, pool, act, dy, dbias):
        _chk(_lib.e2_pool_bias_act_bwd(self.h, C.byref(t5(dout)), C.byref(t5(y)), _fp(bias),
                                       pool[0], pool[1], pool[2], ACT[act], C.byref(t5(dy)),
                                       _fp(dbias)), "e2_pool_bias_act_bwd")

    def maxpool3d_fwd(self, x, pool, out):
        _chk(_lib.e2_maxpool3d_fwd(self.h, C.byref(t5(x)), pool[0], pool[1], pool[2],
                                   C.byref(t5(out))), "e2_maxpool3d_fwd")

    def maxpool3d_bwd(self, dout, x, pool, dx, accumulate=False):
        _chk(_lib.e2_maxpool3d_bwd(self.h, C.byref(t5(dout)), C.byref(t5(x)), pool[0],
                                   pool[1], pool[2], C.byref(t5(dx)), int(accumulate)),
             "e2_maxpool3d_bwd")

    # ---- upconv ------------------------------------------------------------------
    def upconv_ws_bytes(self, cout, cin, pool, x_shape):
        return int(_lib.e2_upconv3d_workspace_bytes(cout, cin, pool[0], pool[1], pool[2],
                                                    x_shape[0], x_shape[2], x_shape[3],
                                                    x_shape[4]))

    def upconv3d_fwd(self, x, w, bias, pool, act, y, ws=None):
        cout, cin = w.shape[:2]
        nb = self.upconv_ws_bytes(cout, cin, pool, x.shape)
        ws = ws if ws is not None else self.workspace("upconv", nb)
        _chk(_lib.e2_upconv3d_fwd(self.h, C.byref(t5(x)), _fp(w), _fp(bias), cout, pool[0],
                                  pool[1], pool[2], ACT[act], C.byref(t5(y)),
                                  C.c_void_p(ws.data_ptr()), ws.numel() * 4),
             "e2_upconv3d_fwd")

    def upconv3d_bwd(self, x, w, y, dout, pool, act, dx, dw, dbias, ws=None):
        cout, cin = w.shape[:2]
        nb = self.upconv_ws_bytes(cout, cin, pool, x.shape)
        ws = ws if ws is not None else self.workspace("upconv", nb)
        _chk(_lib.e2_upconv3d_bwd(self.h, C.byref(t5(x)), _fp(w), C.byref(t5(y)),
                                  C.byref(t5(dout)), pool[0], pool[1], pool[2], ACT[act],
                                  C.byref(t5(dx)) if dx is not None else None,
                                  _fp(dw), _fp(dbias), C.c_void_p(ws.data_ptr()),
                                  ws.numel() * 4), "e2_upconv3d_bwd")

    @staticmethod
    def upconv_image_bytes(cout, cin, pool):
        return int(_lib.e2_upconv3d_image_bytes(int(cout), int(cin), pool[0], pool[1], pool[2]))

    def upconv3d_fwd_packed(self, x, wp_fwd, bias, cout, pool, act, y):
        """the forward with the image a mode-2 pack job keeps current (make_pack_jobs)"""
        _chk(_lib.e2_upconv3d_fwd_packed(self.h, C.byref(t5(x)), _fp(wp_fwd), _fp(bias), int(cout),
                                         pool[0], pool[1], pool[2], ACT[act], C.byref(t5(y))),
             "e2_upconv3d_fwd_packed")

    def upconv3d_bwd_packed(self, x, wp_dgrad, y, dout, pool, act, dx, dw, dbias, ws,
                            accumulate=False):
        """the backward with the data gradient's image of a mode-3 pack job; accumulate: dw
        and dbias are added to (the caller zeroed them)"""
        _chk(_lib.e2_upconv3d_bwd_packed(self.h, C.byref(t5(x)), _fp(wp_dgrad), C.byref(t5(y)),
                                         C.byref(t5(dout)), pool[0], pool[1], pool[2], ACT[act],
                                         C.byref(t5(dx)) if dx is not None else None,
                                         _fp(dw), _fp(dbias), C.c_void_p(ws.data_ptr()),
                                         ws.numel() * 4, int(bool(accumulate))),
             "e2_upconv3d_bwd_packed")

    # ---- layout ------------------------------------------------------------------
    def to_ndhwc(self, src, dst):
        _chk(_lib.e2_transpose_ncdhw_to_ndhwc(self.h, C.byref(t5(src)), _fp(dst)),
             "e2_transpose_ncdhw_to_ndhwc")

    def to_ncdhw(self, src, dst):
        _chk(_lib.e2_transpose_ndhwc_to_ncdhw(self.h, _fp(src), C.byref(t5(dst))),
             "e2_transpose_ndhwc_to_ncdhw")

    def copy5(self, src, dst, accumulate=False):
        _chk(_lib.e2_copy5(self.h, C.byref(t5(src)), C.byref(t5(dst)), int(accumulate)),
             "e2_copy5")

    def fill(self, t, value=0.0):
        if not t.is_contiguous():
            raise TypeError("fill needs a contiguous tensor")
        _chk(_lib.e2_fill(self.h, C.c_void_p(t.data_ptr()), t.numel(), float(value)),
             "e2_fill")

    # ---- loss / optimiser ----------------------------------------------------------
    def softmax_nll_fwd(self, logits, target, probs, stats):
        _chk(_lib.e2_softmax_nll_fwd(self.h, C.byref(t5(logits)), C.byref(t5(target)),
                                     C.byref(t5(probs)), _fp(stats)), "e2_softmax_nll_fwd")

    def softmax_nll_bwd(self, probs, target, stats, dlogits, loss_out):
        _chk(_lib.e2_softmax_nll_bwd(self.h, C.byref(t5(probs)), C.byref(t5(target)),
                                     _fp(stats), C.byref(t5(dlogits)), _fp(loss_out)),
             "e2_softmax_nll_bwd")

    def fill_multi(self, ptrs_dev, counts_dev, n, value=0.0):
        """one launch that fills n flat regions (int64 device tensors of pointers / counts)"""
        _chk(_lib.e2_fill_multi(self.h, C.c_void_p(ptrs_dev.data_ptr()),
                                C.c_void_p(counts_dev.data_ptr()), int(n), float(value)),
             "e2_fill_multi")

    def set_skip_zero_fill(self, on):
        _chk(_lib.e2_set_skip_zero_fill(self.h, 1 if on else 0), "e2_set_skip_zero_fill")

    def conv_last_zero_fill(self):
        """(pointer, count) of the flat region the last conv launch zero-filled, or (0, 0)"""
        p, n = C.c_void_p(), C.c_size_t()
        _chk(_lib.e2_conv_last_zero_fill(self.h, C.byref(p), C.byref(n)), "e2_conv_last_zero_fill")
        return (p.value or 0), int(n.value)

    # ---- bf16 operands in memory (32x32x16 MFMA kernel, csrc/conv_bf16.hip) ---------------
    @staticmethod
    def conv_bf16_ws_bytes(x_shape, cout, k):
        n, cin, d, h, w = (int(v) for v in x_shape)
        return int(_lib.e2_conv3d_bf16_workspace_bytes(n, cin, d, h, w, int(cout), k[0], k[1], k[2]))

    @staticmethod
    def conv_bf16_xkeep_bytes(x_shape, k):
        """bytes of the forward's kept bf16 copy of x (zero-fill it once)"""
        n, cin, d, h, w = (int(v) for v in x_shape)
        return int(_lib.e2_conv3d_bf16_xkeep_bytes(n, cin, d, h, w, k[1], k[2]))

    def conv3d_fwd_bf16(self, x, w, y, bias=None, act='lin', ws=None, xkeep=None):
        """xkeep: a zero-initialised uint8 buffer of conv_bf16_xkeep_bytes that receives the
        channels-last bf16 copy of x (read again by conv3d_wgrad_bf16(xcl=...))"""
        cout, cin, kd, kh, kw = w.shape
        if ws is None:
            ws = self.workspace("conv_bf16", self.conv_bf16_ws_bytes(x.shape, cout, (kd, kh, kw)))
        if xkeep is not None:
            _chk(_lib.e2_conv3d_fwd_bf16_keep(self.h, C.byref(t5(x)), _fp(w), cout, kd, kh, kw,
                                              _fp(bias), ACT[act], C.byref(t5(y)),
                                              C.c_void_p(ws.data_ptr()), ws.numel() * ws.element_size(),
                                              C.c_void_p(xkeep.data_ptr()),
                                              xkeep.numel() * xkeep.element_size()),
                 "e2_conv3d_fwd_bf16_keep")
            return
        _chk(_lib.e2_conv3d_fwd_bf16(self.h, C.byref(t5(x)), _fp(w), cout, kd, kh, kw,
                                     _fp(bias), ACT[act], C.byref(t5(y)),
                                     C.c_void_p(ws.data_ptr()), ws.numel() * ws.element_size()),
             "e2_conv3d_fwd_bf16")

    def conv3d_dgrad_bf16(self, dy_pad, w, dx, ws=None):
        cout, cin, kd, kh, kw = w.shape
        if ws is None:
            ws = self.workspace("conv_bf16", self.conv_bf16_ws_bytes(dx.shape, cout, (kd, kh, kw)))
        _chk(_lib.e2_conv3d_dgrad_bf16(self.h, C.byref(t5(dy_pad)), _fp(w), cin, kd, kh, kw,
                                       C.byref(t5(dx)), C.c_void_p(ws.data_ptr()),
                                       ws.numel() * ws.element_size()), "e2_conv3d_dgrad_bf16")

    @staticmethod
    def wgrad_bf16_ws_bytes(x_shape, cout, k):
        n, cin, d, h, w = (int(v) for v in x_shape)
        return int(_lib.e2_conv3d_wgrad_bf16_workspace_bytes(n, cin, d, h, w, int(cout), k[0], k[1], k[2]))

    def conv3d_wgrad_bf16(self, x, dy, dw, accumulate=False, ws=None, xcl=None):
        """dw (n_f, n_in, kd, kh, kw) from x and the UNPADDED gradient view dy; xcl: the
        forward's kept bf16 copy of x (conv3d_fwd_bf16(xkeep=...)) -- x is not converted again"""
        cout, cin, kd, kh, kw = dw.shape
        if ws is None:
            ws = self.workspace("wgrad_bf16", self.wgrad_bf16_ws_bytes(x.shape, cout, (kd, kh, kw)))
        if xcl is not None:
            _chk(_lib.e2_conv3d_wgrad_bf16_xcl(self.h, C.byref(t5(x)), C.c_void_p(xcl.data_ptr()),
                                               (cin + 15) // 16 * 2, C.byref(t5(dy)), _fp(dw),
                                               kd, kh, kw, int(accumulate), C.c_void_p(ws.data_ptr()),
                                               ws.numel() * ws.element_size()),
                 "e2_conv3d_wgrad_bf16_xcl")
            return
        _chk(_lib.e2_conv3d_wgrad_bf16(self.h, C.byref(t5(x)), C.byref(t5(dy)), _fp(dw), kd, kh, kw,
                                       int(accumulate), C.c_void_p(ws.data_ptr()),
                                       ws.numel() * ws.element_size()), "e2_conv3d_wgrad_bf16")

    def bf16_memory_wgrad(self):
        return getattr(self, '_tiling', {}).get('wgrad', '').startswith('32,')

    def set_input_slack(self, nbytes):
        """promise (or, 0, withdraw the promise) that the x of the conv launches that follow has
        nbytes readable finite bytes behind its last element (e2_set_input_slack)"""
        _chk(_lib.e2_set_input_slack(self.h, int(nbytes)), "e2_set_input_slack")

    def current_tiling(self, kind):
        """the tiling string in force for kind ('igemm' | 'wgrad'), '' when none is pinned"""
        return getattr(self, '_tiling', {}).get(kind, '')

    # ---- the producers' epilogues: operands made ahead of the GEMM launches ------------------
    @staticmethod
    def bf16_tile(tiling):
        """(MB, NB) of a "32,MB,NB[,...]" tiling string, else None"""
        v = (tiling or '').split(',')
        if len(v) >= 3 and v[0] == '32':
            return int(v[1]), int(v[2])
        return None

    @staticmethod
    def conv_bf16_wb_bytes(rows, kk, k, in_w, out_w, tile):
        return int(_lib.e2_conv3d_bf16_wb_bytes(int(rows), int(kk), k[0], k[1], k[2], int(in_w),
                                                int(out_w), tile[0], tile[1]))

    def make_bf16_wjobs(self, jobs):
        """jobs: list of (w5 tensor (nf, nin, kd, kh, kw), mode, in_w, out_w, (MB, NB), wb uint8
        tensor).  Returns (device records, count) for conv3d_bf16_pack_w_multi."""
        rec = int(_lib.e2_bf16_wjob_bytes())
        buf = (C.c_char * (rec * len(jobs)))()
        for n, (w, mode, in_w, out_w, tile, wb) in enumerate(jobs):
            nf, nin, kd, kh, kw = w.shape
            _chk(_lib.e2_bf16_wjob_fill(C.byref(buf, n * rec), _fp(w), nf, nin, kd, kh, kw, int(mode),
                                        int(in_w), int(out_w), tile[0], tile[1],
                                        C.c_void_p(wb.data_ptr()), wb.numel()), "e2_bf16_wjob_fill")
        host = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
        return host.to(self.device), len(jobs)

    def conv3d_bf16_pack_w_multi(self, jobs_dev, njobs):
        _chk(_lib.e2_conv3d_bf16_pack_w_multi(self.h, C.c_void_p(jobs_dev.data_ptr()), njobs),
             "e2_conv3d_bf16_pack_w_multi")

    def conv3d_fwd_bf16_ex(self, x, w, y, bias=None, act='lin', ws=None, xkeep=None, x_ready=False,
                           wb=None, next_xb=None, next_kg=0):
        """conv3d_fwd_bf16 with ready-made operands (include/e2hip.h): x_ready: xkeep holds this
        step's copy of x; wb: the packed filter rows; next_xb: receives the channels-last bf16
        copy of y for the next layer"""
        cout, cin, kd, kh, kw = w.shape
        if ws is None and not (x_ready and wb is not None):
            ws = self.workspace("conv_bf16", self.conv_bf16_ws_bytes(x.shape, cout, (kd, kh, kw)))
        _chk(_lib.e2_conv3d_fwd_bf16_ex(
            self.h, C.byref(t5(x)), _fp(w), cout, kd, kh, kw, _fp(bias), ACT[act], C.byref(t5(y)),
            C.c_void_p(ws.data_ptr()) if ws is not None else None,
            ws.numel() * ws.element_size() if ws is not None else 0,
            C.c_void_p(xkeep.data_ptr()) if xkeep is not None else None,
            xkeep.numel() * xkeep.element_size() if xkeep is not None else 0, int(bool(x_ready)),
            C.c_void_p(wb.data_ptr()) if wb is not None else None,
            C.c_void_p(next_xb.data_ptr()) if next_xb is not None else None, int(next_kg)),
            "e2_conv3d_fwd_bf16_ex")

    def conv3d_dgrad_bf16_ex(self, dy_pad, w, dx, ws=None, dy_cl=None, wb=None):
        cout, cin, kd, kh, kw = w.shape
        if ws is None and not (dy_cl is not None and wb is not None):
            ws = self.workspace("conv_bf16", self.conv_bf16_ws_bytes(dx.shape, cout, (kd, kh, kw)))
        _chk(_lib.e2_conv3d_dgrad_bf16_ex(
            self.h, C.byref(t5(dy_pad)), _fp(w), cin, kd, kh, kw, C.byref(t5(dx)),
            C.c_void_p(ws.data_ptr()) if ws is not None else None,
            ws.numel() * ws.element_size() if ws is not None else 0,
            C.c_void_p(dy_cl.data_ptr()) if dy_cl is not None else None,
            C.c_void_p(wb.data_ptr()) if wb is not None else None), "e2_conv3d_dgrad_bf16_ex")

    @staticmethod
    def wgrad_bf16_geometry(x_shape, cout, k):
        """(elements per dy plane, bytes of the dy planes, bytes of the f32 sums) of
        conv3d_wgrad_bf16_ex's ready-made operands"""
        n, cin, d, h, w = (int(v) for v in x_shape)
        pd, db, sb = C.c_int64(), C.c_size_t(), C.c_size_t()
        _chk(_lib.e2_conv3d_wgrad_bf16_geometry(n, cin, d, h, w, int(cout), k[0], k[1], k[2],
                                                C.byref(pd), C.byref(db), C.byref(sb)),
             "e2_conv3d_wgrad_bf16_geometry")
        return int(pd.value), int(db.value), int(sb.value)

    def conv3d_wgrad_bf16_ex(self, x, dy, dw, accumulate=False, ws=None, xcl=None, dyc=None, sums=None):
        """conv3d_wgrad_bf16 with ready-made operands: xcl (the forward's kept copy of x), dyc
        (the bf16 planes of dy written by its producer), sums (a zeroed f32 buffer of the
        caller's, left zero)"""
        cout, cin, kd, kh, kw = dw.shape
        if ws is None and not (xcl is not None and dyc is not None and sums is not None):
            ws = self.workspace("wgrad_bf16", self.wgrad_bf16_ws_bytes(x.shape, cout, (kd, kh, kw)))
        _chk(_lib.e2_conv3d_wgrad_bf16_ex(
            self.h, C.byref(t5(x)), C.c_void_p(xcl.data_ptr()) if xcl is not None else None,
            (cin + 15) // 16 * 2, C.byref(t5(dy)),
            C.c_void_p(dyc.data_ptr()) if dyc is not None else None, _fp(sums), _fp(dw),
            kd, kh, kw, int(accumulate),
            C.c_void_p(ws.data_ptr()) if ws is not None else None,
            ws.numel() * ws.element_size() if ws is not None else 0), "e2_conv3d_wgrad_bf16_ex")

    def pool_bias_act_bwd_bf16(self, dout, src, bias, pool, act, dy, dbias, dst, parts=1, part_stride=0):
        """pool_bias_act_bwd (bias given: src = conv output) / bias_act_bwd_out (bias None: src =
        activated output) writing dy (f32, or None) and the bf16 copies named by dst (bf16_dst)"""
        dyv = t5(dy) if dy is not None else Tensor5(None, 0, 0, 0, 0, 0, 0, 0, 0, 0)
        _chk(_lib.e2_pool_bias_act_bwd_bf16(self.h, C.byref(t5(dout)), int(part_stride), int(parts),
                                            C.byref(t5(src)), _fp(bias), pool[0], pool[1], pool[2],
                                            ACT[act], C.byref(dyv), _fp(dbias), C.byref(dst)),
             "e2_pool_bias_act_bwd_bf16")

    def pool_bias_act_fwd_bf16(self, y, bias, pool, act, out, dst, parts=1, part_stride=0):
        _chk(_lib.e2_pool_bias_act_fwd_bf16(self.h, C.byref(t5(y)), int(part_stride), int(parts),
                                            _fp(bias), pool[0], pool[1], pool[2], ACT[act],
                                            C.byref(t5(out)), C.byref(dst)),
             "e2_pool_bias_act_fwd_bf16")

    # ---- config 1 (mnist): Perceptron dot product, batch normalisation ------------------
    def dense_fwd(self, x, w, y):
        """y (n, m) = x (n, k) . w (k, m); contiguous 2-D views of device tensors"""
        n, k = x.shape
        _chk(_lib.e2_dense_fwd(self.h, _fp(x), _fp(w), _fp(y), n, k, w.shape[1]), "e2_dense_fwd")

    def dense_dgrad(self, dy, w, dx, accumulate=False):
        n, m = dy.shape
        _chk(_lib.e2_dense_dgrad(self.h, _fp(dy), _fp(w), _fp(dx), n, w.shape[0], m,
                                 int(accumulate)), "e2_dense_dgrad")

    def dense_wgrad(self, x, dy, dw, accumulate=False):
        n, k = x.shape
        _chk(_lib.e2_dense_wgrad(self.h, _fp(x), _fp(dy), _fp(dw), n, k, dy.shape[1],
                                 int(accumulate)), "e2_dense_wgrad")

    def batchnorm_act_fwd(self, x, gamma, bias, run_mean, run_std, train, update_running, act,
                          out, save=None):
        _chk(_lib.e2_batchnorm_act_fwd(self.h, C.byref(t5(x)), _fp(gamma), _fp(bias),
                                       _fp(run_mean), _fp(run_std), int(bool(train)),
                                       int(bool(update_running)), ACT[act], C.byref(t5(out)),
                                       _fp(save)), "e2_batchnorm_act_fwd")

    def batchnorm_act_bwd(self, dout, x, gamma, bias, save, train, act, dx, dgamma, dbias):
        _chk(_lib.e2_batchnorm_act_bwd(self.h, C.byref(t5(dout)), C.byref(t5(x)), _fp(gamma),
                                       _fp(bias), _fp(save), int(bool(train)), ACT[act],
                                       C.byref(t5(dx)) if dx is not None else None,
                                       _fp(dgamma), _fp(dbias)), "e2_batchnorm_act_bwd")

    def set_tiling(self, kind, cfg):
        """force the tiling of the following 'igemm' ("MT,NT,CC,SK": conv fwd / dgrad /
        UpConv) or 'wgrad' ("MT,NT,WK,BP,PS") launches of this context; None / '' returns
        the choice to the library's cost model (e2_set_tiling)"""
        code = {'igemm': 0, 'wgrad': 1}[kind]
        _chk(_lib.e2_set_tiling(self.h, code, (cfg or "").encode()), "e2_set_tiling")
        self._tiling = getattr(self, '_tiling', {})
        self._tiling[kind] = cfg or ""

    def last_launch(self):
        """(kernel family, tiling that ran, 'forced' | 'model' | 'fallback') of the last conv GEMM
        launch of this context (e2_last_launch), or None before the first one"""
        buf = C.create_string_buffer(160)
        _chk(_lib.e2_last_launch(self.h, buf, 160), "e2_last_launch")
        parts = buf.value.decode().split(" ")
        return tuple(parts) if len(parts) == 3 else None

    def tiling_fallbacks(self):
        """launches of this context whose forced tiling was not the one that ran"""
        return int(_lib.e2_tiling_fallbacks(self.h))

    def bf16_memory_form(self):
        """True while the forced igemm tiling selects the kernel with bf16 operands in memory
        ("32,MB,NB", csrc/conv_bf16.hip): the callers that hold the canonical weights then
        call conv3d_fwd_bf16 / conv3d_dgrad_bf16 instead of the packed entry points"""
        return getattr(self, '_tiling', {}).get('igemm', '').startswith('32,')

    def set_mfma_dtype(self, dtype):
        """'f32' (default) or 'bf16': operand rounding of the conv GEMMs (f32 sums)"""
        code = {'f32': 0, 'float32': 0, 'bf16': 1, 'bfloat16': 1}.get(dtype)
        if code is None:
            raise ValueError("mfma dtype must be 'f32' or 'bf16', got %r" % (dtype,))
        _chk(_lib.e2_set_mfma_dtype(self.h, code), "e2_set_mfma_dtype")
        self._mfma = 'bf16' if code else 'f32'

    @property
    def mfma_dtype(self):
        m = getattr(self, '_mfma', None)
        if m is None:
            m = self._mfma = 'bf16' if _lib.e2_get_mfma_dtype(self.h) == 1 else 'f32'
        return m

    def malis_nll(self, probs, pos, neg, norm, dlogits, loss_sum):
        """MALIS NLL of a (1, 2E, z, x, y) pair-softmax; dlogits None = loss only"""
        _chk(_lib.e2_malis_nll(self.h, C.byref(t5(probs)), _fp(pos), _fp(neg), _fp(norm),
                               C.byref(t5(dlogits)) if dlogits is not None else None,
                               _fp(loss_sum)), "e2_malis_nll")

    def set_loss_grad_mode(self, sum_mode, count_out=None):
        """sum_mode: the NLL backward launches that follow leave the gradient unnormalised and
        write this rank's labelled count to ``count_out`` (one-element device tensor)"""
        _chk(_lib.e2_set_loss_grad_mode(self.h, int(bool(sum_mode)), _fp(count_out)),
             "e2_set_loss_grad_mode")

    def adam_step(self, p, g, m, s, seg_off, seg_reg, hyper, gdiv=None, gmul=1.0, zero_g=False):
        """gdiv / gmul: the update sees g * gmul / (gdiv[0] + 1e-5) (gdiv: device scalar, or
        None); zero_g: g is left zero for the next backward pass"""
        _chk(_lib.e2_adam_step_ex(self.h, _fp(p), _fp(g), _fp(m), _fp(s), p.numel(),
                                  C.c_void_p(seg_off.data_ptr()), _fp(seg_reg),
                                  seg_reg.numel(), _fp(hyper), _fp(gdiv), float(gmul),
                                  int(bool(zero_g))), "e2_adam_step_ex")

    def make_upd_jobs(self, jobs, rests):
        """records of adam_pack_step.  jobs: (arena offset, forward image | None, data-gradient
        image | None, (cout, cin, kd, kh, kw), weight-decay multiplier) per conv weight tensor;
        rests: (arena offset, elements, multiplier) for every other trainable run.  Returns the
        handle adam_pack_step takes."""
        rj, rr = int(_lib.e2_upd_job_bytes()), int(_lib.e2_upd_rest_bytes())
        bj = (C.c_char * max(rj * len(jobs), 1))()
        br = (C.c_char * max(rr * len(rests), 1))()
        tile0, lds = 0, 16
        keep = []
        for n, (off, wf, wd_, shape, reg) in enumerate(jobs):
            nt, lb = C.c_int(), C.c_size_t()
            cout, cin, kd, kh, kw = (int(v) for v in shape)
            _chk(_lib.e2_upd_job_fill(C.byref(bj, n * rj), int(off),
                                      C.c_void_p(wf.data_ptr()) if wf is not None else None,
                                      C.c_void_p(wd_.data_ptr()) if wd_ is not None else None,
                                      cout, cin, kd, kh, kw, float(reg), tile0, C.byref(nt), C.byref(lb)),
                 "e2_upd_job_fill")
            tile0 += nt.value
            lds = max(lds, int(lb.value))
            keep += [wf, wd_]
        for n, (off, cnt, reg) in enumerate(rests):
            _chk(_lib.e2_upd_rest_fill(C.byref(br, n * rr), int(off), int(cnt), float(reg)), "e2_upd_rest_fill")
        return dict(jobs=torch.frombuffer(bytearray(bj), dtype=torch.uint8).to(self.device), njobs=len(jobs),
                    ntiles=tile0, rest=torch.frombuffer(bytearray(br), dtype=torch.uint8).to(self.device),
                    nrest=len(rests), lds=lds, keep=keep)

    def adam_pack_step(self, p, g, m, s, upd, hyper, gdiv=None, gmul=1.0, zero_g=False):
        """adam_step + the repack of every conv's weight images in one launch (make_upd_jobs)"""
        _chk(_lib.e2_adam_pack_step(self.h, _fp(p), _fp(g), _fp(m), _fp(s),
                                    C.c_void_p(upd['jobs'].data_ptr()), upd['njobs'], upd['ntiles'],
                                    C.c_void_p(upd['rest'].data_ptr()), upd['nrest'], _fp(hyper),
                                    _fp(gdiv), float(gmul), int(bool(zero_g)), upd['lds']),
             "e2_adam_pack_step")

    def sgd_step(self, p, g, d, seg_off, seg_reg, hyper, gdiv=None, gmul=1.0, zero_g=False):
        _chk(_lib.e2_sgd_step_ex(self.h, _fp(p), _fp(g), _fp(d), p.numel(),
                                 C.c_void_p(seg_off.data_ptr()), _fp(seg_reg),
                                 seg_reg.numel(), _fp(hyper), _fp(gdiv), float(gmul),
                                 int(bool(zero_g))), "e2_sgd_step_ex")

    # ---- first launch of a step that may stand k times in one graph (e2hip.h: e2_step_prologue) ----
    def step_prologue(self, state, ring=None, dst=None, src=None, hist=None):
        """with L = launches on ``state`` (zeroed int64[2]) so far: dst = ring[L % n_slots] (ring:
        (n_slots, slot_floats) f32) and hist[(L - 1) % hist_slots] = src[:hist.shape[1]]"""
        assert state.dtype == torch.int64 and state.numel() >= 2 and state.is_contiguous()
        if ring is not None:
            assert ring.dim() == 2 and ring.is_contiguous() and ring.dtype == torch.float32
            assert dst is not None and dst.is_contiguous() and dst.numel() >= ring.shape[1]
        if hist is not None:
            assert hist.dim() == 2 and hist.is_contiguous() and src is not None and src.numel() >= hist.shape[1]
        _chk(_lib.e2_step_prologue(self.h, _fp(ring), int(ring.shape[0]) if ring is not None else 0,
                                   int(ring.shape[1]) if ring is not None else 0, _fp(dst), _fp(src),
                                   int(hist.shape[1]) if hist is not None else 0, _fp(hist),
                                   int(hist.shape[0]) if hist is not None else 0,
                                   C.c_void_p(state.data_ptr())), "e2_step_prologue")

    # ---- graph capture / events --------------------------------------------------------
    def graph_begin(self):
        _chk(_lib.e2_graph_begin(self.h), "e2_graph_begin")

    def graph_end(self):
        g = C.c_void_p()
        _chk(_lib.e2_graph_end(self.h, C.byref(g)), "e2_graph_end")
        return g

    def graph_launch(self, g):
        _chk(_lib.e2_graph_launch(self.h, g), "e2_graph_launch")

    def graph_destroy(self, g):
        _chk(_lib.e2_graph_destroy(g), "e2_graph_destroy")

    def graph_debug_dot(self, g, path, verbose=True):
        _chk(_lib.e2_graph_debug_dot(g, str(path).encode(), 1 if verbose else 0), "e2_graph_debug_dot")

    def event(self):
        e = C.c_void_p()
        _chk(_lib.e2_event_create(C.byref(e)), "e2_event_create")
        return e

    def record(self, e):
        _chk(_lib.e2_event_record(self.h, e), "e2_event_record")

    def elapsed_ms(self, e0, e1) -> float:
        ms = C.c_float()
        _chk(_lib.e2_event_elapsed_ms(e0, e1, C.byref(ms)), "e2_event_elapsed_ms")
        return float(ms.value)
