"""CPU: libe2hip.so builds (hipcc cross-compiles gfx950 without a GPU), loads,
and exports every entry point include/e2hip.h declares.  No compute calls."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "e2hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(e2_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "elektronn2_amd", "libe2hip.so"))
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    lib.e2_version.restype = ctypes.c_int
    assert lib.e2_version() >= 1
    lib.e2_conv3d_workspace_bytes.restype = ctypes.c_size_t
    assert lib.e2_conv3d_workspace_bytes(200, 200, 1, 3, 3) > 200 * 200 * 9 * 4


def test_backend_binds_all_symbols_and_has_no_cpu_fallback():
    import pytest
    import torch
    from elektronn2_amd import backend
    assert set(declared_symbols()) == set(backend.EXPORTED_SYMBOLS)
    if not torch.cuda.is_available():
        with pytest.raises(backend.E2Error, match="HIP-only"):
            backend.Context(0)
        with pytest.raises(backend.E2Error, match="no CPU fallback"):
            backend.t5(torch.zeros(1, 1, 1, 1, 1))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "elektronn2_amd")
    bad = []
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                if re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(dp, fn)).read(), re.M):
                    bad.append(fn)
    assert not bad, bad
