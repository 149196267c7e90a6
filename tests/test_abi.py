"""The C-ABI library loads and exports every symbol include/gg.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import duckdb_pgq_amd.gg as ggmod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "gg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _header_functions() == sorted(ggmod.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(ggmod.LIB_PATH)
    for name in _header_functions():
        assert hasattr(lib, name), name


def test_version_and_error_strings():
    lib = ggmod.load_library()
    assert lib.gg_version().startswith(b"gg ")
    assert isinstance(lib.gg_last_error(), bytes)


def test_no_cpu_fallback():
    """Without a HIP device the context cannot be created (and nothing computes)."""
    lib = ggmod.load_library()
    n = ctypes.c_int(-1)
    assert lib.gg_device_count(ctypes.byref(n)) == 0
    if n.value == 0:
        h = ctypes.c_void_p()
        assert lib.gg_ctx_create(0, ctypes.byref(h)) == -7  # GG_ERR_NO_DEVICE
        assert b"no CPU fallback" in lib.gg_last_error()
