"""The C-ABI shared library loads and exports every symbol include/gandtr_hip.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gandtr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gdt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gandtr_amd import _hip
    if not os.path.exists(_hip.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_hip.lib_path())
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing symbol %s" % n
    assert set(names) == set(_hip.SIGNATURES), set(names) ^ set(_hip.SIGNATURES)


def test_binding_loads_and_reports_version():
    from gandtr_amd import _hip
    lib = _hip.load()
    assert lib.gdt_version().decode().startswith("gandtr_hip")
    assert lib.gdt_last_error() is not None


def test_invalid_arguments_raise_value_error_without_gpu():
    """argument validation happens before any HIP call, so it is testable on a GPU-less host"""
    from gandtr_amd import _hip
    lib = _hip.load()
    h = ctypes.c_void_p()
    _hip.check(lib.gdt_net_create(ctypes.byref(h)))
    out = ctypes.c_int()
    with pytest.raises(ValueError):
        _hip.check(lib.gdt_net_input(h, 9, None, None, None, ctypes.byref(out)))       # > 8 channels
    _hip.check(lib.gdt_net_input(h, 3, None, None, None, ctypes.byref(out)))
    with pytest.raises(ValueError):
        _hip.check(lib.gdt_net_maxpool(h, 99, 2, 2, 0, ctypes.byref(out)))             # unknown tensor id
    with pytest.raises(ValueError):
        _hip.check(lib.gdt_net_forward(h, None, 1, 8, 8, 8, 8, 1.0, None, 0, None, 0, None))   # not finalized
    lib.gdt_net_destroy(h)


def test_missing_library_fails_loudly(monkeypatch):
    from gandtr_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "_LIB_PATH", "/nonexistent/libgandtr_hip.so")
    with pytest.raises(_hip.HipLibraryMissing):
        _hip.load()
