"""The C-ABI shared library loads and exports every symbol include/cdftools_hip.h declares
(CPU-only: no compute call is made without a GPU)."""
import ctypes as C
import os
import re

import pytest

from cdftools_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "cdftools_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cdft_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    syms = header_symbols()
    assert syms, "no entry points parsed from the header"
    assert sorted(_lib.SIGNATURES) == syms


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in cdftools_hip.h but not exported"
    assert lib.cdft_version() >= 100


def test_device_count_never_fails():
    n = C.c_int(-1)
    assert _lib.load().cdft_device_count(C.byref(n)) == 0
    assert n.value >= 0


def test_compute_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    st = _lib.load().cdft_moc_create(C.byref(h), 8, 8, 4, 1)
    assert st == _lib.CDFT_ERR_NODEVICE
    assert b"no CPU fallback" in _lib.load().cdft_last_error()
    with pytest.raises(_lib.CdftError):
        from cdftools_amd.hip import MocPlan

        MocPlan(8, 8, 4, 1)


def test_argument_checks_precede_device_use():
    h = C.c_void_p()
    lib = _lib.load()
    assert lib.cdft_moc_create(C.byref(h), 0, 8, 4, 1) == _lib.CDFT_ERR_ARG
    assert lib.cdft_moc_create(C.byref(h), 8, 8, 1, 1) == _lib.CDFT_ERR_ARG
    assert lib.cdft_moc_create(C.byref(h), 8, 8, 4, 9) == _lib.CDFT_ERR_ARG
    assert lib.cdft_moc_create(None, 8, 8, 4, 1) == _lib.CDFT_ERR_ARG
