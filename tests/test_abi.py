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


def test_fortran_binding_constants_equal_the_header():
    """cdftools_hip.f90 repeats the header's integer constants (Fortran cannot include a C header): every CDFT_* parameter
    of the module has the value of the header's #define, and every entry point of the header has its BIND(C) interface."""
    hdr = open(os.path.join(ROOT, "include", "cdftools_hip.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(CDFT_[A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\b", hdr, flags=re.M)}
    f90 = open(os.path.join(ROOT, "cdftools_amd", "fortran", "cdftools_hip.f90")).read()
    f90 = re.sub(r"&\s*\n\s*&", " ", f90)                                   # continuation lines
    pars = {}
    for line in re.findall(r"INTEGER\(C_INT\),\s*PARAMETER\s*::\s*(.*)", f90):
        for name, val in re.findall(r"(CDFT_[A-Z0-9_]+)\s*=\s*(-?\d+)", line):
            pars[name] = int(val)
    assert len(pars) >= 15
    for name, val in pars.items():
        assert name in defs, f"{name} is not a #define of the header"
        assert defs[name] == val, f"{name}: {val} in cdftools_hip.f90, {defs[name]} in the header"
    bound = set(re.findall(r"BIND\(C,\s*name='(cdft_[a-z0-9_]+)'\)", f90))
    syms = header_symbols()
    assert bound and not [x for x in bound if x not in syms], "an interface of cdftools_hip.f90 binds a name the header does not declare"
    # the entries a program needs (plans and computations); the harness hooks (timers, tuning, cdft_malloc, ...) stay C-only
    needed = [x for x in syms if re.search(r"_(create|create_sharded|set_mesh|compute|destroy)$", x) and not x.startswith("cdft_timer_")]
    missing = [x for x in needed if x not in bound]
    assert not missing, f"no Fortran interface for {missing}"


def test_python_mirror_constants_equal_the_header():
    from cdftools_amd import hip
    hdr = open(os.path.join(ROOT, "include", "cdftools_hip.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(CDFT_[A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\b", hdr, flags=re.M)}
    seen = 0
    for mod in (hip, _lib):
        for name, val in vars(mod).items():
            key = name if name.startswith("CDFT_") else "CDFT_" + name
            if isinstance(val, int) and not isinstance(val, bool) and name.isupper() and key in defs:
                assert defs[key] == val, f"{mod.__name__}.{name} = {val}, the header says {defs[key]}"
                seen += 1
    assert seen >= 15


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
