"""The bundled NetCDF backend (2 000 lines of hand-written binary I/O: header parsing of untrusted files, byte swapping
on threads, raw HDF5 chunk paths) under AddressSanitizer + UndefinedBehaviorSanitizer: `make -C cdftools_amd/netcdf asan`
builds instrumented libraries and the NetCDF tests are rerun against them in a child Python that preloads the sanitizer
runtime.  CPU only (GPU AddressSanitizer is not available on this pool)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cdftools_amd", "netcdf")


def runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_netcdf_backend_is_clean_under_asan_and_ubsan():
    asan, ubsan = runtime("libasan.so"), runtime("libubsan.so")
    if not asan:
        pytest.skip("no libasan.so next to this gcc")
    subprocess.check_call(["make", "-B", "-C", SRC, "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "cdftools_amd", "lib", "asan", "libncclassic.so")
    env = dict(os.environ, LD_PRELOAD=":".join(x for x in (asan, ubsan) if x), CDFT_NCCLASSIC_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    plug = os.path.join(ROOT, "cdftools_amd", "lib", "asan", "libnch5.so")
    tests = [os.path.join(ROOT, "tests", "test_ncclassic.py")]
    if os.path.exists(plug):
        env["CDFT_NCH5_LIB"] = plug
        tests.append(os.path.join(ROOT, "tests", "test_netcdf4.py"))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"] + tests,
                       env=env, capture_output=True, text=True, timeout=1200)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert p.returncode == 0 and " passed" in out, out[-6000:]
