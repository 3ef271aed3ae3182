#!/usr/bin/env python3
"""What an HBM allocation costs on this box: hipMalloc + hipFree of 1 .. 120 GB through the C ABI (cdft_malloc / cdft_free),
each size three times in this process, the first allocation of the process timed apart.  (Why: the cdfmoc program's resident
pool is one 61 GB allocation at eORCA12 x 12 records; a file read delivers ~35 GB/s from the page cache.)
    python tools/alloc_probe.py  ->  one JSON line"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdftools_amd import _lib  # noqa: E402

lib = _lib.load()
out = {}
t0 = time.perf_counter()
p = C.c_void_p()
assert lib.cdft_malloc(C.byref(p), 1 << 20) == 0
lib.cdft_free(p)
out["first_call_s"] = round(time.perf_counter() - t0, 4)
for gb in (1, 4.7, 15, 30, 61, 120):
    n = int(gb * 1e9)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        rc = lib.cdft_malloc(C.byref(p), n)
        t1 = time.perf_counter()
        if rc != 0:
            ts.append(None)
            continue
        lib.cdft_free(p)
        t2 = time.perf_counter()
        ts.append([round(t1 - t0, 4), round(t2 - t1, 4)])
    out[f"{gb}GB_malloc_free_s"] = ts
print(json.dumps(out))
