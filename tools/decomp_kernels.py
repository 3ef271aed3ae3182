#!/usr/bin/env python3
"""cdfmoc -decomp at a grid's size, one record through cdft_moc_decomp (host arrays in and out): run under tools/kstat.sh to get
the kernels' durations.   bash tools/kstat.sh dec_,eos_,moc_ decomp tools/decomp_kernels.py eorca025"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cdftools_amd import hip  # noqa: E402
import bench_tools  # noqa: E402

npi, npj, npk = bench_tools.GRIDS[sys.argv[1] if len(sys.argv) > 1 else "eorca025"]
rng = np.random.default_rng(5)
f32 = np.float32
nb = 5
e1v = (2000 + 26000 * rng.random((npj, npi))).astype(f32)
e3v = (1 + 199 * rng.random((npk, npj, npi))).astype(f32)
vmask = (rng.random((npk, npj, npi)) > 0.3).astype(np.uint8)
ib = (rng.random((nb, npj, npi)) > 0.4).astype(np.int16)
ib[0] = 1
p = hip.MocPlan(npi, npj, npk, nb)
p.set_mesh(e1v, e3v * vmask, ib)
p.decomp_set_mesh(e3v * vmask, e1v, (1e-4 * (0.5 + rng.random((npj, npi)))).astype(f32), np.linspace(0.5, 5800, npk).astype(f32), vmask, vmask)
zv = (rng.random((npk, npj, npi)) - 0.5).astype(f32)
t = (-2 + 32 * rng.random((npk, npj, npi))).astype(f32)
s = (30 + 8 * rng.random((npk, npj, npi))).astype(f32)
for _ in range(4):
    out = p.decomp(zv, t, s)
print({k: float(np.abs(v).max()) for k, v in out.items()})
