#!/usr/bin/env python3
"""cdfmean with 1 / 2 / 5 basins, two or four vectors per lane (CDFT_MEAN_VPT): kernels alone, HIP events.
    python3 tools/mean_nb_vpt.py eorca025"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cdftools_amd import hip  # noqa: E402
import bench_tools  # noqa: E402

grid = sys.argv[1] if len(sys.argv) > 1 else "eorca025"
npi, npj, npk = bench_tools.GRIDS[grid]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *sh: torch.rand(sh, device=dev, generator=g)   # noqa: E731
e1, e2 = (2000 + 26000 * rnd(npj, npi)).cpu().numpy(), (2000 + 26000 * rnd(npj, npi)).cpu().numpy()
e3 = (1 + 199 * rnd(npk, npj, npi)).cpu().numpy()
tm = (rnd(npk, npj, npi) > 0.35).float().cpu().numpy()
zv = -2 + 32 * rnd(npk, npj, npi)
for nb in (1, 2, 5):
    ib = (np.random.default_rng(nb).random((npj, npi, nb)) > 0.4).astype(np.int16)
    p = hip.MeanPlan(npi, npj, npk, nb)
    p.set_mesh(e1, e2, tm, ib, e3=e3)
    p.set_form(3)
    dsum = torch.empty((npk, nb, 4), dtype=torch.float64, device=dev)
    for vpt in ("2", "4"):
        os.environ["CDFT_MEAN_VPT"] = vpt
        ms = bench_tools.timed(lambda: p.compute(zv, device_out=dsum), 20)
        print(json.dumps({"grid": grid, "basins": nb, "vectors_per_lane": int(vpt), "kernels_ms": round(ms, 4),
                          "frac_of_hbm_peak": round(npi * npj * npk * 9.0 / (ms * 1e-3) / 8e12, 4)}), flush=True)
    del p
