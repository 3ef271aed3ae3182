#!/usr/bin/env python3
"""Runs cdfvint's and cdfvertmean's kernels once on host arrays of a grid size; meant to be run under
rocprofv3 --kernel-trace --stats, whose kernel durations are the measurement (the wrappers take host arrays, so a
wall-clock time here would be the PCIe copy)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdftools_amd import hip  # noqa: E402

GRIDS = {"eorca025": (1442, 1207, 75), "eorca12": (4322, 3606, 75)}
ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="eorca025", choices=sorted(GRIDS))
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
npi, npj, npk = GRIDS[args.grid]
rng = np.random.default_rng(0)
t = rng.random((npk, npj, npi), dtype=np.float32)
m = (rng.random((npk, npj, npi), dtype=np.float32) > 0.35).astype(np.float32)
e3 = (1 + 199 * rng.random((npk, npj, npi), dtype=np.float32)).astype(np.float32)
kout = np.arange(npk, dtype=np.int32)                 # an output depth after every level
hit = np.zeros(npk, np.int32)
clip = np.full(npk, 1e9, np.float32)
dv = np.ones(npk, np.float32)
for _ in range(args.reps):
    hip.vint_cumul(t, m, kout, hit, clip, dv, npk, e3t=e3)
    hip.vertmean(t, m, 0.0, 5000.0, 10.0, 4000.0, e3=e3)
print("done", args.grid)
