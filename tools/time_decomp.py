#!/usr/bin/env python3
"""cdfmoc -decomp on one record of a realistic grid (host arrays in, as the program hands them): wall time per call, and --
under rocprofv3 --kernel-trace --stats -- the time of each dec_* kernel.    python tools/time_decomp.py eorca025|eorca12"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from cdftools_amd import hip
import bench
bench.hip = hip
grid = sys.argv[1] if len(sys.argv) > 1 else "eorca025"
npi, npj, npk = {"eorca025": (1442, 1207, 75), "eorca12": (4322, 3606, 75)}[grid]
dev = torch.device("cuda", 0)
e1v, e3v, ib, zv = bench.build_inputs(npi, npj, npk, 1, 5, 0, npj, dev)
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(s, device=dev, generator=g)
t = (-2 + 30 * rnd(npk, npj, npi)); s = (33 + 4 * rnd(npk, npj, npi))
tm = (rnd(npk, npj, npi) > 0.3).to(torch.uint8); um = (rnd(npk, npj, npi) > 0.3).to(torch.uint8)
e1u = 2000 + 26000 * rnd(npj, npi); zcoef = rnd(npj, npi) - 0.5
gdept = torch.linspace(5, 5800, npk)
plan = hip.MocPlan(npi, npj, npk, 5)
plan.set_mesh(e1v, e3v, ib)
plan.decomp_set_mesh(e3v.cpu().numpy(), e1u.cpu().numpy(), zcoef.cpu().numpy(), gdept.numpy(), tm.cpu().numpy(), um.cpu().numpy())
z = zv[0].cpu().numpy(); t = t.cpu().numpy(); s = s.cpu().numpy()
for _ in range(2): o = plan.decomp(z, t, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n): o = plan.decomp(z, t, s)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
cells = npi * npj * npk
print(grid, "decomp ms", round(ms, 3), "cells/s %.3e" % (cells / ms * 1e3), "GB/s at 22 B/cell %.0f" % (22 * cells / ms / 1e6))
