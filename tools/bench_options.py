#!/usr/bin/env python3
"""The option paths of the §8a tools that tools/bench_tools.py does not time (it takes each tool's usual form): one line per
path, resident inputs, HIP events, with the bytes that path has to move -- to catch a path that runs far below its neighbours
(cdfmean with five sub-basins was one: 0.19 of the roofline until round 4).
    python3 tools/bench_options.py --grid eorca025"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cdftools_amd import hip  # noqa: E402
import bench_tools  # noqa: E402

PEAK = 8000.0


def line(tool, grid, cells, bpc, ms, note=""):
    gbs = cells * bpc / (ms * 1e-3) / 1e9
    print(json.dumps({"tool": tool, "grid": grid, "cells": cells, "bytes_per_cell": bpc, "ms": round(ms, 4), "GBps": round(gbs, 1),
                      "frac_of_hbm_peak": round(gbs / PEAK, 4), "note": note}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="eorca025", choices=sorted(bench_tools.GRIDS))
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    npi, npj, npk = bench_tools.GRIDS[args.grid]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    rnd = lambda *sh: torch.rand(sh, device=dev, generator=g)   # noqa: E731
    n3 = npi * npj * npk
    e1, e2 = 2000 + 26000 * rnd(npj, npi), 2000 + 26000 * rnd(npj, npi)
    e3 = 1 + 199 * rnd(npk, npj, npi)
    e31d = torch.from_numpy(np.linspace(1, 200, npk).astype(np.float32)).to(dev)
    T = lambda fn: bench_tools.timed(fn, args.reps)   # noqa: E731

    # ---- cdfheatc: full-step, r4 mask, mixed-layer forms, in both sum modes
    tmask8 = (rnd(npk, npj, npi) > 0.35).to(torch.uint8)
    temp = -2 + 32 * rnd(npk, npj, npi)
    mxl = 10 + 500 * rnd(npj, npi)
    gdepw = np.linspace(0, 5800, npk).astype(np.float32)
    dsum = torch.empty((npk, 3), dtype=torch.float64, device=dev)
    for name, kw, mesh, bpc in (("-full", {}, dict(e31d=e31d), 5.0), ("-mxloption 1", dict(rmxldep=mxl, gdepw=gdepw, mxloption=1), dict(e3t=e3), 9.0),
                                ("r4 mask (-M)", {}, dict(e3t=e3, r4mask=True), 12.0)):
        p = hip.HeatcPlan(npi, npj, npk)
        r4 = mesh.pop("r4mask", False)
        p.set_mesh(e1, e2, (tmask8.float() * 0.5) if r4 else tmask8, **mesh)
        line("cdfheatc " + name + ", r8 sums", args.grid, n3, bpc, T(lambda: p.compute(temp, device_out=dsum, **kw)))
        p.set_sum_mode(True)
        line("cdfheatc " + name + ", reference-order r4 sums", args.grid, n3, bpc, T(lambda: p.compute(temp, **kw)), "the call (host result)")
        del p
    del tmask8, temp

    # ---- cdfmhst from V, T, S (not -vt), and per level (-zdim)
    v, t, s = rnd(npk, npj, npi) - 0.5, -2 + 32 * rnd(npk, npj, npi), 30 + 8 * rnd(npk, npj, npi)
    m = (rnd(5, npj, npi) > 0.4).float()
    p = hip.MhstPlan(npi, npj, npk, 5)
    p.set_mesh(e1, m, e3v=e3)
    dhs = (torch.empty((npj, 5), dtype=torch.float64, device=dev), torch.empty((npj, 5), dtype=torch.float64, device=dev))
    line("cdfmhst from V, T, S", args.grid, n3, 16.0, T(lambda: p.compute(zv=v, zt=t, zs=s, device_out=dhs)), "v, t, s, e3v: 16 B per cell (the rows to the north from cache)")
    dz = (torch.empty((npk, npj, 5), dtype=torch.float64, device=dev), torch.empty((npk, npj, 5), dtype=torch.float64, device=dev))
    line("cdfmhst -vt -zdim", args.grid, n3, 12.0, T(lambda: p.compute(vt=v, vs=t, zdim=True, device_out=dz)))
    del p, m

    # ---- cdfpsi -V, -mean
    p = hip.PsiPlan(npi, npj, npk)
    line("cdfpsi -V", args.grid, n3, 8.0, T(lambda: p.compute(zv=v, e1v=e1, e3v=e3, host_out=False, want_r8=False)), "the scan runs along i")
    line("cdfpsi -mean", args.grid, n3, 16.0, T(lambda: p.compute(zu=t, e2u=e2, e3u=e3, zv=v, e1v=e1, e3v=e3, host_out=False, want_r8=False)))
    line("cdfpsi -full", args.grid, n3, 4.0, T(lambda: p.compute(zu=t, e2u=e2, e31d=e31d, host_out=False, want_r8=False)))
    del p

    # ---- cdfcurl at T points, r8 output
    p = hip.CurlPlan(npi, npj)
    p.set_mesh(e1, e2, e1, e2, lperio=True)
    line("cdfcurl -T", args.grid, n3, 12.0, T(lambda: p.compute(v, t, ltpoint=True, out_r8=False)), "two passes: F points to a work array, then T points")
    line("cdfcurl, r8 result", args.grid, n3, 16.0, T(lambda: p.compute(v, t, out_r8=True)))
    del p

    # ---- cdfvT -vvl
    nfr = 3
    u4, v4 = rnd(nfr, npk, npj, npi) - 0.5, rnd(nfr, npk, npj, npi) - 0.5
    t4, s4 = -2 + 32 * rnd(nfr, npk, npj, npi), 30 + 8 * rnd(nfr, npk, npj, npi)
    e3u4, e3v4 = 1 + 199 * rnd(nfr, npk, npj, npi), 1 + 199 * rnd(nfr, npk, npj, npi)
    line("cdfvT -vvl (3 frames)", args.grid, n3 * nfr, 24.0 + 24.0 / nfr, T(lambda: hip.vt_compute(u4, v4, t4, s4, e3u=e3u4, e3v=e3v4, out_r4=True)))
    line("cdfvT (3 frames)", args.grid, n3 * nfr, 16.0 + 16.0 / nfr, T(lambda: hip.vt_compute(u4, v4, t4, s4, out_r4=True)))
    del u4, v4, t4, s4, e3u4, e3v4

    # ---- cdfzonalmean with -max, one basin
    tm = (rnd(npk, npj, npi) > 0.35).float().cpu().numpy()
    sea = torch.from_numpy(tm[0]).to(dev) > 0
    p = hip.ZonalPlan(npi, npj, npk, 1)
    p.set_mesh(e1.cpu().numpy(), e2.cpu().numpy(), tm, sea[:, :, None].float().cpu().numpy())
    line("cdfzonalmean, 1 basin", args.grid, n3, 5.0, T(lambda: p.compute(t, mean_form=True, device_out=True)))
    line("cdfzonalmean -max, 1 basin", args.grid, n3, 5.0, T(lambda: p.compute(t, mean_form=True, minmax=True, device_out=True)))
    del p

    # ---- the EOS members in their library forms (r8 result), the per-point reference depth, JMD94, the neutral density
    te, se = -2 + 32 * rnd(n3), 30 + 8 * rnd(n3)
    o8 = torch.empty(n3, dtype=torch.float64, device=dev)
    pr = 4000 * rnd(n3)
    for name, fn in (("sigma0, r8 result", lambda: hip.sigma0(te, se, out=o8)), ("sigmai 2000 m, r8 result", lambda: hip.sigmai(te, se, 2000.0, out=o8)),
                     ("sigmai, reference depth per point", lambda: hip.sigmai(te, se, pref2d=pr, out=o8)),
                     ("sigmai 2000 m, JMD94", lambda: hip.sigmai(te, se, 2000.0, eos=hip.EOS_JMD94, out=o8)),
                     ("sigmantr", lambda: hip.sigmai(te, se, 0.0, eos=hip.EOS_NTR, out=o8))):
        line(name, args.grid, n3, 16.0, T(fn), "T 4 + S 4 + out 8 B per point; bound by the r8 vector pipeline, not by HBM")


if __name__ == "__main__":
    main()
