#!/usr/bin/env python3
"""A long fuzz of cdfmoc's zonal kernels against the oracle: random row lengths (1 .. 20 000 cells: one wave, several
waves, rows cut into segments), rows / levels / records / basins, records that are or are not 16-byte aligned, every
kernel family and random shapes of it (cdft_moc_set_tuning), basin masks with few or many bit patterns, and -- one case
in four -- NaN / Inf / overflowing velocities on land, in the ocean, outside some basin.  Stated tolerance of the path
(|d| <= 1e-13 * the same integral over |terms|), NaN where and only where the reference's expression gives NaN.
tests/test_moc.py and tests/test_nonfinite.py run fixed cases; this runs new ones for FUZZ_SECONDS (default 300) and
stops at the first failure.  Run on the GPU box:  FUZZ_SECONDS=300 python tools/fuzz_moc.py [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import synth  # noqa: E402
from cdftools_amd import hip  # noqa: E402
from cdftools_amd._lib import CdftError  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else int(time.time()) & 0xffffff
rng = np.random.default_rng(seed)
stop = time.time() + float(os.environ.get("FUZZ_SECONDS", "300"))
LENGTHS = [1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 255, 256, 257, 259, 333, 1023, 1024, 1190, 1442, 2303, 2304, 2305, 4322, 4607, 4608, 4609,
           9207, 9208, 9209, 9216, 9300, 10000, 12960, 12962, 16000, 16383, 16385, 19999, 24576, 24577, 30000]
n = refused = cells = 0
while time.time() < stop:
    npi = int(rng.choice(LENGTHS)) if rng.random() < 0.7 else int(rng.integers(1, 40000))
    npj, npk = int(rng.integers(1, 6)), int(rng.integers(2, 6))
    npt = int(rng.choice([1, 1, 2, 3, 5, 12, 17, 33]))
    if npi * npj * npk * npt > 6_000_000:
        npt = max(1, 6_000_000 // (npi * npj * npk))
    nb = int(rng.choice([1, 2, 3, 5, 5, 5, 8]))
    g = synth.make_grid(npi, npj, npk, seed=int(rng.integers(1 << 30)), nbasins=nb, land=float(rng.choice([0.0, 0.35, 0.9])))
    e1v, e3v, ib = synth.moc_inputs(g)
    if rng.random() < 0.3:                                    # overlapping basins: many bit patterns (the select kernel's business)
        ib = (rng.random(ib.shape) < 0.5).astype(np.int16)
    zv = synth.velocity(g, npt, seed=int(rng.integers(1 << 30)))
    poisoned = rng.random() < 0.25
    if poisoned:
        flat = zv.reshape(-1)
        k = int(rng.integers(1, 6))
        flat[rng.integers(0, flat.size, k)] = rng.choice([np.nan, np.inf, -np.inf, 3e38, -3e38, 1e37], k).astype(np.float32)
    n3 = npi * npj * npk
    aligned = bool(rng.integers(2))
    stride = n3 + (int(rng.integers(0, 5)) * 4 if aligned else int(rng.choice([1, 2, 3, 5, 7])))
    if not aligned and stride % 4 == 0:
        stride += 1
    buf = torch.zeros(4 + npt * stride, dtype=torch.float32, device="cuda")
    view = buf[4:] if aligned else buf[int(rng.choice([1, 2, 3])):]
    for t in range(npt):
        view[t * stride: t * stride + n3] = torch.from_numpy(zv[t].ravel()).cuda()
    plan = hip.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    tune = (0, 0, 0)
    if rng.random() < 0.6:
        tune = (int(rng.choice([0, 1, 2, 3, 4, 6, 8, 16])), int(rng.choice([0, 1, 2, 3, 5, 6, 8, 9, 10, 12])), int(rng.integers(0, 4)))
    out = torch.empty((npt, npk, npj, nb), dtype=torch.float64, device="cuda")
    try:
        plan.set_tuning(*tune)
        plan.compute(view, out=out, rec_stride=stride, npt=npt)
    except CdftError:                                          # a shape this row length / mode does not allow: refused, not wrong
        refused += 1
        plan.close()
        continue
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    desc = plan.describe()
    # the same records staged into the plan's pool (a row's weights and records in one run): cdft_moc_compute_staged
    got_pool = None
    try:
        plan.stage(zv if rng.random() < 0.5 else torch.from_numpy(zv).cuda())
        got_pool = plan.compute_staged(np.zeros(got.shape))
        desc_pool = plan.describe()
        # ... and read a window of levels per launch (the cdfmoc program's stream route): the same bits, NaNs included
        nlev = int(rng.integers(1, npk + 1))
        got_win = plan.compute_staged_windows(nlev, np.zeros(got.shape))
        if not np.array_equal(got_win, got_pool, equal_nan=True):
            print(f"FAIL seed={seed} case={n} npi={npi} npj={npj} npk={npk} npt={npt} nb={nb} windows of {nlev} levels differ from one launch\n  {desc_pool}", flush=True)
            sys.exit(1)
    except CdftError:                                          # (the tuned shape may not cover a pool row either)
        pass
    plan.close()
    with np.errstate(all="ignore"):
        ref = oracle_lib.moc(e1v, e3v, ib, zv)
        clean = np.where(np.isfinite(zv) & (np.abs(zv) < 1e30), zv, 0).astype(np.float32)
        lim = 1e-13 * np.abs(oracle_lib.moc(e1v, e3v, ib, np.abs(clean))) + 1e-300
    ok = np.array_equal(np.isnan(got), np.isnan(ref))
    if ok:
        inf = np.isinf(ref)
        ok = np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf])
    if ok:
        fin = np.isfinite(ref)
        # with poisoned records the bound of a row that holds an overflow is not the clean one: compare the clean rows strictly
        err = np.abs(got[fin] - ref[fin])
        ok = bool(np.all(err <= (lim[fin] if not poisoned else np.maximum(lim[fin], 1e-13 * np.abs(ref[fin])))))
    if ok and got_pool is not None:                            # the staged path: the same checks
        ok = np.array_equal(np.isnan(got_pool), np.isnan(ref))
        if ok:
            inf = np.isinf(ref)
            ok = np.array_equal(np.isinf(got_pool), inf) and np.array_equal(got_pool[inf], ref[inf])
        if ok:
            fin = np.isfinite(ref)
            err = np.abs(got_pool[fin] - ref[fin])
            ok = bool(np.all(err <= (lim[fin] if not poisoned else np.maximum(lim[fin], 1e-13 * np.abs(ref[fin])))))
        if not ok:
            desc = desc_pool + "  [staged records]"
    n += 1
    cells += n3 * npt
    if not ok:
        print(f"FAIL seed={seed} case={n} npi={npi} npj={npj} npk={npk} npt={npt} nb={nb} aligned={aligned} stride={stride} tune={tune} "
              f"poisoned={poisoned}\n  {desc}", flush=True)
        sys.exit(1)
print(f"fuzz_moc: seed {seed}: {n} cases ({refused} shapes refused), {cells:.3e} cells, every result within the stated tolerance of the oracle")
