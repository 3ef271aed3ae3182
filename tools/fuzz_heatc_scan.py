#!/usr/bin/env python3
"""A long fuzz of cdfheatc's reference-order sums: heatc_scan_kernel (integer sums inside a binade, decoupled look-back)
against the plain chain of v_add_f32 (CDFT_HEATC_SEQ=chain), bit for bit, on random shapes around the chunk / block /
ticket sizes, every mask kind, e3t / -full, -mxloption, start values handed in, and value distributions chosen to break
the integer picture (ties, sign changes, sums that cross powers of two and zero, huge ranges, non-finite values, -0.0).
tests/test_heatc.py runs a fixed set of such cases; this runs new ones for FUZZ_SECONDS (default 300) and stops at the
first mismatch.  Run on the GPU box:  FUZZ_SECONDS=300 python tools/fuzz_heatc_scan.py [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdftools_amd import hip  # noqa: E402

f32 = np.float32
seed = int(sys.argv[1]) if len(sys.argv) > 1 else int(time.time()) & 0xffffff
rng = np.random.default_rng(seed)
stop = time.time() + float(os.environ.get("FUZZ_SECONDS", "300"))
SIZES = [1, 3, 4, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1442, 2047, 4095, 4096, 4097, 8191, 32767, 32768, 32769, 65537, 131071, 131073]
KINDS = ["uniform", "normal0", "pow2", "halves", "huge_range", "tiny", "nonfinite", "land", "minus_zero", "alternating", "const"]
n = cells = 0
while time.time() < stop:
    total = int(rng.choice(SIZES)) * int(rng.integers(1, 40))
    npi = int(rng.choice([d for d in (1, 2, 3, 4, 5, 7, 16, 61, 128, 181, 512, 1442, 4322) if d <= total] or [1]))
    npj = max(1, total // npi)
    npk = int(rng.integers(1, 6))
    shape = (npk, npj, npi)
    kind = KINDS[int(rng.integers(len(KINDS)))]
    scale = 10.0 ** rng.integers(-8, 9)
    e1t = (scale * (0.5 + rng.random((npj, npi)))).astype(f32)
    e2t = (scale * (0.5 + rng.random((npj, npi)))).astype(f32)
    e3t = (1 + 199 * rng.random(shape)).astype(f32)
    e31d = (1 + 199 * rng.random(npk)).astype(f32)
    mask = (rng.random(shape) > rng.choice([0.0, 0.1, 0.5, 0.95, 0.999])).astype(f32)
    temp = (-2 + 32 * rng.random(shape)).astype(f32)
    with np.errstate(all="ignore"):
        if kind == "normal0":
            temp = rng.normal(0, 1, shape).astype(f32)
        elif kind == "pow2":
            e1t[:] = (2.0 ** rng.integers(0, 4, (npj, npi))).astype(f32); e2t[:] = 1
            e3t[:] = (2.0 ** rng.integers(-3, 3, shape)).astype(f32)
            temp = (2.0 ** rng.integers(-2, 3, shape) * rng.choice([-1, 1], shape)).astype(f32)
        elif kind == "halves":                                 # x / ulp(s) lands on .5 again and again
            e1t[:] = 1; e2t[:] = 1; e3t[:] = 1
            temp = (rng.integers(1, 8, shape) * 0.5).astype(f32)
        elif kind == "huge_range":
            temp = (temp * 10.0 ** rng.integers(-25, 25, shape)).astype(f32)
        elif kind == "tiny":
            e1t[:] = f32(1e-18); e2t[:] = f32(1e-18); e3t[:] = f32(1e-6)
        elif kind == "nonfinite":
            flat = temp.reshape(-1)
            flat[rng.integers(0, flat.size, 3)] = rng.choice([np.inf, -np.inf, np.nan, 3e38, -3e38], 3)
        elif kind == "land":
            mask[:, : npj // 2] = 0
        elif kind == "minus_zero":
            temp = np.where(rng.random(shape) < 0.7, f32(-0.0), temp).astype(f32)
        elif kind == "alternating":
            temp = (np.where(np.arange(npi) % 2 == 0, 1, -1)[None, None, :] * (1 + rng.random(shape))).astype(f32)
        elif kind == "const":
            e1t[:] = f32(1000.0); e2t[:] = f32(2500.0); e3t[:] = f32(10.0); temp[:] = f32(4.0)
    full, m8, opt = bool(rng.integers(2)), bool(rng.integers(2)), int(rng.integers(-1, 2))
    mxl = (10 + 300 * rng.random((npj, npi))).astype(f32)
    gdepw = np.cumsum(np.concatenate([[0], e31d[:-1]])).astype(f32)
    m = mask.astype(np.uint8) if m8 else (mask * f32(rng.choice([1.0, 0.75, -1.0]))).astype(f32)
    carry = (rng.normal(0, 1, (npk, 3)) * scale * scale * rng.choice([0.0, 0.0, 1.0, 1e3])).astype(f32)
    if rng.random() < 0.1:
        carry[:] = f32(-0.0)
    plan = hip.HeatcPlan(npi, npj, npk)
    plan.set_mesh(e1t, e2t, m, **(dict(e31d=e31d) if full else dict(e3t=e3t)))
    plan.set_sum_mode(True)
    kw = dict(rmxldep=mxl if opt else None, gdepw=gdepw, mxloption=opt)
    os.environ.pop("CDFT_HEATC_SEQ", None)
    # round 6: the level walkers with their default ring, with rings of 1..3 slots (the streaming workgroups then wait for slots),
    # with small tickets, and off (every workgroup walks its own ticket)
    for key in ("CDFT_HEATC_SCAN_WALKERS", "CDFT_HEATC_SCAN_RING", "CDFT_HEATC_SCAN_CHUNKS"):
        os.environ.pop(key, None)
    form = rng.random()
    if form < 0.15:
        os.environ["CDFT_HEATC_SCAN_WALKERS"] = "0"
    elif form < 0.45:
        os.environ["CDFT_HEATC_SCAN_RING"] = str(int(rng.integers(1, 4)))
    if rng.random() < 0.3:
        os.environ["CDFT_HEATC_SCAN_CHUNKS"] = "64"
    plan.chain(carry_in=carry)
    a = plan.compute(temp, **kw).view(np.uint64)
    os.environ["CDFT_HEATC_SEQ"] = "chain"
    plan.chain(carry_in=carry)
    b = plan.compute(temp, **kw).view(np.uint64)
    os.environ.pop("CDFT_HEATC_SEQ")
    plan.close()
    n += 1
    cells += npk * npj * npi
    # a NaN is a NaN: its sign and payload depend on which two operands met (Inf - Inf here, NaN + x there) and mean nothing
    fa, fb = a.view(np.float64), b.view(np.float64)
    same = np.array_equal(np.isnan(fa), np.isnan(fb)) and np.array_equal(a[~np.isnan(fa)], b[~np.isnan(fb)])
    if not same:
        print(f"MISMATCH seed={seed} case={n} kind={kind} shape={shape} full={full} m8={m8} opt={opt} env={ {k: v for k, v in os.environ.items() if k.startswith('CDFT_HEATC')} }", flush=True)
        print(a, b)
        sys.exit(1)
print(f"fuzz_heatc_scan: seed {seed}: {n} cases, {cells:.3e} cells, scan kernel == chain bit for bit on all")
