#!/usr/bin/env python3
"""Which LAYOUT of the records in HBM does the cdfmoc zonal kernel stream best, and how much does the time move with where
the allocator put them?  One process, the bench workload (eORCA12, 12 records), every state timed over `--launches`
launches by HIP events on the launch stream, every result compared bit for bit with the dense layout's.

  dense        the reference's arrays as they are: records 4.68 GB apart, rows npiglo apart (cdft_moc_compute)
  stride       dense rows, the record stride padded / the base moved (cdft_moc_compute_pitched, row_stride = npiglo)
  pitch        rows P apart (whole 128-byte lines), records outer
  rows         rows outer: the npt records of a row next to each other, P apart; weights in their own pitched array
  rows+w       the same with the row's weights in front of its records: ONE contiguous run per workgroup
Each state is measured again after the pool has been freed and allocated anew (`--moves` times): the spread over the
moves is what the physical placement does to that layout.

Run on the GPU box:  python3 tools/layout_probe.py [--out gpurun_out/r04/layout_probe.jsonl]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from cdftools_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="eorca12")
ap.add_argument("--launches", type=int, default=12)
ap.add_argument("--moves", type=int, default=3)
ap.add_argument("--out", default="")
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()

npi, npj, npk, npt, nb = bench.WORKLOADS[args.workload]
dev = torch.device("cuda", 0)
e1v, e3v, ib, zv = bench.build_inputs(npi, npj, npk, npt, nb, 0, npj, dev)
rows = npj * npk
n3 = npi * rows
alg = 4.0 * (1.0 + 1.0 / npt) * float(npi) * npj * (npk - 1) * npt
out = torch.empty((npt, npk, npj, nb), dtype=torch.float64, device=dev)
ref = torch.empty_like(out)
lines = []


def emit(d):
    d["frac_of_8TBs"] = alg / (d["mean_ms"] * 1e-3) / 8e12
    lines.append(d)
    print(json.dumps(d), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "a") as f:
            f.write(json.dumps(d) + "\n")


def timeit(plan, call, n=None):
    call()
    plan.profile_enable(True)
    for _ in range(n or args.launches):
        call()
    plan.profile_enable(False)
    e = plan.profile_read_each()
    return float(e.min()), float(e.mean()), float(e.max())


def measure(label, plan, call, extra=None, check=True):
    out.zero_()
    mn, mean, mx = timeit(plan, call)
    same = bool(torch.equal(out, ref)) if check else None
    emit({"layout": label, "min_ms": mn, "mean_ms": mean, "max_ms": mx, "same_bits_as_dense": same, "kernel": plan.describe(),
          **(extra or {})})


def pool_alloc(nbytes, move):
    """a fresh allocation; move > 0: after a spacer of an odd size that stays allocated meanwhile"""
    torch.cuda.empty_cache()
    sp = torch.empty(move * (1027 * 2**20 + 4096 * 13), dtype=torch.uint8, device=dev) if move else None
    p = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return p, sp


# ------------------------------------------------------------------ dense
plan0 = hip.MocPlan(npi, npj, npk, nb)
plan0.set_mesh(e1v, e3v, ib)
plan0.compute(zv, out=ref)
torch.cuda.synchronize()
measure("dense (the caller's torch tensor)", plan0, lambda: plan0.compute(zv, out=out), {"zv_ptr_mod_2MB": zv.data_ptr() % (2 << 20)})

# ------------------------------------------------------------------ stride / base sweeps, dense rows
MB2 = 2 << 20
tight = n3 * 4
up2mb = (tight + MB2 - 1) // MB2 * MB2
stride_cases = [("tight", tight, 0), ("2MB multiple", up2mb, 0), ("2MB multiple + 4 KB", up2mb + 4096, 0),
                ("2MB multiple + 64 KB", up2mb + 65536, 0), ("2MB multiple + 160 KB (2MB/13)", up2mb + 163840, 0),
                ("2MB multiple + 1 MB", up2mb + (1 << 20), 0), ("tight, base + 1 MB", tight, 1 << 20),
                ("2MB multiple + 8 KB + 128", up2mb + 8192 + 128, 0)]
if args.quick:
    stride_cases = stride_cases[:2]
for move in range(args.moves):
    for name, sbytes, base in stride_cases:
        pool, sp = pool_alloc(base + sbytes * npt + 4096, move)
        fl = pool[base:base + sbytes * npt].view(torch.float32)
        z = fl.as_strided((npt, npk, npj, npi), (sbytes // 4, npj * npi, npi, 1))
        z.copy_(zv)
        ptr = fl.data_ptr()
        measure("stride: " + name, plan0, lambda: plan0.compute_pitched(ptr, sbytes // 4, npi, npt, out),
                {"move": move, "rec_stride_bytes": sbytes, "base": base, "pool_ptr_mod_2MB": pool.data_ptr() % MB2})
        del z, fl, pool, sp

# ------------------------------------------------------------------ pitched layouts
pitches = [4352] if args.quick else [4324, 4352, 4416]
for P in pitches:
    if P < npi:
        continue
    planp = hip.MocPlan(npi, npj, npk, nb)
    planp.set_layout(P)
    planp.set_mesh(e1v, e3v, ib)
    for move in range(args.moves):
        # records outer, rows P apart
        pool, sp = pool_alloc(rows * P * 4 * npt + 4096, move)
        fl = pool[:rows * P * 4 * npt].view(torch.float32)
        fl.zero_()
        z = fl.as_strided((npt, npk, npj, npi), (rows * P, npj * P, P, 1))
        z.copy_(zv)
        ptr = fl.data_ptr()
        measure(f"pitch {P}: records outer", planp, lambda: planp.compute_pitched(ptr, rows * P, P, npt, out), {"move": move, "pitch": P})
        # rows outer: the records of a row next to each other
        fl.zero_()
        z = fl.as_strided((npt, npk, npj, npi), (P, npj * npt * P, npt * P, 1))
        z.copy_(zv)
        measure(f"pitch {P}: rows outer (a row's {npt} records contiguous)", planp, lambda: planp.compute_pitched(ptr, P, npt * P, npt, out),
                {"move": move, "pitch": P})
        del z, fl, pool, sp
    del planp
    # the row's weights in front of its records
    for move in range(args.moves):
        pool, sp = pool_alloc(rows * P * 4 * (npt + 1) + 4096, move)
        fl = pool[:rows * P * 4 * (npt + 1)].view(torch.float32)
        fl.zero_()
        planw = hip.MocPlan(npi, npj, npk, nb)
        planw.set_layout((npt + 1) * P, w_pool=fl)
        planw.set_mesh(e1v, e3v, ib)
        z = fl[P:].as_strided((npt, npk, npj, npi), (P, npj * (npt + 1) * P, (npt + 1) * P, 1))
        z.copy_(zv)
        ptr = fl.data_ptr() + 4 * P
        measure(f"pitch {P}: rows outer, weights in front (one run per row)", planw,
                lambda: planw.compute_pitched(ptr, P, (npt + 1) * P, npt, out), {"move": move, "pitch": P})
        if move == 0 and not args.quick:
            for tune in ((2, 9), (3, 6), (4, 5), (1, 12)):
                try:
                    planw.set_tuning(tune[0], tune[1], 0)
                    measure(f"pitch {P}: rows outer, weights in front, waves x vectors = {tune}", planw,
                            lambda: planw.compute_pitched(ptr, P, (npt + 1) * P, npt, out), {"move": move, "pitch": P})
                except Exception as exc:  # a shape the kernel refuses
                    print("skip", tune, exc)
            planw.set_tuning(0, 0, 0)
            os.environ["CDFT_MOC_XCDMAP"] = "0"
            measure(f"pitch {P}: rows outer, weights in front, no XCD bands", planw,
                    lambda: planw.compute_pitched(ptr, P, (npt + 1) * P, npt, out), {"move": move, "pitch": P})
            del os.environ["CDFT_MOC_XCDMAP"]
        del planw, z, fl, pool, sp

# dense again at the end: has the box drifted?
measure("dense (again, at the end)", plan0, lambda: plan0.compute(zv, out=out))
