#!/usr/bin/env python3
"""Sweep the zonal-kernel shape (waves per row, float4 vectors per lane, accumulate mode) on one
GPU with the bench workload resident in HBM; prints one line per variant (interleaved rounds, one
process: cdna_hip_programming.md rule 24)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cdftools_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="eorca12")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", default="")
    ap.add_argument("--npt", type=int, default=0)
    ap.add_argument("--nb", type=int, default=0, help="number of basins (default: the workload's)")
    ap.add_argument("--staged", action="store_true", help="the records staged into the plan's pool (what bench.py and the program's routes read)")
    args = ap.parse_args()
    npi, npj, npk, npt, nb = bench.WORKLOADS[args.workload]
    if args.npt:
        npt = args.npt
    if args.nb:
        nb = args.nb
    dev = torch.device("cuda", 0)
    e1v, e3v, ib, zv = bench.build_inputs(npi, npj, npk, npt, nb, 0, npj, dev)
    plan = hip.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    del e3v
    torch.cuda.empty_cache()
    out = torch.empty((npt, npk, npj, nb), dtype=torch.float64, device=dev)
    if args.staged:
        plan.stage(zv)
    run = (lambda: plan.compute_staged(out)) if args.staged else (lambda: plan.compute(zv, out=out))
    if args.variants:
        variants = [tuple(int(x) for x in v.split(",")) for v in args.variants.split(";")]
    else:
        nvec = (npi + 6) // 4
        variants = []
        for mode in (1, 2, 3):
            for nv in {1: (1, 2, 3, 4, 5, 6, 8, 9, 12, 16), 2: (1, 2, 3, 4), 3: (1, 2, 3, 4, 5, 6, 8, 9, 12)}[mode]:
                k = -(-nvec // (64 * nv))
                if k <= 16 and not (k > 4 and (nv > 6 or mode == 2)):
                    variants.append((k, nv, mode))
    alg = 4.0 * (1 + 1.0 / npt) * npi * npj * (npk - 1) * npt
    best = {}
    ref = None
    for rnd in range(args.rounds):
        for v in variants:
            try:
                plan.set_tuning(*v)
                run()                              # warm
            except Exception as e:                 # a shape this row length or mode does not allow
                if rnd == 0:
                    print(json.dumps({"variant": v, "refused": str(e)}), flush=True)
                continue
            plan.profile_enable(True)
            for _ in range(5 if args.staged else 3):
                run()
            plan.profile_enable(False)
            ms, n = plan.profile_read()
            ms /= n
            if ref is None:
                ref = out.clone()
            maxdiff = float((out - ref).abs().max())
            best.setdefault(v, []).append(ms)
            print(json.dumps({"round": rnd, "variant": v, "kernel": plan.describe(), "ms": round(ms, 4),
                              "GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4),
                              "maxdiff_vs_first": maxdiff}), flush=True)
    print("SUMMARY (min ms over rounds)")
    for v, t in sorted(best.items(), key=lambda kv: min(kv[1])):
        print(v, "min %.4f ms  median %.4f ms  %.1f GB/s" % (min(t), sorted(t)[len(t) // 2], alg / min(t) / 1e6))


if __name__ == "__main__":
    main()
