#!/usr/bin/env python3
"""One plan, one set of records, one process: the bench workload timed with an environment hook of the zonal kernel off and
on, alternating (the kernel's time moves by 3 % with where the records were allocated, so two processes cannot be compared).
    python tools/ab_moc_env.py CDFT_MOC_XCDMAP [--rounds 4 --steps 30]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cdftools_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("var")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--workload", default="eorca12")
    ap.add_argument("--values", default="0,1", help="the values to alternate between")
    a = ap.parse_args()
    bench.hip = hip
    dev = torch.device("cuda", 0)
    npi, npj, npk, npt, nb = bench.WORKLOADS[a.workload]
    e1v, e3v, ib, zv = bench.build_inputs(npi, npj, npk, npt, nb, 0, npj, dev)
    plan = hip.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    out = torch.empty((npt, npk, npj, nb), dtype=torch.float64, device=dev)
    ref = None
    for r in range(a.rounds):
        for val in a.values.split(","):
            os.environ[a.var] = val
            for _ in range(3):
                plan.compute(zv, out=out)
            torch.cuda.synchronize()
            plan.profile_enable(True)
            for _ in range(a.steps):
                plan.compute(zv, out=out)
            torch.cuda.synchronize()
            ms = plan.profile_read_each()
            plan.profile_enable(False)
            same = None
            if ref is None:
                ref = out.clone()
            else:
                same = bool(torch.equal(out, ref))
            print(f"round {r} {a.var}={val}: mean {ms.mean():.4f} ms  min {ms.min():.4f}  same_bits {same}", flush=True)


if __name__ == "__main__":
    main()
