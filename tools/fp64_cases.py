#!/usr/bin/env python3
"""One fp64-bound kernel, a few launches, nothing else: the unit the counter passes of tools/fp64_pmc.sh profile
(SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 per dispatch).  The same fields as tools/bench_tools.py.
    python3 tools/fp64_cases.py --case sigma2 --grid eorca12 [--land noise|contiguous|none] [--reps 3]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_tools import GRIDS, eos_fields, mocsig_fields  # noqa: E402
from cdftools_amd import hip  # noqa: E402

CASES = ["sigma0", "sigma0_fma", "sigma0_teos10", "sigma2", "bn2", "mocsig"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True, choices=CASES)
    ap.add_argument("--grid", default="eorca12", choices=sorted(GRIDS))
    ap.add_argument("--land", default="none", choices=["none", "noise", "contiguous"])
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    npi, npj, npk = GRIDS[a.grid]
    dev = torch.device("cuda", 0)
    n3 = npi * npj * npk
    if a.case == "mocsig":
        f = mocsig_fields(npi, npj, npk, dev, a.land if a.land != "none" else "noise", smooth=True)
        for _ in range(a.reps):
            hip.mocsig(f["zv"], f["zt"], f["zs"], f["e1"], f["ib"], 158, 30.0, 0.05, pref=2000.0, e3v=f["e3"])
    else:
        t, s = eos_fields(npi, npj, npk, dev, a.land)
        o4 = torch.empty(n3, dtype=torch.float32, device=dev)
        for _ in range(a.reps):
            if a.case == "sigma0":
                hip.sig_tool(t, s, 0.0, 0.0, 0, 0, out=o4)
            elif a.case == "sigma0_fma":
                hip.sig_tool(t, s, 0.0, 0.0, 0, 1, out=o4)
            elif a.case == "sigma0_teos10":
                hip.sig_tool(t, s, 0.0, 0.0, 1, 0, out=o4)
            elif a.case == "sigma2":
                hip.sig_tool(t, s, 2000.0, 0.0, 0, 0, out=o4)
            else:
                e3 = 1 + 199 * torch.rand(n3, device=dev)
                hip.eosbn2(t, t + 0.5, s, s * 1.01, e3, 2000.0, out=o4)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
