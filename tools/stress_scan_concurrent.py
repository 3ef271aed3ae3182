#!/usr/bin/env python3
"""heatc_scan_kernel waits on its own earlier tickets; this runs it on one stream while cdfmoc's zonal kernel keeps every
CU busy on another (two host threads), for a few seconds, and checks that the sums never change and nothing hangs.
Run on the GPU box:  timeout 120 python tools/stress_scan_concurrent.py"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdftools_amd import hip  # noqa: E402

npi, npj, npk = 1442, 1207, 75
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(s, device=dev, generator=g)  # noqa: E731
e1, e2 = 2000 + 26000 * rnd(npj, npi), 2000 + 26000 * rnd(npj, npi)
e3 = 1 + 199 * rnd(npk, npj, npi)
mask = (rnd(npk, npj, npi) > 0.35).to(torch.uint8)
temp = -2 + 32 * rnd(npk, npj, npi)
hp = hip.HeatcPlan(npi, npj, npk)
hp.set_mesh(e1, e2, mask, e3t=e3)
hp.set_sum_mode(True)
ref = hp.compute(temp).copy()

mp = hip.MocPlan(npi, npj, npk, 5)
ib = (rnd(npj, npi, 5) > 0.5).to(torch.int16)
mp.set_mesh(e1, e3, ib)
zv = rnd(4, npk, npj, npi) - 0.5
stop = time.time() + float(os.environ.get("STRESS_SECONDS", "6"))
count = {"heatc": 0, "moc": 0, "bad": 0}


def run_heatc():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while time.time() < stop:
            got = hp.compute(temp)
            count["heatc"] += 1
            if not np.array_equal(got.view(np.uint64), ref.view(np.uint64)):
                count["bad"] += 1


def run_moc():
    s = torch.cuda.Stream()
    out = torch.empty((4, npk, npj, 5), dtype=torch.float64, device=dev)
    with torch.cuda.stream(s):
        while time.time() < stop:
            mp.compute(zv, out=out)
            count["moc"] += 1
        s.synchronize()


ts = [threading.Thread(target=run_heatc), threading.Thread(target=run_moc)]
[t.start() for t in ts]
[t.join() for t in ts]
torch.cuda.synchronize()
print(count)
assert count["bad"] == 0 and count["heatc"] > 10 and count["moc"] > 10
print("ok")
