#!/usr/bin/env python3
"""Does the PLACEMENT of the caller's arrays move the cdfmoc kernel's time?  One process, one plan, the bench workload:
the 56 GB of velocities are cloned to a new allocation several times (spacers of odd sizes in between), then the plan --
and with it the library's own 4.7 GB of weights -- is built again three times; each state is timed over 12 launches
(HIP events on the launch stream, min and mean).  Run on the GPU box: python tools/placement_probe.py
(profiles/r02/placement_probe.txt: 9.56 .. 9.88 ms for the same kernel on the same data, repeatable to 0.03 ms inside a
state: the run-to-run spread of bench.py on one box is where the allocator put the records, not the kernel.)"""
import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from cdftools_amd import hip
npi, npj, npk, npt, nb = bench.WORKLOADS["eorca12"]
dev = torch.device("cuda", 0)
e1v, e3v, ib, zv = bench.build_inputs(npi, npj, npk, npt, nb, 0, npj, dev)
def timeit(plan, z, out, n=12):
    plan.compute(z, out=out)
    plan.profile_enable(True)
    for _ in range(n): plan.compute(z, out=out)
    plan.profile_enable(False)
    e = plan.profile_read_each()
    return float(e.min()), float(e.mean())
out = torch.empty((npt, npk, npj, nb), dtype=torch.float64, device=dev)
res = []
plan = hip.MocPlan(npi, npj, npk, nb); plan.set_mesh(e1v, e3v, ib)
res.append(("first", timeit(plan, zv, out)))
# new placement of zv: allocate a spacer of odd size, clone, free the old
for i, spacer_mb in enumerate((0, 1, 3, 1027, 4099)):
    sp = torch.empty(spacer_mb * 2**20 + 4096 * i, dtype=torch.uint8, device=dev) if spacer_mb else None
    z2 = zv.clone()
    del zv; torch.cuda.empty_cache()
    zv = z2
    res.append((f"zv moved (spacer {spacer_mb} MB) ptr%2MB={zv.data_ptr() % (2<<20)}", timeit(plan, zv, out)))
    del sp
# new plan (new placement of the weights)
for i in range(3):
    sp = torch.empty((i + 1) * 1000003, dtype=torch.uint8, device=dev)
    p2 = hip.MocPlan(npi, npj, npk, nb); p2.set_mesh(e1v, e3v, ib)
    res.append((f"new plan {i}", timeit(p2, zv, out)))
    del p2, sp
for r in res: print(r)
