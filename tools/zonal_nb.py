import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cdftools_amd import hip
import bench_tools
grid = sys.argv[1] if len(sys.argv) > 1 else "eorca025"
npi, npj, npk = bench_tools.GRIDS[grid]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *sh: torch.rand(sh, device=dev, generator=g)
e1, e2 = (2000 + 26000 * rnd(npj, npi)).cpu().numpy(), (2000 + 26000 * rnd(npj, npi)).cpu().numpy()
tm = (rnd(npk, npj, npi) > 0.35).float().cpu().numpy()
zv = -2 + 32 * rnd(npk, npj, npi)
sea = torch.from_numpy(tm[0]).to(dev) > 0
band = ((torch.arange(npi, device=dev) * 3) // npi)[None, :]
for nb in (1, 2, 5):
    masks = [sea, sea & (band == 0), sea & (band > 0), sea & (band == 2), sea & (band == 1)][:nb]
    zm = torch.stack(masks, dim=2).float().cpu().numpy()
    p = hip.ZonalPlan(npi, npj, npk, nb)
    p.set_mesh(e1, e2, tm, zm)
    for mf in (False, True):
        for env in ({}, {"CDFT_ZONAL_NOCLASS": "1"}):
            os.environ.update(env)
            ms = bench_tools.timed(lambda: p.compute(zv, mean_form=mf, device_out=True), 20)
            for k in env: os.environ.pop(k)
            print(json.dumps({"grid": grid, "basins": nb, "mean_form": mf, "class_kernel": not env, "ms": round(ms, 4), "frac": round(npi*npj*npk*5.0/(ms*1e-3)/8e12, 3)}), flush=True)
    del p
