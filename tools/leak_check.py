#!/usr/bin/env python3
"""Does a plan give back what it took?  Each plan kind of the library is created, given a mesh, run and destroyed 200
times (sizes that make every optional buffer appear: row segments, sharded blocks, reference-order sums, -decomp work
arrays, the stateless entries' staging buffers); the free device memory before and after must agree to 64 MiB.
Run on the GPU box:  python tools/leak_check.py"""
import gc
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from cdftools_amd import hip  # noqa: E402
from test_stencil import frames  # noqa: E402
from test_transport import mhst_masks, tracer_fields  # noqa: E402

f32 = np.float32


def free_mib():
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0] / 2**20


npi, npj, npk = 9300, 12, 4                      # rows long enough for cdfmoc's row segments
g = synth.make_grid(npi, npj, npk, seed=1)
e1v, e3v, ib = synth.moc_inputs(g)
zv = synth.velocity(g, 2, seed=2)
t, s, u, v = tracer_fields(g, 3)
fr = frames(g, 2, seed=4)
m4 = mhst_masks(g, 4)
temp = (-2 + 32 * np.random.default_rng(5).random((npk, npj, npi))).astype(f32)
tm8 = g["tmask"].astype(np.uint8)


def moc(devices=None):
    p = hip.MocPlan(npi, npj, npk, 5, devices=devices)
    p.set_mesh(e1v, e3v, ib)
    p.compute(zv)
    p.stage(zv)                                   # the pool of staged records (round 4) goes with the plan too
    p.compute_staged(np.zeros((2, npk, npj, 5)))
    p.close()


def heatc(devices=None, r4seq=True):
    p = hip.HeatcPlan(npi, npj, npk, devices=devices)
    p.set_mesh(g["e1t"], g["e2t"], tm8, e3t=g["e3t_0"])
    p.set_sum_mode(r4seq)
    p.compute(temp)
    p.close()


def mhst(devices=None):
    p = hip.MhstPlan(npi, npj, npk, 4, devices=devices)
    p.set_mesh(g["e1v"], m4, e3v=g["e3v_0"])
    p.compute(zv=v, zt=t, zs=s)
    p.close()


def psi(devices=None):
    p = hip.PsiPlan(npi, npj, npk, devices=devices)
    p.compute(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], zv=v, e1v=g["e1v"], e3v=g["e3v_0"])
    p.close()


def curl():
    p = hip.CurlPlan(npi, npj)
    p.set_mesh(g["e1u"], g["e2v"], g["e1f"], g["e2f"])
    p.compute(u, v)
    p.close()


def mean_zonal():
    ibm = np.ones((npj, npi, 1), np.int16)
    p = hip.MeanPlan(npi, npj, npk, 1)
    p.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(f32), ibm, e3=g["e3t_0"])
    p.compute(temp)
    p.close()
    z = hip.ZonalPlan(npi, npj, npk, 1)
    z.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(f32), np.ones((npj, npi, 1), f32))
    z.compute(temp, mean_form=True, minmax=True)
    z.close()


def stateless():
    hip.vt_compute(*fr[:4])
    hip.vt_compute(*fr[:4], devices=[0, 0])
    hip.sigmai(t, s, 2000.0)
    hip.sig_tool(t, s, 1000.0)
    hip.alpha(t, s, 100.0)
    hip.mocsig(v, t, s, e1v, ib, 100, 20.0, 0.1, e3v=e3v)
    hip.vint_transport(v, g["e1v"], e3=g["e3v_0"])


CASES = [("cdfmoc", moc), ("cdfmoc, 3 row blocks", lambda: moc([0, 0, 0])), ("cdfheatc r4seq", heatc), ("cdfheatc r8", lambda: heatc(None, False)),
         ("cdfheatc, 2 row blocks", lambda: heatc([0, 0])), ("cdfmhst", mhst), ("cdfmhst, 2 row blocks", lambda: mhst([0, 0])), ("cdfpsi", psi),
         ("cdfpsi, 3 row blocks", lambda: psi([0, 0, 0])), ("cdfcurl", curl), ("cdfmean + cdfzonal", mean_zonal), ("stateless entries", stateless)]
bad = 0
for name, fn in CASES:
    fn()                                          # first use: lazily created streams / staging buffers of the library
    before = free_mib()
    for _ in range(200):
        fn()
    after = free_mib()
    print(f"{name:28s} free before {before:10.1f} MiB  after {after:10.1f} MiB  delta {before - after:8.1f} MiB")
    bad += (before - after) > 64
print("leak_check:", "FAILED" if bad else "every plan gives back what it took")
sys.exit(1 if bad else 0)
