"""cdfvtrp -bathy (src/cdfvtrp.f90:210-252): the two vertically integrated transports brought to the T point and rotated
onto the direction of the bathymetric slope -- cdft_vint_slope_rotate, a 2-D kernel (the program holds no arithmetic).

tests/golden/lift_cdfvtrp_bathy.npz holds what the reference's own statements hand to putvar under lbathy (the harness of
oracle/ref_lift.py: REAL(dtrpu), REAL(dtrpv), REAL(along), REAL(cross)).  Everything but three library calls is pointwise
r8 / r4 arithmetic in the reference's order, hence bit-exact; ATAN2, COS and SIN are single-precision LIBRARY results, and
three implementations meet here: the Fortran runtime behind the golden, the C library behind the oracle, and the GPU, which
forms them in r8 and rounds (the correctly rounded r4 value except on near-ties).  Stated tolerance, against the golden:
an ulp(r4) on the angle and on each of the two factors moves a component by at most 3 * 2^-23 * (|dwku| + |dwkv|); the
written floats then agree within that plus their own rounding.  The tests also count how many values are identical: the
oracle and the golden nearly everywhere (the same C library behind both), the correctly rounded angle on about nine points
in ten (this glibc's atan2f is within an ulp of the correctly rounded value, not always on it)."""
import os

import numpy as np
import pytest

import lift_cases as lc
import oracle_tools as ot

f32 = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lift_cdfvtrp_bathy.npz")


def case(name):
    g, t, s, u, v = lc.grid8f()
    kw = dict(e31d=g["e3t_1d"]) if name == "full" else None
    return g, u, v, kw, lc.vtrp_hdepw(g), g["tmask"][0].astype(f32)


def by(vint, rotate, name):
    """-> (npt,4,npj,npi) r4 laid out like the harness output, and the r8 bound |dwku| + |dwkv| per record"""
    g, u, v, kw, hdepw, tm = case(name)
    out = np.zeros((2, 4) + tm.shape, f32)
    bound = np.zeros((2,) + tm.shape)
    for k in range(2):
        du = vint(u[k], g["e2u"], **(kw or dict(e3=g["e3u_0"])))
        dv = vint(v[k], g["e1v"], **(kw or dict(e3=g["e3v_0"])))
        a, c = rotate(du, dv, hdepw, g["e1u"], g["e2v"], tm)
        out[k] = np.stack([x.astype(f32) for x in (du, dv, a, c)])
        wu, wv = np.zeros_like(du), np.zeros_like(dv)
        wu[:, 1:] = 0.5 * (np.abs(du[:, 1:]) + np.abs(du[:, :-1]))
        wu[:, 0] = wu[:, -2]
        wv[1:, :] = 0.5 * (np.abs(dv[1:, :]) + np.abs(dv[:-1, :]))
        bound[k] = wu + wv
    return out, bound


def check(got, bound, want, min_identical):
    assert np.array_equal(got[:, :2], want[:, :2])                      # the transports themselves: bit-exact
    tol = 3 * 2.0 ** -23 * bound[:, None] + np.spacing(np.abs(want[:, 2:]).astype(f32))
    assert np.all(np.abs(got[:, 2:].astype(np.float64) - want[:, 2:]) <= tol)
    same = np.mean(got[:, 2:] == want[:, 2:])
    assert same >= min_identical, same
    assert np.abs(want[:, 2:]).max() > 0


@pytest.mark.parametrize("name", ["default", "full"])
def test_oracle_vtrp_bathy_against_reference_statements(name):
    got, bound = by(ot.vint_transport, ot.vtrp_slope, name)
    check(got, bound, np.load(GOLD)[name], 0.99)


def test_oracle_vtrp_slope_matches_numpy_restatement():
    """the C restatement against the formulas written out in NumPy (r8 libm rounded to r4 for the three functions)"""
    g, u, v, kw, hdepw, tm = case("default")
    du = ot.vint_transport(u[0], g["e2u"], e3=g["e3u_0"])
    dv = ot.vint_transport(v[0], g["e1v"], e3=g["e3v_0"])
    wu, wv = np.zeros_like(du), np.zeros_like(dv)
    wu[:, 1:] = 0.5 * (du[:, 1:] + du[:, :-1])
    wu[:, 0] = wu[:, -2]
    wv[1:, :] = 0.5 * (dv[1:, :] + dv[:-1, :])
    dhdx, dhdy = np.zeros_like(tm), np.zeros_like(tm)
    dhdx[:, 1:-1] = (hdepw[:, 2:] - hdepw[:, :-2]) / (g["e1u"][:, 1:-1] + g["e1u"][:, :-2]) * tm[:, 1:-1]
    dhdy[1:-1, :] = (hdepw[2:, :] - hdepw[:-2, :]) / (g["e2v"][1:-1, :] + g["e2v"][:-2, :]) * tm[1:-1, :]
    al = (np.arctan2(dhdx.astype(np.float64), dhdy.astype(np.float64)).astype(f32) * tm).astype(f32)
    c, s = np.cos(al.astype(np.float64)).astype(f32).astype(np.float64), np.sin(al.astype(np.float64)).astype(f32).astype(np.float64)
    a, x = ot.vtrp_slope(du, dv, hdepw, g["e1u"], g["e2v"], tm)
    scale = np.abs(wu) + np.abs(wv)
    assert np.all(np.abs(a - (wu * c + wv * s) * tm) <= 3 * 2.0 ** -23 * scale)
    assert np.all(np.abs(x - (-wu * s + wv * c) * tm) <= 3 * 2.0 ** -23 * scale)
    # (identical on 9 points in 10 only: the C library's atan2f is within an ulp of the correctly rounded value, not on it)
    assert np.mean(a == (wu * c + wv * s) * tm) > 0.85


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "full"])
def test_hip_vtrp_bathy_against_reference_statements(gpu, name):
    got, bound = by(gpu.vint_transport, gpu.vint_slope_rotate, name)
    check(got, bound, np.load(GOLD)[name], 0.85)       # (the GPU's correctly rounded angle against a library's: see the NumPy test)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj", [(1, 1), (2, 2), (3, 5), (64, 4), (65, 5), (1442, 9)])
def test_hip_slope_rotate_shapes_and_resident_arrays(gpu, npi, npj):
    """plane sizes around the launch shape, host and HBM-resident operands: against the oracle under the stated tolerance,
    and the two memory spaces bit for bit against each other"""
    import torch
    rng = np.random.default_rng(npi * 100 + npj)
    du, dv = rng.normal(0, 1e6, (npj, npi)), rng.normal(0, 1e6, (npj, npi))
    hdepw = rng.uniform(0, 5000, (npj, npi)).astype(f32)
    e1u, e2v = rng.uniform(2e3, 3e4, (npj, npi)).astype(f32), rng.uniform(2e3, 3e4, (npj, npi)).astype(f32)
    tm = (rng.random((npj, npi)) > 0.3).astype(f32)
    a, c = gpu.vint_slope_rotate(du, dv, hdepw, e1u, e2v, tm)
    ra, rc = ot.vtrp_slope(du, dv, hdepw, e1u, e2v, tm)
    wu, wv = np.zeros_like(du), np.zeros_like(dv)
    wu[:, 1:] = 0.5 * (np.abs(du[:, 1:]) + np.abs(du[:, :-1]))
    if npi >= 2:
        wu[:, 0] = wu[:, -2]
    wv[1:, :] = 0.5 * (np.abs(dv[1:, :]) + np.abs(dv[:-1, :]))
    tol = 3 * 2.0 ** -23 * (wu + wv)
    assert np.all(np.abs(a - ra) <= tol) and np.all(np.abs(c - rc) <= tol)
    dev = [torch.from_numpy(x).cuda() for x in (du, dv, hdepw, e1u, e2v, tm)]
    da, dc = gpu.vint_slope_rotate(*dev)
    assert np.array_equal(da.cpu().numpy(), a) and np.array_equal(dc.cpu().numpy(), c)
    with pytest.raises(ValueError):
        gpu.vint_slope_rotate(du, dv, hdepw, e1u, e2v, torch.from_numpy(tm).cuda())      # mixed memory spaces
