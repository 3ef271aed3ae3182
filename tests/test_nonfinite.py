"""Non-finite field values (NaN / Inf used as land fill) in the masked reductions.

The reference applies every mask as an r4 MULTIPLY (src/cdfmoc.f90:367-380, src/cdfheatc.f90:218-240,
src/cdfmean.f90:511-530, src/cdfzonalmean.f90:538-560, src/cdfmhst.f90:300-345), so a NaN or an
Inf sitting on a masked point is not removed: 0*NaN = 0*Inf = NaN enters the sum.  A kernel that
SELECTS on the mask would silently return finite numbers where the reference returns NaN.  These
tests pin the reference's behaviour: the non-finite pattern of the result must be the oracle's, and
the finite entries must agree to the usual bound.
"""
import numpy as np
import pytest

import oracle_lib
import oracle_tools
import synth


def poison(a, landmask, seed, n_land=5, n_ocean=0):
    """Copy of `a` with NaN/+Inf/-Inf on `n_land` points where landmask is true and +Inf on
    `n_ocean` points where it is false."""
    a = a.copy()
    rng = np.random.default_rng(seed)
    flat = a.reshape(-1)
    land = np.flatnonzero(np.broadcast_to(landmask, a.shape).reshape(-1))
    sea = np.flatnonzero(~np.broadcast_to(landmask, a.shape).reshape(-1))
    vals = np.array([np.nan, np.inf, -np.inf], dtype=a.dtype)
    if land.size:
        pick = rng.choice(land, min(n_land, land.size), replace=False)
        flat[pick] = vals[np.arange(pick.size) % 3]
    if sea.size and n_ocean:
        flat[rng.choice(sea, min(n_ocean, sea.size), replace=False)] = np.inf
    return a


def same(got, ref, lim):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), (int(np.isnan(got).sum()), int(np.isnan(ref).sum()))
    inf = np.isinf(ref)
    assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf])
    ok = np.isfinite(ref)
    lim = np.broadcast_to(np.asarray(lim, dtype=np.float64), ref.shape)
    assert np.all(np.abs(got[ok] - ref[ok]) <= lim[ok])


def test_oracle_propagates_a_masked_nan_like_the_reference_expression():
    g = synth.make_grid(12, 5, 4, seed=3, nbasins=5)
    e1v, e3v, ib = synth.moc_inputs(g)
    zv = synth.velocity(g, 1, seed=1)
    j, i = np.argwhere(ib[:, :, 0] == 0)[0]
    zv[0, 1, j, i] = np.nan
    d = oracle_lib.moc(e1v, e3v, ib, zv)[0]                      # (npk, npj, nb)
    assert np.isnan(d[:2, j, :]).all()                           # level 1 and, through the cumulation, level 0
    assert np.isfinite(d[2:, j, :]).all()
    assert np.isfinite(np.delete(d, j, axis=1)).all()


# ----------------------------------------------------------------------------- GPU
def overflowing(zv, e1v, e3v, ib, seed, with_nan=False):
    """Copy of zv with huge FINITE values on a few ocean cells that lie in the global basin but outside at least one
    sub-basin: the r4 product (e1v*e3v)*zv overflows to +-Inf there.  The reference forms ((e1v*e3v)*ibmask)*zv, so the
    basins the cell is not in get (w*0)*zv = 0, not 0*Inf = NaN (cdfmoc.f90:373-374).  with_nan: a NaN on a cell outside
    every basin of the SAME rows as well, which must still reach every basin."""
    zv = zv.copy()
    rng = np.random.default_rng(seed)
    w = e1v[None] * e3v                                                  # (npk, npj, npi)
    nb = ib.shape[2]
    partial = (ib[:, :, 0] == 1) & (ib[:, :, 1:].sum(axis=2) < max(nb - 1, 1)) if nb > 1 else (ib[:, :, 0] == 1)
    cand = np.argwhere((w[:-1] > 1.0) & partial[None])
    pick = cand[rng.choice(len(cand), min(4, len(cand)), replace=False)]
    rows = []
    for n, (k, j, i) in enumerate(pick):
        zv[:, k, j, i] = np.float32(3.0e38) * (1 if n % 2 == 0 else -1)
        rows.append((k, j))
    if with_nan:
        nowhere = np.argwhere(ib.max(axis=2) == 0)
        for k, j in rows[:2]:
            on_row = nowhere[nowhere[:, 0] == j]
            if len(on_row):
                zv[1 % zv.shape[0], k, j, on_row[0, 1]] = np.nan
    return zv, rows


def test_oracle_gives_zero_not_nan_for_an_overflowed_product_outside_a_basin():
    g = synth.make_grid(24, 18, 7, seed=24, nbasins=5)
    e1v, e3v, ib = synth.moc_inputs(g)
    zv, rows = overflowing(synth.velocity(g, 2, seed=4), e1v, e3v, ib, seed=1)
    d = oracle_lib.moc(e1v, e3v, ib, zv)                         # (npt, npk, npj, nb)
    assert rows and np.isinf(d[..., 0]).any() and not np.isnan(d).any()
    # at the poisoned level itself (before the cumulation spreads it upwards... the cumulation runs upwards from the
    # bottom, so the levels BELOW k never see it): a basin the cell is not in keeps a finite value on that row
    checked = 0
    for k, j in rows:
        i = np.flatnonzero(np.abs(zv[0, k, j]) > 1e38)
        for b in range(5):
            if ib[j, i, b].sum() == 0:
                assert np.isfinite(d[:, :, j, b]).all()
                checked += 1
            else:
                assert np.isinf(d[:, k, j, b]).all()
    assert checked


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 5), (182, 149, 31, 5), (259, 6, 4, 8), (67, 5, 3, 1), (1442, 9, 4, 5)])
def test_moc_finite_overflow_is_not_a_nan_outside_the_basin(gpu, npi, npj, npk, nb, mode):
    """the zero term is (w*0)*zv, keyed on the velocity, not 0*(w*zv)."""
    g = synth.make_grid(npi, npj, npk, seed=npi, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g)
    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    plan.set_tuning(mode=mode)
    for with_nan in (False, True):
        zv, rows = overflowing(synth.velocity(g, 3, seed=4), e1v, e3v, ib, seed=npi, with_nan=with_nan)
        assert rows
        ref = oracle_lib.moc(e1v, e3v, ib, zv)
        clean = np.where(np.abs(zv) < 1e38, zv, 0).astype(np.float32)
        clean[~np.isfinite(clean)] = 0
        lim = 1e-13 * np.abs(oracle_lib.moc(e1v, e3v, ib, np.abs(clean))).max() + 1e-300
        same(plan.compute(zv), ref, lim)
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 5), (182, 149, 31, 5), (259, 6, 4, 8), (67, 5, 3, 1)])
def test_moc(gpu, npi, npj, npk, nb, mode):
    g = synth.make_grid(npi, npj, npk, seed=npi, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g)
    zv = synth.velocity(g, 3, seed=4)
    outside_every_basin = (ib.max(axis=2) == 0)[None, None]
    outside_global = (ib[:, :, 0] == 0)[None, None]
    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    plan.set_tuning(mode=mode)
    for land, n_ocean in ((outside_every_basin, 0), (outside_global, 0), (outside_global, 2)):
        bad = poison(zv, land, seed=npi + n_ocean, n_land=6, n_ocean=n_ocean)
        ref = oracle_lib.moc(e1v, e3v, ib, bad)
        clean = np.where(np.isfinite(bad), bad, 0).astype(np.float32)
        bound = oracle_lib.moc(e1v, e3v, ib, np.abs(clean))
        lim = 1e-13 * np.abs(bound).max() + 1e-300
        same(plan.compute(bad), ref, lim)
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb", [(1442, 9, 4, 5), (259, 6, 4, 5), (1190, 5, 3, 3)])
def test_moc_row_segments(gpu, npi, npj, npk, nb):
    """The class kernel with its rows cut into segments (one vector per lane, one wave per segment: 6 / 2 / 5 of them): a
    segment settles its own zero terms -- a NaN/Inf velocity outside a basin in ANY segment reaches the basin, an overflowed
    product does not."""
    g = synth.make_grid(npi, npj, npk, seed=npi, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g)
    zv = synth.velocity(g, 3, seed=4)
    outside_every_basin = (ib.max(axis=2) == 0)[None, None]
    outside_global = (ib[:, :, 0] == 0)[None, None]
    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    plan.set_tuning(1, 1, 3)
    for land, n_ocean in ((outside_every_basin, 0), (outside_global, 0), (outside_global, 2)):
        bad = poison(zv, land, seed=npi + n_ocean, n_land=6, n_ocean=n_ocean)
        ref = oracle_lib.moc(e1v, e3v, ib, bad)
        clean = np.where(np.isfinite(bad), bad, 0).astype(np.float32)
        lim = 1e-13 * np.abs(oracle_lib.moc(e1v, e3v, ib, np.abs(clean))).max() + 1e-300
        same(plan.compute(bad), ref, lim)
    assert "row_segments" in plan.describe()
    for with_nan in (False, True):
        big, rows = overflowing(zv, e1v, e3v, ib, seed=npi, with_nan=with_nan)
        assert rows
        ref = oracle_lib.moc(e1v, e3v, ib, big)
        clean = np.where(np.abs(big) < 1e38, big, 0).astype(np.float32)
        clean[~np.isfinite(clean)] = 0
        lim = 1e-13 * np.abs(oracle_lib.moc(e1v, e3v, ib, np.abs(clean))).max() + 1e-300
        same(plan.compute(big), ref, lim)
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk", [(24, 18, 7), (182, 149, 5), (1031, 7, 3)])
def test_heatc(gpu, npi, npj, npk):
    g = synth.make_grid(npi, npj, npk, seed=npi)
    rng = np.random.default_rng(1)
    temp = (-2.0 + 32.0 * rng.random((npk, npj, npi), dtype=np.float32)).astype(np.float32)
    plan = gpu.HeatcPlan(npi, npj, npk)
    plan.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e3t=g["e3t_0"])
    for n_ocean in (0, 1):
        bad = poison(temp, g["tmask"] == 0, seed=5 + n_ocean, n_land=4, n_ocean=n_ocean)
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], bad, e3t=g["e3t_0"])
        clean = np.where(np.isfinite(bad), bad, 0).astype(np.float32)
        bound = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], np.abs(clean), e3t=g["e3t_0"])
        same(plan.compute(bad), ref, 1e-13 * bound + 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 4), (182, 149, 5, 1), (1023, 7, 3, 3)])
def test_mean(gpu, npi, npj, npk, nb):
    from test_cli_next import mean_inputs

    g, zv, ib = mean_inputs(npi, npj, npk, nb, seed=npi + 1)
    tm = g["tmask"].astype(np.float32)
    plan = gpu.MeanPlan(npi, npj, npk, nb)
    plan.set_mesh(g["e1t"], g["e2t"], tm, ib, e3=g["e3t_0"])
    outside = (ib.max(axis=2) == 0)[None] | (tm == 0)
    for land, n_ocean in ((tm == 0, 0), (outside, 0), (outside, 1)):
        bad = poison(zv, land, seed=npj + n_ocean, n_land=4, n_ocean=n_ocean)
        ref = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, bad, e3=g["e3t_0"])
        clean = np.where(np.isfinite(bad), bad, 0).astype(np.float32)
        bound = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, np.abs(clean), e3=g["e3t_0"])
        same(plan.compute(bad), ref, 1e-13 * bound + 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["class", "fast", "general"])
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 5), (182, 149, 5, 5), (1031, 4, 3, 1), (1600, 3, 2, 5)])
def test_zonal(gpu, monkeypatch, npi, npj, npk, nb, kernel):
    from test_cli_next import zonal_inputs

    if kernel != "class":
        monkeypatch.setenv("CDFT_ZONAL_GENERAL" if kernel == "general" else "CDFT_ZONAL_NOCLASS", "1")
    g, zv, zm = zonal_inputs(npi, npj, npk, nb, seed=npi + 9)
    tm = g["tmask"].astype(np.float32)
    plan = gpu.ZonalPlan(npi, npj, npk, nb)
    plan.set_mesh(g["e1t"], g["e2t"], tm, zm)
    for n_ocean in (0, 1):
        bad = poison(zv, tm == 0, seed=npk + n_ocean, n_land=4, n_ocean=n_ocean)
        clean = np.where(np.isfinite(bad), bad, 0).astype(np.float32)
        for mean_form in (False, True):
            ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, bad, mean_form=mean_form)
            bound = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, np.abs(clean), mean_form=mean_form)["sum"]
            got = plan.compute(bad, mean_form=mean_form, minmax=False)
            same(got["sum"], ref["sum"], 1e-13 * bound + 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 4), (182, 149, 5, 4), (333, 5, 4, 8)])
def test_mhst(gpu, npi, npj, npk, nb):
    from test_transport import mhst_masks, tracer_fields

    g = synth.make_grid(npi, npj, npk, seed=npi + 1)
    t, s, _, v = tracer_fields(g, 9)
    m = mhst_masks(g, nb)
    vt = ((np.float32(0.5) * (t + np.roll(t, -1, axis=1))).astype(np.float32) * v).astype(np.float32)
    vs = ((np.float32(0.5) * (s + np.roll(s, -1, axis=1))).astype(np.float32) * v).astype(np.float32)
    plan = gpu.MhstPlan(npi, npj, npk, nb)
    plan.set_mesh(g["e1v"], m, e3v=g["e3v_0"])
    mm = np.asarray(m)
    land = (mm.reshape(-1, npj, npi).max(axis=0) == 0)[None] if mm.ndim == 3 else (mm == 0)[None]
    for n_ocean in (0, 1):
        bvt = poison(vt, land, seed=3 + n_ocean, n_land=4, n_ocean=n_ocean)
        bvs = poison(vs, land, seed=4 + n_ocean, n_land=4, n_ocean=0)
        ref = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], vt=bvt, vs=bvs)
        cvt, cvs = (np.where(np.isfinite(x), x, 0).astype(np.float32) for x in (bvt, bvs))
        bound = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], vt=np.abs(cvt), vs=np.abs(cvs))
        got = plan.compute(vt=bvt, vs=bvs)
        for k in (0, 1):
            same(got[k], ref[k], 1e-13 * np.abs(bound[k]).max() + 1e-300)
        bv = poison(v, land, seed=8 + n_ocean, n_land=4, n_ocean=n_ocean)
        ref = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], zv=bv, zt=t, zs=s)
        cv = np.where(np.isfinite(bv), bv, 0).astype(np.float32)
        bound = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], zv=np.abs(cv), zt=np.abs(t), zs=np.abs(s))
        got = plan.compute(zv=bv, zt=t, zs=s)
        for k in (0, 1):
            same(got[k], ref[k], 1e-13 * np.abs(bound[k]).max() + 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 5), (130, 40, 4, 8), (31, 11, 5, 1)])
def test_moc_decomp(gpu, npi, npj, npk, nb):
    """cdfmoc -decomp: a NaN velocity on a point outside every basin reaches the total, the ageostrophic
    remainder and -- where the column has a depth -- the barotropic part through 0 * (its depth mean)
    (src/cdfmoc.f90:403-419), exactly as in the reference's expressions."""
    from test_moc_decomp import decomp_inputs

    a = decomp_inputs(npi, npj, npk, nb, seed=npi + 7)
    land = (a["ibmask"].max(axis=2) == 0)[None]
    a["zv"] = poison(a["zv"], land, seed=2, n_land=3)
    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(a["e1v"], a["e3v"], a["ibmask"])
    plan.decomp_set_mesh(a["e3v"], a["e1u"], a["zcoef"], a["gdept"], a["tmask"], a["umask"])
    ref = oracle_lib.moc_decomp(**a)
    got = plan.decomp(a["zv"], a["ztemp"], a["zsal"])
    assert np.isnan(ref["total"]).any()
    for k in ref:
        fin = np.isfinite(ref[k])
        scale = np.abs(ref[k][fin]).max() if fin.any() else 1.0
        same(got[k], ref[k], 1e-9 * scale + 1e-300)
    plan.close()
