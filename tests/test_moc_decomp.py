"""cdfmoc -decomp (src/cdfmoc.f90:353,:360-365,:390-517): total, geostrophic-shear, barotropic and
ageostrophic overturning of one record.

  CPU : the C oracle against an independent NumPy restatement, including the two reference quirks it
        keeps (dmoc_sh accumulates across records; the border of the geostrophic velocity).
  GPU : cdft_moc_decomp_* through the C ABI against the oracle.  Per-cell quantities are formed with
        the reference's exact r4/r8 mixing; the zonal sums regroup r8 additions, so the stated
        tolerance is, element by element, 1e-13 of the sum of the ABSOLUTE values of the terms that
        enter it (decomp_bounds: the form every other reduction of this repo is held to; an r4
        slip in a term would be ~1e-7 of it).
"""
import numpy as np
import pytest

import oracle_lib
import synth

f32 = np.float32


def decomp_inputs(npi, npj, npk, nb, seed):
    g = synth.make_grid(npi, npj, npk, seed=seed, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g)
    rng = np.random.default_rng(seed + 1)
    zv = ((rng.random((npk, npj, npi), dtype=f32) - f32(0.5)) * g["vmask"]).astype(f32)
    t = ((f32(-2) + f32(30) * rng.random((npk, npj, npi), dtype=f32)) * g["tmask"]).astype(f32)
    s = ((f32(33) + f32(4) * rng.random((npk, npj, npi), dtype=f32)) * g["tmask"]).astype(f32)
    rpi = f32(np.arccos(f32(-1.0)))
    f = (f32(2 * 2) * rpi / f32(24.0 * 3600.0) * np.sin(rpi * g["gphiv"] / f32(180.0), dtype=f32)).astype(f32)   # :428
    zcoef = np.where(f != 0, f32(-9.81) / f32(1025.0) / np.where(f != 0, f, f32(1)), f32(0)).astype(f32)        # :429-433
    return dict(e1v=e1v, e3v=e3v, ibmask=ib, zv=zv, ztemp=t, zsal=s, tmask=g["tmask"].astype(np.uint8),
                umask=g["umask"].astype(np.uint8), e1u=g["e1u"], zcoef=zcoef, gdept=g["gdept_1d"].astype(f32))


def numpy_decomp(a, teos10=False, dmoc_sh=None):
    """Independent restatement, array-at-a-time; zonal sums with np.sum (pairwise), so only close."""
    e1v, e3v, ib, zv3 = a["e1v"], a["e3v"], a["ibmask"], a["zv"]
    npk, npj, npi = e3v.shape
    nb = ib.shape[-1]
    w = (e1v[None] * e3v).astype(f32)                                        # (npk,npj,npi)
    wb = (w[..., None] * ib[None].astype(f32)).astype(f32)                   # (npk,npj,npi,nb)
    d = {}
    tot = -(wb * zv3[..., None]).astype(f32).astype(np.float64).sum(axis=2)
    tot[npk - 1] = 0

    def cum(raw, div):
        out = np.zeros_like(raw)
        for k in range(npk - 2, -1, -1):
            out[k] = out[k + 1] + raw[k] / div
        return out
    d["total"] = cum(tot, 1e6)
    dvbt, hdep = np.zeros((npj, npi)), np.zeros((npj, npi), f32)
    for k in range(npk - 1):
        dvbt += (e3v[k] * zv3[k]).astype(np.float64)
        hdep = (hdep + e3v[k]).astype(f32)
    dvbt = np.where(hdep != 0, dvbt / np.where(hdep != 0, hdep, 1).astype(np.float64), 0.0)
    raw_bt = -(wb.astype(np.float64) * dvbt[None, :, :, None]).sum(axis=2)
    raw_bt[npk - 1] = 0
    d["bt"] = cum(raw_bt, 1e6)
    dvgeo = np.zeros((npj, npi))
    dvbt2 = np.zeros((npj, npi))
    zv = zv3[npk - 2].copy()
    sh = np.zeros((npk, npj, nb)) if dmoc_sh is None else np.array(dmoc_sh, np.float64)
    um = a["umask"].astype(f32)
    for k in range(npk - 2, -1, -1):
        sig = oracle_lib.sigmai(a["ztemp"][k], a["zsal"][k], float(a["gdept"][k]), teos10)
        z0 = (sig * a["tmask"][k].astype(np.float64)).astype(f32)
        J, I = slice(1, npj - 1), slice(1, npi - 1)
        Jn, Im, Ip = slice(2, npj), slice(0, npi - 2), slice(2, npi)
        isum = np.maximum(1, a["umask"][k][Jn, Im].astype(int) + a["umask"][k][Jn, I] + a["umask"][k][J, Im] + a["umask"][k][J, I])
        zmsv = (f32(1) / isum.astype(f32)).astype(f32)
        g = ((z0[Jn, I] - z0[Jn, Im]) * um[k][Jn, Im] / a["e1u"][Jn, Im]).astype(f32)
        g = (g + ((z0[Jn, Ip] - z0[Jn, I]) * um[k][Jn, I] / a["e1u"][Jn, I]).astype(f32)).astype(f32)
        g = (g + ((z0[J, I] - z0[J, Im]) * um[k][J, Im] / a["e1u"][J, Im]).astype(f32)).astype(f32)
        g = (g + ((z0[J, Ip] - z0[J, I]) * um[k][J, I] / a["e1u"][J, I]).astype(f32)).astype(f32)
        inc = a["zcoef"][J, I].astype(np.float64) * g.astype(np.float64) * zmsv.astype(np.float64)
        inc = inc * ib[J, I, 0].astype(np.float64) * e3v[k][J, I].astype(np.float64)
        new = dvgeo[J, I] + inc
        zv[J, I] = (0.5 * (new + dvgeo[J, I])).astype(f32)
        dvgeo[J, I] = new
        dvbt2 += (e3v[k] * zv).astype(np.float64)
        sh[k] += -(wb[k] * zv[..., None]).astype(f32).astype(np.float64).sum(axis=1)
    dvbt2 = np.where(hdep != 0, dvbt2 / np.where(hdep != 0, hdep, 1).astype(np.float64), 0.0)
    raw_btw = -(wb.astype(np.float64) * dvbt2[None, :, :, None]).sum(axis=2)
    raw_btw[npk - 1] = 0
    sh = sh - raw_btw
    out = sh.copy()
    for k in range(npk - 2, -1, -1):
        out[k] = out[k + 1] + sh[k] / 1e6
    d["sh"] = out
    d["ag"] = d["total"] - d["sh"] - d["bt"]
    return d


def close(got, ref, rel=1e-10):
    for k in ("total", "sh", "bt", "ag"):
        tol = rel * (np.abs(ref[k]).max() + 1e-300)
        assert np.all(np.abs(got[k] - ref[k]) <= tol), (k, float(np.abs(got[k] - ref[k]).max()), tol)


def decomp_bounds(a, teos10=False, dmoc_sh=None, prior_bound=None):
    """For every element of every component, the sum of the absolute values of the terms the reference adds up to get it
    (src/cdfmoc.f90:437-517): sum over i and over the levels below of |w * v| / 1e6, v being the velocity, the
    barotropic velocity, the geostrophic velocity and its own barotropic part; the shear part of the record before where
    it is accumulated onto (:471) -- or `prior_bound`, that part's own bound, when the two sides of a comparison each
    accumulated onto their own; ag = total - sh - bt: the three together.  An error bound's scale, not a result: formed
    array-at-a-time from the same per-cell quantities as numpy_decomp."""
    e1v, e3v, ib, zv3 = a["e1v"], a["e3v"], a["ibmask"], a["zv"]
    npk, npj, npi = e3v.shape
    w = (e1v[None] * e3v).astype(f32)
    wb = np.abs((w[..., None] * ib[None].astype(f32)).astype(np.float64))    # (npk,npj,npi,nb)

    def cum(raw):
        out = np.zeros_like(raw)
        for k in range(npk - 2, -1, -1):
            out[k] = out[k + 1] + raw[k] / 1e6
        return out
    b = {}
    raw = (wb * np.abs(zv3.astype(np.float64))[..., None]).sum(axis=2)
    raw[npk - 1] = 0
    b["total"] = cum(raw)
    hdep = e3v[:npk - 1].astype(np.float64).sum(axis=0)
    col = np.abs(e3v[:npk - 1].astype(np.float64) * zv3[:npk - 1]).sum(axis=0)          # sum |e3v * v| over the column
    dvbt = np.where(hdep != 0, col / np.where(hdep != 0, hdep, 1), 0.0)
    raw = (wb * dvbt[None, :, :, None]).sum(axis=2)
    raw[npk - 1] = 0
    b["bt"] = cum(raw)
    # the geostrophic velocity: |increments| accumulated from the bottom, as the reference accumulates the signed ones
    dvgeo = np.zeros((npj, npi))
    zvg = np.zeros((npk, npj, npi))
    um = a["umask"].astype(f32)
    J, I = slice(1, npj - 1), slice(1, npi - 1)
    Jn, Im, Ip = slice(2, npj), slice(0, npi - 2), slice(2, npi)
    zvg[npk - 2] = np.abs(zv3[npk - 2])
    for k in range(npk - 2, -1, -1):
        sig = oracle_lib.sigmai(a["ztemp"][k], a["zsal"][k], float(a["gdept"][k]), teos10)
        z0 = (sig * a["tmask"][k].astype(np.float64)).astype(f32).astype(np.float64)
        isum = np.maximum(1, a["umask"][k][Jn, Im].astype(int) + a["umask"][k][Jn, I] + a["umask"][k][J, Im] + a["umask"][k][J, I])
        g = (np.abs(z0[Jn, I] - z0[Jn, Im]) * um[k][Jn, Im] / a["e1u"][Jn, Im] + np.abs(z0[Jn, Ip] - z0[Jn, I]) * um[k][Jn, I] / a["e1u"][Jn, I]
             + np.abs(z0[J, I] - z0[J, Im]) * um[k][J, Im] / a["e1u"][J, Im] + np.abs(z0[J, Ip] - z0[J, I]) * um[k][J, I] / a["e1u"][J, I])
        inc = np.abs(a["zcoef"][J, I].astype(np.float64)) * g / isum * ib[J, I, 0] * e3v[k][J, I]
        if k < npk - 2:
            zvg[k] = zvg[k + 1]
        zvg[k][J, I] = dvgeo[J, I] + inc
        dvgeo[J, I] += inc
    raw = (wb * zvg[..., None]).sum(axis=2)
    col = (e3v[:npk - 1].astype(np.float64) * zvg[:npk - 1]).sum(axis=0)
    dvbt2 = np.where(hdep != 0, col / np.where(hdep != 0, hdep, 1), 0.0)
    raw = raw + (wb * dvbt2[None, :, :, None]).sum(axis=2)
    raw[npk - 1] = 0
    if prior_bound is not None:
        raw = raw + prior_bound
    elif dmoc_sh is not None:
        raw = raw + np.abs(np.asarray(dmoc_sh, np.float64))                  # (accumulated onto as it is, before the division by 1e6)
    b["sh"] = cum(raw)
    b["ag"] = b["total"] + b["sh"] + b["bt"]
    return b


def within_bounds(got, ref, bounds, rel=1e-13):
    for k in ("total", "sh", "bt", "ag"):
        err, tol = np.abs(got[k] - ref[k]), rel * bounds[k] + 1e-300
        assert np.all(err <= tol), (k, float((err / tol).max()), np.unravel_index(np.argmax(err / tol), err.shape))


@pytest.mark.parametrize("npi,npj,npk,nb,teos", [(24, 18, 7, 5, False), (31, 11, 5, 1, True), (3, 3, 3, 5, False)])
def test_oracle_decomp_matches_numpy_restatement(npi, npj, npk, nb, teos):
    a = decomp_inputs(npi, npj, npk, nb, seed=npi)
    first = oracle_lib.moc_decomp(**a, teos10=teos)
    close(first, numpy_decomp(a, teos))
    within_bounds(first, numpy_decomp(a, teos), decomp_bounds(a, teos))
    assert np.array_equal(first["total"], oracle_lib.moc(a["e1v"], a["e3v"], a["ibmask"], a["zv"])[0])
    assert np.array_equal(first["ag"], first["total"] - first["sh"] - first["bt"])
    if npi > 3:
        assert np.abs(first["sh"]).max() > 0 and np.abs(first["bt"]).max() > 0
    # the reference accumulates the shear part of a record onto the previous record's result (:471)
    second = oracle_lib.moc_decomp(**a, teos10=teos, dmoc_sh=first["sh"])
    close(second, numpy_decomp(a, teos, dmoc_sh=first["sh"]))
    if npi > 3:
        assert not np.array_equal(second["sh"], first["sh"])


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb,teos", [(24, 18, 7, 5, False), (182, 149, 31, 5, False), (31, 11, 5, 1, True),
                                                 (3, 3, 3, 5, False), (1031, 9, 6, 5, True), (130, 40, 4, 8, False)])
def test_hip_decomp_matches_oracle(gpu, monkeypatch, npi, npj, npk, nb, teos):
    a = decomp_inputs(npi, npj, npk, nb, seed=npi + 7)
    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(a["e1v"], a["e3v"], a["ibmask"])
    plan.decomp_set_mesh(a["e3v"], a["e1u"], a["zcoef"], a["gdept"], a["tmask"], a["umask"])
    eos = gpu.EOS_TEOS10 if teos else gpu.EOS_80
    ref = oracle_lib.moc_decomp(**a, teos10=teos)
    got = plan.decomp(a["zv"], a["ztemp"], a["zsal"], eos=eos)
    within_bounds(got, ref, decomp_bounds(a, teos))
    ref2 = oracle_lib.moc_decomp(**a, teos10=teos, dmoc_sh=ref["sh"])
    got2 = plan.decomp(a["zv"], a["ztemp"], a["zsal"], dmoc_sh=ref["sh"], eos=eos)
    within_bounds(got2, ref2, decomp_bounds(a, teos, dmoc_sh=ref["sh"]))
    again = plan.decomp(a["zv"], a["ztemp"], a["zsal"], eos=eos)
    assert all(np.array_equal(again[k], got[k]) for k in got)            # fixed order: run-to-run identical
    # one sigmai launch over all levels + one kernel that walks the columns (what runs) against a launch of each per level:
    # the same statements per cell and level, the same bits
    monkeypatch.setenv("CDFT_MOC_DECOMP_LEVELS", "1")
    by_level = plan.decomp(a["zv"], a["ztemp"], a["zsal"], eos=eos)
    monkeypatch.delenv("CDFT_MOC_DECOMP_LEVELS")
    assert all(np.array_equal(by_level[k], got[k]) for k in got)
    plan.close()


@pytest.mark.gpu
def test_hip_decomp_needs_its_mesh_and_a_domain_of_3_by_3(gpu):
    a = decomp_inputs(24, 18, 7, 5, seed=1)
    plan = gpu.MocPlan(24, 18, 7, 5)
    plan.set_mesh(a["e1v"], a["e3v"], a["ibmask"])
    with pytest.raises(gpu.CdftError):
        plan.decomp(a["zv"], a["ztemp"], a["zsal"])
    small = gpu.MocPlan(2, 5, 3, 1)
    with pytest.raises(gpu.CdftError):
        small.decomp_set_mesh(np.zeros((3, 5, 2), f32), np.ones((5, 2), f32), np.ones((5, 2), f32), np.ones(3, f32),
                              np.ones((3, 5, 2), np.uint8), np.ones((3, 5, 2), np.uint8))


@pytest.mark.gpu
def test_mask_e3v_on_the_device_is_get_e3v(gpu):
    """cdft_moc_mask_e3v = e3v * INTEGER(2) mask of src/cdfmoc.f90:576-596, bit for bit: a NaN of the metric under a 0 of the
    mask stays a NaN (0 * NaN), a negative value becomes -0."""
    import torch
    rng = np.random.default_rng(5)
    n = 3 * 257 * 131 + 3
    e3 = (1 + 200 * rng.random(n)).astype(np.float32)
    e3[::97] = -e3[::97]
    e3[5::1001] = np.nan
    m = (rng.random(n) > 0.4).astype(np.int8)
    want = e3 * m.astype(np.int16).astype(np.float32)
    got = gpu.moc_mask_e3v(torch.from_numpy(e3).cuda(), torch.from_numpy(m).cuda()).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    with pytest.raises(TypeError):
        gpu.moc_mask_e3v(torch.from_numpy(e3), torch.from_numpy(m))

