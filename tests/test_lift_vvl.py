"""-vvl (time-varying vertical metrics) pinned like everything else in tests/test_lift.py / test_lift_next.py.

Every program that parses -vvl reads its e3 at the record's time index (src/cdfmoc.f90:339-348 and :710-719 for -rapid,
cdfheatc.f90:197-215, cdfmhst.f90:303-330, cdfpsi.f90:257-294, cdfmean.f90:397-412, cdfsum.f90:292-314, cdfmocsig.f90:362-388,
cdfvertmean.f90:225-243, cdfvhst.f90:145-165, cdfvint.f90:269-285, cdfvtrp.f90:181-197; cdfvT's -vvl is in
tests/test_lift.py).  The harnesses of oracle/ref_lift.py take `kvvl` and an e3 array with one field per record, so
`lg_vvl = .TRUE.` runs through the reference's own statements; tests/golden/lift_vvl.npz holds their outputs on cases whose
metric differs from record to record (tests/lift_cases.py, "-vvl").

  CPU : oracle == golden BIT FOR BIT (the oracle is called once per record with that record's metric, which is what the
        programs of this build do under -vvl: set_mesh per record).
  GPU : the HIP path, one set_mesh + compute per record, against the same goldens under each tool's stated tolerance.
"""
import os

import numpy as np
import pytest

import lift_cases as lc
import oracle_lib
import oracle_tools as ot
import ref_lift_lib as rl
import synth
from test_cli_next import vint_schedule
from test_cli_rapid import WIND, numpy_rapid
from test_lift import (decomp_vars, heatc_vars, mhst_masks5, mhst_vars, moc_vars, psi_by, same, ulp4)
from test_lift_next import mean_vars

f32 = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def G():
    return np.load(os.path.join(GOLD, "lift_vvl.npz"))


def test_the_metric_really_changes_the_result():
    """The vvl goldens differ from the static-metric goldens of the same inputs (a harness that ignored kvvl would not)."""
    g = G()
    assert not same(g["moc.vvl5.dmoc"], np.load(os.path.join(GOLD, "lift_cdfmoc.npz"))["small5.dmoc"])
    assert not same(g["heatc.vvl.sums"], np.load(os.path.join(GOLD, "lift_cdfheatc.npz"))["default.sums"])
    assert not same(g["psi.vvl_u.dpsi"], np.load(os.path.join(GOLD, "lift_cdfpsi.npz"))["u.dpsi"])
    assert not same(g["vertmean"], np.load(os.path.join(GOLD, "lift_cdfvint_cdfvertmean.npz"))["vertmean.15_120"])


# ================================================================================================ cdfmoc
def moc_records(name, moc, decomp):
    """moc(e1v, e3v, ib, zv_record) -> (npk,npj,nb); decomp(g, e1v, e3v, ib, rec, zv, t, s, teos, sh) -> dict."""
    g, zv, nb, dec, teos, e3v_t = lc.moc_vvl_case(name)
    e1v, _, ib = synth.moc_inputs(g, nb)
    vm = g["vmask"].astype(f32)
    tot, parts, sh = [], [], None
    for t in range(zv.shape[0]):
        e3v = (e3v_t[t] * vm).astype(f32)               # get_e3v(jk, it): e3v of the record times vmask (src/cdfmoc.f90:589-594)
        if dec is None:
            tot.append(moc(e1v, e3v, ib, zv[t]))
        else:
            o = decomp(g, e1v, e3v, ib, zv[t], dec["t"][t], dec["s"][t], teos, sh)
            sh = o["sh"]
            tot.append(o["total"])
            parts.append(np.stack([o["sh"], o["bt"], o["ag"]]))
    return g, nb, dec, np.stack(tot), (np.stack(parts) if parts else None), e1v, ib, zv, e3v_t


def oracle_decomp(g, e1v, e3v, ib, zv, zt, zs, teos, sh):
    return oracle_lib.moc_decomp(e1v=e1v, e3v=e3v, ibmask=ib, zv=zv, ztemp=zt, zsal=zs, tmask=g["tmask"].astype(np.uint8),
                                 umask=g["umask"].astype(np.uint8), e1u=g["e1u"], zcoef=lc.moc_zcoef(g),
                                 gdept=g["gdept_1d"].astype(f32), teos10=teos, dmoc_sh=sh)


@pytest.mark.parametrize("name", lc.MOC_VVL_CASES)
def test_oracle_moc_vvl(name):
    g, nb, dec, tot, parts, *_ = moc_records(name, lambda e1v, e3v, ib, zv: oracle_lib.moc(e1v, e3v, ib, zv)[0], oracle_decomp)
    gd = G()
    assert same(tot, gd[f"moc.{name}.dmoc"])
    if dec is None:
        assert same(moc_vars(tot, nb), gd[f"moc.{name}.var"])
    else:
        assert same(parts, gd[f"moc.{name}.dec"]) and same(decomp_vars(tot, parts, nb), gd[f"moc.{name}.var"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.MOC_VVL_CASES)
def test_hip_moc_vvl(gpu, name):
    g0, _, nb0, *_ = lc.moc_vvl_case(name)
    npk, npj, npi = g0["tmask"].shape
    plan = gpu.MocPlan(npi, npj, npk, nb0)

    def moc(e1v, e3v, ib, zv):
        plan.set_mesh(e1v, e3v, ib)                      # the program's -vvl path: the mesh again at every record
        return plan.compute(zv[None])[0]

    from test_moc_decomp import decomp_bounds
    bounds = []

    def decomp(g, e1v, e3v, ib, zv, zt, zs, teos, sh):
        plan.set_mesh(e1v, e3v, ib)
        plan.decomp_set_mesh(e3v, g["e1u"], lc.moc_zcoef(g), g["gdept_1d"], g["tmask"], g["umask"])
        a = dict(e1v=e1v, e3v=e3v, ibmask=ib, zv=zv, ztemp=zt, zsal=zs, tmask=g["tmask"].astype(np.uint8), umask=g["umask"].astype(np.uint8),
                 e1u=g["e1u"], zcoef=lc.moc_zcoef(g), gdept=g["gdept_1d"].astype(f32))
        bounds.append(decomp_bounds(a, teos, prior_bound=bounds[-1]["sh"] if bounds else None))
        return plan.decomp(zv, zt, zs, dmoc_sh=sh, eos=gpu.EOS_TEOS10 if teos else gpu.EOS_80)
    g, nb, dec, tot, parts, e1v, ib, zv, e3v_t = moc_records(name, moc, decomp)
    gd = G()
    vm = g["vmask"].astype(f32)
    bound = np.stack([-oracle_lib.moc(e1v, (e3v_t[t] * vm).astype(f32), ib, np.abs(zv[t]))[0] for t in range(zv.shape[0])])
    want = gd[f"moc.{name}.dmoc"]
    if dec is None:
        assert np.all(np.abs(tot - want) <= 1e-13 * bound + 1e-300)
        assert np.all(np.abs(moc_vars(tot, nb).astype(np.float64) - gd[f"moc.{name}.var"]) <= ulp4(gd[f"moc.{name}.var"]))
    else:                                                # the tolerance of tests/test_lift.py's -decomp cases: 1e-13 of the sum of |terms|
        for t in range(zv.shape[0]):
            for got, ref, key in [(tot[t], want[t], "total")] + [(parts[t, c], gd[f"moc.{name}.dec"][t, c], k) for c, k in enumerate(("sh", "bt", "ag"))]:
                assert np.all(np.abs(got - ref) <= 1e-13 * bounds[t][key] + 1e-300), (t, key)


# ================================================================================================ cdfheatc
@pytest.mark.parametrize("name", lc.HEATC_VVL_CASES)
def test_oracle_heatc_vvl(name):
    g, temp, mxlopt, mxl, e3t_t = lc.heatc_vvl_case(name)
    o4 = np.stack([ot.heatc(g["e1t"], g["e2t"], g["tmask"], temp[t], mxl=None if mxl is None else mxl[t], gdepw=g["gdepw_1d"],
                            mxlopt=mxlopt, r4sum=True, e3t=e3t_t[t]) for t in range(temp.shape[0])])
    assert same(o4, G()[f"heatc.{name}.sums"]) and same(heatc_vars(o4), G()[f"heatc.{name}.var"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.HEATC_VVL_CASES)
def test_hip_heatc_vvl(gpu, name):
    g, temp, mxlopt, mxl, e3t_t = lc.heatc_vvl_case(name)
    npk, npj, npi = g["tmask"].shape
    want = G()[f"heatc.{name}.sums"]
    tm = g["tmask"].astype(np.uint8)
    n = npi * npj
    for r4seq, devices in ((False, None), (True, None), (True, [0, 0, 0])):
        plan = gpu.HeatcPlan(npi, npj, npk, devices=devices)
        plan.set_sum_mode(r4seq)
        for t in range(temp.shape[0]):
            plan.set_mesh(g["e1t"], g["e2t"], tm, e3t=e3t_t[t])
            got = plan.compute(temp[t], rmxldep=None if mxl is None else mxl[t], gdepw=g["gdepw_1d"], mxloption=mxlopt)
            if r4seq:                                    # the reference's own sequential r4 sums, bit for bit
                assert same(got, want[t]), (devices, t)
            else:
                absb = ot.heatc(g["e1t"], g["e2t"], g["tmask"], np.abs(temp[t]), mxl=None if mxl is None else mxl[t],
                                gdepw=g["gdepw_1d"], mxlopt=mxlopt, e3t=e3t_t[t])
                assert np.all(np.abs(got - want[t]) <= n * 2.0 ** -24 * absb + 1e-300)


# ================================================================================================ cdfmhst
def mhst_recs(name):
    g, nb, kw = lc.mhst_vvl_case(name)
    a = {k: kw[k] for k in ("vt", "vs", "zv", "zt", "zs") if k in kw}
    return g, nb, kw, a, mhst_masks5(g, nb), next(iter(a.values())).shape[0]


@pytest.mark.parametrize("name", lc.MHST_VVL_CASES)
def test_oracle_mhst_vvl(name):
    g, nb, kw, a, m, npt = mhst_recs(name)
    gd, npk, nz = G(), g["tmask"].shape[0], m.shape[0]
    for t in range(npt):
        h, s = ot.mhst(g["e1v"], m, e3v=kw["e3v_t"][t], **{k: v[t] for k, v in a.items()})
        assert same(h.T, gd[f"mhst.{name}.zon"][t, 0:nz]) and same(s.T, gd[f"mhst.{name}.zon"][t, 4:4 + nz])
        assert same(mhst_vars(h, s, nb, False, npk), gd[f"mhst.{name}.var"][t])


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.MHST_VVL_CASES)
def test_hip_mhst_vvl(gpu, name):
    g, nb, kw, a, m, npt = mhst_recs(name)
    gd, (npk, npj, npi), nz = G(), g["tmask"].shape, m.shape[0]
    plan = gpu.MhstPlan(npi, npj, npk, nz)
    for t in range(npt):
        plan.set_mesh(g["e1v"], m, e3v=kw["e3v_t"][t])
        rec = {k: v[t] for k, v in a.items()}
        h, s = plan.compute(**rec)
        bh, bs = ot.mhst(g["e1v"], m, e3v=kw["e3v_t"][t], **{k: np.abs(v) for k, v in rec.items()})
        assert np.all(np.abs(h.T - gd[f"mhst.{name}.zon"][t, 0:nz]) <= 1e-13 * bh.T + 1e-300)
        assert np.all(np.abs(s.T - gd[f"mhst.{name}.zon"][t, 4:4 + nz]) <= 1e-13 * bs.T + 1e-300)
        want = gd[f"mhst.{name}.var"][t]
        assert np.all(np.abs(mhst_vars(h, s, nb, False, npk).astype(np.float64) - want) <= 2 * ulp4(want))


# ================================================================================================ cdfpsi
@pytest.mark.parametrize("name", lc.PSI_VVL_CASES)
def test_oracle_psi_vvl(name):
    var, dpsi = psi_by(lambda npk, **k: ot.psi(npk, **k), ot.psi_open, name, case=lc.psi_vvl_case)
    assert same(var, G()[f"psi.{name}.var"]) and same(dpsi, G()[f"psi.{name}.dpsi"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.PSI_VVL_CASES)
def test_hip_psi_vvl(gpu, name):
    g, _ = lc.psi_vvl_case(name)
    npk, npj, npi = g["tmask"].shape
    p = gpu.PsiPlan(npi, npj, npk)

    def psi_open(n, zu, e2u, e3u, zv, e1v, e3v, lmean=False, iiref=None, ijref=None, fmask=None):
        p.integrate(zu=zu, e2u=e2u, e3u=e3u, zv=zv, e1v=e1v, e3v=e3v)
        p.scan_open(lmean)
        return p.finish(None, npi if iiref is None else iiref, npj if ijref is None else ijref, fmask)
    var, dpsi = psi_by(lambda n, **k: p.compute(**k), psi_open, name, case=lc.psi_vvl_case)
    assert same(var, G()[f"psi.{name}.var"]) and same(dpsi, G()[f"psi.{name}.dpsi"])          # bit-exact, like the static cases


# ================================================================================================ the 8f tools
def by_8f(vhst, vint, mean_fn, vertmean, vint_cumul_fn):
    """Every 8f tool of the -vvl golden from a set of per-record callables (the oracle's or the HIP path's)."""
    g, t, s, u, v = lc.grid8f()
    e3t_t, e3u_t, e3v_t = lc.e3_8f()
    tm = g["tmask"].astype(f32)
    _, vt, vs, ut, us, _ = lc.vhst_case("default")
    out = {}
    h = np.zeros((2, 4) + g["e1t"].shape, f32)
    tr = np.zeros((2, 2) + g["e1t"].shape, f32)
    for k in range(2):
        a, b = vhst(vt[k], vs[k], g["e1v"], e3=e3v_t[k])
        c, d = vhst(ut[k], us[k], g["e2u"], e3=e3u_t[k])
        h[k] = np.stack([x.astype(f32) for x in (a, b, c, d)])
        tr[k, 0] = vint(u[k], g["e2u"], e3=e3u_t[k]).astype(f32)
        tr[k, 1] = vint(v[k], g["e1v"], e3=e3v_t[k]).astype(f32)
    out["vhst"], out["vtrp"] = h, tr
    _, field, bas, _ = lc.mean_case("basins3_var_sum")
    ib = np.ascontiguousarray(np.moveaxis(bas, 0, -1)).astype(np.int16)
    out["mean.sums"] = np.stack([mean_fn(g["e1t"], g["e2t"], tm, ib, field[k], e3t_t[k], 0) for k in range(2)])
    one = np.ones(g["e1t"].shape + (1,), np.int16)
    out["sum.sums"] = np.stack([mean_fn(g["e1t"], g["e2t"], tm, one, t[k], e3t_t[k], 1)[:, 0, :3] for k in range(2)])
    gdep = g["gdepw_1d"]
    up, down = 15.0, 120.0
    ik1, ik2 = int(np.nonzero(gdep <= up)[0].max()), int(np.nonzero(gdep <= down)[0].max())
    sl = slice(ik1, ik2 + 1)
    out["vertmean"] = np.stack([vertmean(t[k][sl], tm[sl], gdep[ik1], gdep[ik2], up, down, e3=e3t_t[k][sl]).astype(f32) for k in range(2)])
    for name, full, gsop, occi, mean in lc.VINT_CASES:
        if not full:
            gdepo = lc.vint_gdepo(g, gsop, occi)
            out[f"vint.{name}"] = np.stack([vint_cumul_fn(g, t[k], tm, gdepo, gsop or occi, lc.vint_sclf(mean), mean, e3t_t[k])
                                            for k in range(2)])
    return out


def check_8f(out, exact_only=None, bounds=None):
    gd = G()
    for k, v in out.items():
        if k in ("mean.sums", "sum.sums") and bounds is not None:
            assert np.all(np.abs(v - gd[k]) <= 1e-13 * np.abs(bounds[k]) + 1e-300), k
        else:
            assert same(v, gd[k]), k


def test_oracle_8f_tools_vvl():
    out = by_8f(ot.vhst_transport, ot.vint_transport, lambda e1, e2, tm, ib, zv, e3, form: ot.mean(e1, e2, tm, ib, zv, e3=e3, form=form),
                ot.vertmean, lambda g, zt, tm, gdepo, lstd, sclf, mean, e3t: ot.vint_cumul(zt, tm, g["e3t_1d"], gdepo, lstd=lstd, sclf=sclf,
                                                                                          mean=mean, e3t=e3t))
    check_8f(out)
    gd = G()
    v, v0 = mean_vars(out["mean.sums"], True, True)
    assert same(v, gd["mean.var"]) and same(v0, gd["mean.var0"])
    lev = np.where(out["sum.sums"][..., 1] != 0, out["sum.sums"][..., 2].astype(f32), f32(99999.0))
    assert same(lev, gd["sum.var"][:, 0])


@pytest.mark.gpu
def test_hip_8f_tools_vvl(gpu):
    g, *_ = lc.grid8f()
    npk, npj, npi = g["tmask"].shape
    plans = {}

    def hip_mean(e1, e2, tm, ib, zv, e3, form):
        nb = ib.shape[-1]
        if nb not in plans:
            plans[nb] = gpu.MeanPlan(npi, npj, npk, nb)
        p = plans[nb]
        p.set_mesh(e1, e2, tm, ib, e3=e3)                # -vvl: the mesh again at every record
        p.set_form(form)
        return p.compute(zv)

    def hip_vint(g, zt, tm, gdepo, lstd, sclf, mean, e3t):
        kout, hit, clip, rdep = vint_schedule(g["e3t_1d"], gdepo, lstd)
        return gpu.vint_cumul(zt, tm, kout, hit, clip, rdep, len(gdepo), sclf=float(sclf), mean=mean, e3t=e3t)
    out = by_8f(gpu.vhst_transport, gpu.vint_transport, hip_mean, gpu.vertmean, hip_vint)
    bounds = by_8f(ot.vhst_transport, ot.vint_transport,
                   lambda e1, e2, tm, ib, zv, e3, form: ot.mean(e1, e2, tm, ib, np.abs(zv), e3=e3, form=form), ot.vertmean,
                   lambda g, zt, tm, gdepo, lstd, sclf, mean, e3t: np.zeros(1))
    check_8f(out, bounds=bounds)                         # per-column integrals bit-exact, horizontal sums 1e-13 * sum|terms|


# ================================================================================================ cdfmocsig
def mocsig_vvl(fn, eos_kw):
    name, pref, (nbins, sigmin, sigstp), teos, full, eiv, ntr = lc.MOCSIG_VVL_CASE
    g, zv, zt, zs = lc.mocsig_fields()
    e1v, _, ib = synth.moc_inputs(g, 5)
    e3v_t = lc.mocsig_e3v()
    outs = [fn(zv[k], zt[k], zs[k], e1v, ib, nbins, f32(sigmin), f32(sigstp), pref=pref, gdep=-g["gdept_1d"], isodep=True,
               e3v=e3v_t[k], **eos_kw) for k in range(2)]
    return name, np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs])


def test_oracle_mocsig_vvl():
    name, dmoc, depi = mocsig_vvl(oracle_lib.mocsig, dict(teos10=False, ntr=False))
    gd = G()
    assert same(dmoc, gd[f"mocsig.{name}.dmoc"]) and same(depi, gd[f"mocsig.{name}.depi"])
    var = gd[f"mocsig.{name}.var"]
    assert same(np.moveaxis(dmoc, (1, 2, 3), (3, 2, 1)).astype(f32), var[:, :5])


@pytest.mark.gpu
def test_hip_mocsig_vvl(gpu):
    name, dmoc, depi = mocsig_vvl(gpu.mocsig, dict(eos=gpu.EOS_80))
    ref = G()[f"mocsig.{name}.dmoc"]
    assert np.all(np.abs(dmoc - ref) <= 1e-10 * np.abs(ref).max() + 1e-300)
    assert np.allclose(depi, G()[f"mocsig.{name}.depi"], rtol=1e-10, atol=0)


# ================================================================================================ cdfmoc -rapid (host code)
def test_numpy_rapid_vvl_equals_the_reference_statements():
    """The NumPy restatement the cdfmoc -rapid program is held to (tests/test_cli_rapid.py), with the section's e3v of each
    record, against the lifted rapid_amoc under lg_vvl (src/cdfmoc.f90:710-724)."""
    gold = G()["rapid"]
    g, v, t, s, taux, (iiw, iie, iigs, ij), full = lc.rapid_case("partial")
    e3v_t = lc.rapid_e3v()
    for rec in range(2):
        ref = numpy_rapid(dict(g, e3v_0=e3v_t[rec]), v[rec], t[rec], s[rec], taux[rec], iiw, iie, iigs, ij)
        for n, nm in enumerate(lc.RAPID_NAMES):
            want = gold[rec, :, n] if nm == "amoc_rapid" else gold[rec, 0, n]
            if nm in WIND:
                assert np.allclose(ref[nm], want, rtol=2e-6, atol=0), nm
            else:
                assert np.array_equal(np.asarray(ref[nm], f32), np.asarray(want, f32)), nm
    static = np.load(os.path.join(GOLD, "lift_cdfmoc_rapid.npz"))["partial"]
    assert not same(static, gold)


# ================================================================================================ live pin
@pytest.mark.skipif(not rl.available(), reason="oracle/_ref/liblift_ref.so is only built where /root/reference exists")
def test_every_golden_is_what_the_harness_returns_now():
    """All committed lift_*.npz files -- the -vvl one and the fifteen others -- array by array against the harness as built in
    this container from /root/reference (a stale or hand-edited golden fails here)."""
    import make_golden_lift
    assert make_golden_lift.check() == []
