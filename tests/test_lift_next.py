"""The SURVEY.md 8f tools -- cdfvhst, cdfvtrp, cdfmean, cdfsum, cdfvertmean, cdfvint, cdfzonalsum, cdfzonalmean,
cdfmocsig -- pinned like the six of tests/test_lift.py: tests/golden/lift_*.npz hold OUTPUTS of harnesses built from the
reference's own statements (oracle/ref_lift.py), the C oracle must equal them BIT FOR BIT (CPU), and the HIP path is
compared with the same goldens under each tool's stated tolerance (GPU): bit-exact for the per-column integrals (cdfvhst,
cdfvtrp, cdfvint, cdfvertmean: every cell walks its levels in the reference's order); 1e-13 * sum|terms| for the
regrouped horizontal sums (cdfmean, cdfsum, cdfzonal*), 1e-10 of the largest value for cdfmocsig (histogram sums)."""
import os

import numpy as np
import pytest

import lift_cases as lc
import oracle_lib
import oracle_tools as ot
import ref_lift_lib as rl
from test_cli_next import vint_schedule

f32 = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, f"lift_{name}.npz"))


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def e3kw(g, full, pt="t"):
    return dict(e31d=g["e3t_1d"]) if full else dict(e3=g[f"e3{pt}_0"])


# ================================================================================================ cdfvhst / cdfvtrp
def vhst_by(vhst, vint, name):
    g, vt, vs, ut, us, full = lc.vhst_case(name)
    _, t, s, u, v = lc.grid8f()
    h = np.zeros((2, 4) + g["e1t"].shape, f32)
    tr = np.zeros((2, 2) + g["e1t"].shape, f32)
    for k in range(2):
        a, b = vhst(vt[k], vs[k], g["e1v"], **e3kw(g, full, "v"))
        c, d = vhst(ut[k], us[k], g["e2u"], **e3kw(g, full, "u"))
        h[k] = np.stack([x.astype(f32) for x in (a, b, c, d)])            # SNGL(dtrp..), src/cdfvhst.f90:177-180
        tr[k, 0] = vint(u[k], g["e2u"], **e3kw(g, full, "u")).astype(f32)   # REAL(dtrpu), src/cdfvtrp.f90:207-208
        tr[k, 1] = vint(v[k], g["e1v"], **e3kw(g, full, "v")).astype(f32)
    return h, tr


@pytest.mark.parametrize("name", lc.VHST_CASES)
def test_oracle_vhst_vtrp_equal_reference_statements(name):
    h, tr = vhst_by(ot.vhst_transport, ot.vint_transport, name)
    G = gold("cdfvhst_cdfvtrp")
    assert same(h, G[f"vhst.{name}"]) and same(tr, G[f"vtrp.{name}"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.VHST_CASES)
def test_hip_vhst_vtrp_against_reference_statements(gpu, name):
    h, tr = vhst_by(gpu.vhst_transport, gpu.vint_transport, name)
    G = gold("cdfvhst_cdfvtrp")
    assert same(h, G[f"vhst.{name}"]) and same(tr, G[f"vtrp.{name}"])


# ================================================================================================ cdfmean / cdfsum
def mean_vars(sums, var, lsum):
    """The values src/cdfmean.f90:427-480 hands to putvar / putvar0d: per level mean [, variance] [, sum] and their 3-D
    counterparts, each REAL*4 (rdummymean / rdummy).  sums (npt,npk,nb,4) r8."""
    npt, npk, nb, _ = sums.shape
    nvb = 2 + (2 if var else 0) + (2 if lsum else 0)
    v = np.zeros((npt, nvb * nb, npk), f32)
    v0 = np.zeros((npt, nvb * nb), f32)
    for t in range(npt):
        for b in range(nb):
            base = nvb * b
            dvol = dsum = dvar = 0.0
            for k in range(npk):
                surf, vol, sm, vr = sums[t, k, b]
                dvol, dsum, dvar = dvol + vol, dsum + sm, dvar + vr
                mean = sm / vol if vol != 0 else 99999.0
                v[t, base, k] = mean
                i = 2
                if var:
                    v[t, base + i, k] = (vr / vol - mean * mean) if vol != 0 else 99999.0
                    i += 2
                if lsum:
                    v[t, base + i, k] = sm
            v0[t, base + 1] = dsum / dvol
            i = 2
            if var:
                v0[t, base + i + 1] = dvar / dvol - dsum / dvol * dsum / dvol
                i += 2
            if lsum:
                v0[t, base + i + 1] = dsum
    return v, v0


def mean_inputs(name):
    g, field, bas, full = lc.mean_case(name)
    tm = g["tmask"].astype(f32)
    ib = np.ascontiguousarray(np.moveaxis(bas, 0, -1)).astype(np.int16)
    return g, field, tm, ib, full


@pytest.mark.parametrize("name", lc.MEAN_CASES)
def test_oracle_mean_equals_reference_statements(name):
    g, field, tm, ib, full = mean_inputs(name)
    G = gold("cdfmean_cdfsum")
    sums = np.stack([ot.mean(g["e1t"], g["e2t"], tm, ib, field[t], **e3kw(g, full)) for t in range(2)])
    assert same(sums, G[f"mean.{name}.sums"])
    v, v0 = mean_vars(sums, "var" in name, "sum" in name)
    assert same(v, G[f"mean.{name}.var"])
    # putvar0d is handed rdummy, assigned from the r8 totals: identical up to the r4 rounding of the same values
    assert same(v0, G[f"mean.{name}.var0"])


def sum_oracle(name, fn):
    g, t, s, u, v = lc.grid8f()
    tm = g["tmask"].astype(f32)
    one = np.ones(g["e1t"].shape + (1,), np.int16)
    if name == "model":
        return np.stack([fn(g["e1t"], g["e2t"], tm, one, t[k], e3=g["e3t_0"], form=1)[:, 0, :3] for k in range(2)])
    if name == "forcing":
        return np.stack([fn(g["e1t"], g["e2t"], tm[:1], one, t[k, :1], e3=g["e3t_0"][:1], form=2)[:, 0, :3] for k in range(2)])
    ones2, ones3 = np.ones_like(g["e1t"]), np.ones_like(g["e3t_0"])                    # -nowght -nomsk: src/cdfsum.f90:276-277,:303,:312
    return np.stack([fn(ones2, ones2, np.ones_like(tm), one, t[k], e3=ones3, form=1)[:, 0, :3] for k in range(2)])


@pytest.mark.parametrize("name", lc.SUM_CASES)
def test_oracle_sum_equals_reference_statements(name):
    G = gold("cdfmean_cdfsum")
    sums = sum_oracle(name, ot.mean)
    want = G[f"sum.{name}.sums"]
    if name == "forcing":                                   # the forcing form has no volume sum (src/cdfsum.f90:331-332)
        assert same(sums[..., [0, 2]], want[..., [0, 2]])
    else:
        assert same(sums, want)
    lev = np.where((sums[..., 1] != 0) | (name == "forcing"), sums[..., 2].astype(f32), f32(99999.0))     # rdummy(1,1), :325-328,:338
    assert same(lev, G[f"sum.{name}.var"][:, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.MEAN_CASES + ["sum." + n for n in lc.SUM_CASES])
def test_hip_mean_and_sum_against_reference_statements(gpu, name):
    G = gold("cdfmean_cdfsum")
    if name.startswith("sum."):
        def hip_mean(e1, e2, zmask, ib, zv, e3=None, form=0):
            npk, npj, npi = zv.shape
            p = gpu.MeanPlan(npi, npj, npk, 1)
            p.set_mesh(e1, e2, zmask, ib, e3=e3)
            p.set_form(form)
            return p.compute(zv)
        got = sum_oracle(name[4:], hip_mean)
        bound = sum_oracle(name[4:], lambda *a, **k: ot.mean(a[0], a[1], a[2], a[3], np.abs(a[4]), **k))
        want = G[f"{name}.sums"]
        cols = [0, 2] if name == "sum.forcing" else [0, 1, 2]
        assert np.all(np.abs(got - want)[..., cols] <= 1e-13 * np.abs(bound)[..., cols] + 1e-300)
        return
    g, field, tm, ib, full = mean_inputs(name)
    npk, npj, npi = tm.shape
    p = gpu.MeanPlan(npi, npj, npk, ib.shape[-1])
    p.set_mesh(g["e1t"], g["e2t"], tm, ib, **e3kw(g, full))
    for t in range(2):
        got = p.compute(field[t])
        bound = ot.mean(g["e1t"], g["e2t"], tm, ib, np.abs(field[t]), **e3kw(g, full))
        assert np.all(np.abs(got - G[f"mean.{name}.sums"][t]) <= 1e-13 * bound + 1e-300)
        # cdfmean without -var (CDFT_MEAN_FORM_MEAN_NOVAR): the same three sums, bit for bit, and no variance sum
        p.set_form(3)
        nov = p.compute(field[t])
        p.set_form(0)
        assert np.array_equal(nov[..., :3], got[..., :3]) and not nov[..., 3].any()
    with pytest.raises(gpu.CdftError):
        p.set_form(4)


# ================================================================================================ cdfvertmean / cdfvint
def column_by(vertmean, vint_cumul_fn):
    g, t, s, u, v = lc.grid8f()
    tm, gdep = g["tmask"].astype(f32), g["gdepw_1d"]
    out = {}
    for name, up, down, full in lc.VERTMEAN_CASES:
        ik1 = int(np.nonzero(gdep <= up)[0].max()) if (gdep <= up).any() else 0            # src/cdfvertmean.f90:214-218
        ik2 = int(np.nonzero(gdep <= down)[0].max()) if (gdep <= down).any() else len(gdep) - 1
        sl = slice(ik1, ik2 + 1)
        kw = dict(e31d=g["e3t_1d"][sl]) if full else dict(e3=g["e3t_0"][sl])
        out[f"vertmean.{name}"] = np.stack([vertmean(t[k][sl], tm[sl], gdep[ik1], gdep[ik2], up, down, **kw).astype(f32) for k in range(2)])
    for name, full, gsop, occi, mean in lc.VINT_CASES:
        gdepo = lc.vint_gdepo(g, gsop, occi)
        out[f"vint.{name}"] = np.stack([vint_cumul_fn(g, t[k], tm, gdepo, gsop or occi, lc.vint_sclf(mean), mean, full) for k in range(2)])
    return out


def oracle_vint(g, zt, tm, gdepo, lstd, sclf, mean, full):
    return ot.vint_cumul(zt, tm, g["e3t_1d"], gdepo, lstd=lstd, sclf=sclf, mean=mean, e3t=None if full else g["e3t_0"], full=full)


def test_oracle_vertmean_and_vint_equal_reference_statements():
    G = gold("cdfvint_cdfvertmean")
    for k, v in column_by(ot.vertmean, oracle_vint).items():
        assert same(v, G[k]), k


@pytest.mark.gpu
def test_hip_vertmean_and_vint_against_reference_statements(gpu):
    def hip_vint(g, zt, tm, gdepo, lstd, sclf, mean, full):
        kout, hit, clip, rdep = vint_schedule(g["e3t_1d"], gdepo, lstd)
        kw = dict(e31d=g["e3t_1d"]) if full else dict(e3t=g["e3t_0"])
        return gpu.vint_cumul(zt, tm, kout, hit, clip, rdep, len(gdepo), sclf=float(sclf), mean=mean, **kw)
    G = gold("cdfvint_cdfvertmean")
    for k, v in column_by(gpu.vertmean, hip_vint).items():
        assert same(v, G[k]), k


# ================================================================================================ cdfzonalsum / cdfzonalmean
def zonal_expected(o, nb, opt, g, mean_form):
    """dict of what the harness captured from one oracle-shaped result {sum [, area, max, min]} (nb,nlev,npj)."""
    if not mean_form:
        s = o["sum"]
        if opt:                                               # -pdeg: dzosum / alpha, src/cdfzonalsum.f90:262-264,:320
            z2pi = f32(2.0) * f32(np.arccos(f32(-1.)))
            alpha = (g["e2t"][:, 0] * f32(360.) / z2pi / f32(6371229.)).astype(f32)
            s = s / alpha[None, None, :].astype(np.float64)
        return {"r8": s, "var": s.astype(f32)}
    mean = np.where(o["area"] != 0, o["sum"] / np.where(o["area"] != 0, o["area"], 1), 0.0)      # :332-336, zspval = 0
    d = {"r8": mean, "area": o["area"], "var": mean.astype(f32)}
    if opt:                                                   # -max: :337-341
        d["max"] = np.where(o["area"] == 0, f32(0), o["max"])
        d["min"] = np.where(o["area"] == 0, f32(0), o["min"])
    return d


@pytest.mark.parametrize("name,nb,opt", lc.ZONAL_CASES)
def test_oracle_zonal_equals_reference_statements(name, nb, opt):
    g, t, s, u, v = lc.grid8f()
    tm, zm = g["tmask"].astype(f32), lc.zonal_masks(g, nb)
    G = gold("cdfzonal")
    for k in range(2):
        e = zonal_expected(ot.zonal(g["e1t"], g["e2t"], tm, zm, t[k]), nb, opt, g, False)
        assert same(e["r8"], G[f"sum.{name}.r8"][k]) and same(e["var"], G[f"sum.{name}.var"][k])
        e = zonal_expected(ot.zonal(g["e1t"], g["e2t"], tm, zm, t[k], mean_form=True), nb, opt, g, True)
        assert same(e["r8"], G[f"mean.{name}.r8"][k]) and same(e["area"], G[f"mean.{name}.area"][k])
        assert same(e["var"], G[f"mean.{name}.var"][k, :nb])
        if opt:
            assert same(e["max"], G[f"mean.{name}.var"][k, nb:2 * nb]) and same(e["min"], G[f"mean.{name}.var"][k, 2 * nb:])


@pytest.mark.gpu
@pytest.mark.parametrize("name,nb,opt", lc.ZONAL_CASES)
def test_hip_zonal_against_reference_statements(gpu, name, nb, opt):
    g, t, s, u, v = lc.grid8f()
    tm, zm = g["tmask"].astype(f32), lc.zonal_masks(g, nb)
    npk, npj, npi = tm.shape
    G = gold("cdfzonal")
    p = gpu.ZonalPlan(npi, npj, npk, nb)
    p.set_mesh(g["e1t"], g["e2t"], tm, zm)
    for k in range(2):
        bound = ot.zonal(g["e1t"], g["e2t"], tm, zm, np.abs(t[k]))["sum"]
        got = p.compute(t[k])
        scale = 1.0
        if opt:
            z2pi = f32(2.0) * f32(np.arccos(f32(-1.)))
            scale = 1.0 / (g["e2t"][:, 0] * f32(360.) / z2pi / f32(6371229.)).astype(f32)[None, None, :].astype(np.float64)
        assert np.all(np.abs(got["sum"] * scale - G[f"sum.{name}.r8"][k]) <= 1e-13 * bound * scale + 1e-300)
        gm = p.compute(t[k], mean_form=True, minmax=opt)
        assert np.all(np.abs(gm["area"] - G[f"mean.{name}.area"][k]) <= 1e-13 * np.abs(G[f"mean.{name}.area"][k]) + 1e-300)
        e = zonal_expected(gm, nb, opt, g, True)
        tol = 1e-13 * np.where(G[f"mean.{name}.area"][k] != 0, ot.zonal(g["e1t"], g["e2t"], tm, zm, np.abs(t[k]), mean_form=True)["sum"]
                               / np.where(G[f"mean.{name}.area"][k] != 0, G[f"mean.{name}.area"][k], 1), 0) * 4 + 1e-300
        assert np.all(np.abs(e["r8"] - G[f"mean.{name}.r8"][k]) <= tol)
        if opt:                                               # extrema are exact
            assert same(e["max"], G[f"mean.{name}.var"][k, nb:2 * nb]) and same(e["min"], G[f"mean.{name}.var"][k, 2 * nb:])


# ================================================================================================ cdfmocsig
def mocsig_by(fn, case, eos_arg):
    name, pref, (nbins, sigmin, sigstp), teos, full, eiv, ntr = case
    import synth
    g, zv, zt, zs = lc.mocsig_fields()
    e1v, e3v, ib = synth.moc_inputs(g, 5)
    gdep = -g["gdept_1d"]
    kw = dict(e31d=g["e3t_1d"]) if full else dict(e3v=g["e3v_0"])
    outs = [fn(zv[k], zt[k], zs[k], e1v, ib, nbins, f32(sigmin), f32(sigstp), pref=pref, gdep=gdep, isodep=True,
               zveiv=(zv[k] * f32(0.1)).astype(f32) if eiv else None, **eos_arg(teos, ntr), **kw) for k in range(2)]
    return np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs])


@pytest.mark.parametrize("case", lc.MOCSIG_CASES, ids=[c[0] for c in lc.MOCSIG_CASES])
def test_oracle_mocsig_equals_reference_statements(case):
    G = gold("cdfmocsig")
    dmoc, depi = mocsig_by(oracle_lib.mocsig, case, lambda teos, ntr: dict(teos10=teos, ntr=ntr))
    assert same(dmoc, G[f"{case[0]}.dmoc"]) and same(depi, G[f"{case[0]}.depi"])
    nb = 5
    var = G[f"{case[0]}.var"]                                # (npt, 2*nb, nbins, npj): REAL(dmoc(b,bin,:)), REAL(depi(b,bin,:))
    assert same(np.moveaxis(dmoc, (1, 2, 3), (3, 2, 1)).astype(f32), var[:, :nb])
    assert same(np.moveaxis(depi, (1, 2, 3), (3, 2, 1)).astype(f32), var[:, nb:])


@pytest.mark.gpu
@pytest.mark.parametrize("case", lc.MOCSIG_CASES, ids=[c[0] for c in lc.MOCSIG_CASES])
def test_hip_mocsig_against_reference_statements(gpu, case):
    G = gold("cdfmocsig")
    dmoc, depi = mocsig_by(gpu.mocsig, case, lambda teos, ntr: dict(eos=gpu.EOS_NTR if ntr else gpu.EOS_TEOS10 if teos else gpu.EOS_80))
    ref = G[f"{case[0]}.dmoc"]
    assert np.all(np.abs(dmoc - ref) <= 1e-10 * np.abs(ref).max() + 1e-300)
    assert np.allclose(depi, G[f"{case[0]}.depi"], rtol=1e-10, atol=0)


# ================================================================================================ cdfsig0, cdfsigi
@pytest.mark.parametrize("case", lc.SIG_CASES, ids=[c[0] for c in lc.SIG_CASES])
def test_oracle_sig_tool_equals_reference_statements(case):
    """The level loops of cdfsig0 (src/cdfsig0.f90:139-155) and cdfsigi (src/cdfsigi.f90:130-145) -- mask on S == spval,
    sigma0 / sigmai of the reference's compiled eos module, times the mask, stored in r4 -- against oracle_sig_tool."""
    name, pref, teos, spval = case
    zt, zs = lc.sig_fields(spval)
    got = oracle_lib.sig_tool(zt, zs, pref=0.0 if pref is None else pref, spval=spval, teos10=teos)
    assert got.dtype == f32 and same(got, gold("cdfsig")[name])


@pytest.mark.gpu
@pytest.mark.parametrize("case", lc.SIG_CASES, ids=[c[0] for c in lc.SIG_CASES])
def test_hip_sig_tool_against_reference_statements(gpu, case):
    name, pref, teos, spval = case
    zt, zs = lc.sig_fields(spval)
    got = gpu.sig_tool(zt, zs, pref=0.0 if pref is None else pref, spval=spval, eos=gpu.EOS_TEOS10 if teos else gpu.EOS_80)
    assert got.dtype == f32 and same(got, gold("cdfsig")[name])


# ================================================================================================ live pin
@pytest.mark.skipif(not rl.available(), reason="oracle/_ref/liblift_ref.so is only built where /root/reference exists")
def test_goldens_of_the_8f_tools_are_what_the_harness_returns_now():
    g, t, s, u, v = lc.grid8f()
    tm = g["tmask"].astype(f32)
    assert same(rl.cdfvtrp(g, u, v)[:, :2], gold("cdfvhst_cdfvtrp")["vtrp.default"])
    assert same(rl.cdfzonalsum(g, g["e1t"], g["e2t"], g["gphiv"], g["gdept_1d"], tm, t, nb=5)["r8"], gold("cdfzonal")["sum.b5.r8"])
    assert same(rl.cdfvertmean(g["gdepw_1d"], g["e3t_1d"], g["e3t_0"], tm, t, 15.0, 120.0), gold("cdfvint_cdfvertmean")["vertmean.15_120"])


# ================================================================================================ cdfmaxmoc (host program)
def test_cdfmaxmoc_program_equals_reference_statements(tmp_path):
    """cdfmaxmoc is host code (it reads a moc.nc and searches a window): the program itself runs here, on a moc.nc written by
    scipy, and its six numbers equal those of the harness around the reference's own statements (src/cdfmaxmoc.f90:122-178)."""
    import os
    import subprocess
    from scipy.io import netcdf_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cdftools_amd", "bin", "cdfmaxmoc")
    if not os.path.exists(exe):
        pytest.skip("cdftools_amd/bin/cdfmaxmoc is not built")
    depw, lat, moc = lc.maxmoc_fields()
    npk, npj = moc.shape
    d = str(tmp_path)
    f = netcdf_file(os.path.join(d, "moc.nc"), "w", version=2)
    f.createDimension("time_counter", None)
    f.createDimension("depthw", npk)
    f.createDimension("y", npj)
    f.createDimension("x", 1)
    for nm, vals in (("nav_lon", np.zeros((npj, 1), f32)), ("nav_lat", lat[:, None])):
        v = f.createVariable(nm, "f4", ("y", "x")); v[:] = vals
    v = f.createVariable("depthw", "f4", ("depthw",)); v[:] = depw.astype(f32)
    v = f.createVariable("time_counter", "f8", ("time_counter",)); v[:] = np.array([86400.0])
    v = f.createVariable("zomsfatl", "f4", ("time_counter", "depthw", "y", "x")); v[:] = moc[None, :, :, None]
    f.close()
    want = gold("cdfmaxmoc")["six"]
    for n, w in enumerate(lc.MAXMOC_WINDOWS):
        r = subprocess.run([exe, "-f", "moc.nc", "-b", "atl", "-w"] + [repr(x) for x in w] + ["-o", f"max{n}.nc"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        o = netcdf_file(os.path.join(d, f"max{n}.nc"), "r", mmap=False)
        got = np.array([o.variables[k][:].ravel()[0] for k in ("maxmoc", "minmoc", "latmaxmoc", "latminmoc", "depthmaxmoc", "depthminmoc")], f32)
        o.close()
        assert same(got, want[n]), (w, got, want[n])
