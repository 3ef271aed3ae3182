"""cdfmoc -rapid (src/cdfmoc.f90:598-1004, cdf_findij src/cdftools.f90:41-264): host-only option, so the whole
test runs without a GPU.  The checker is a NumPy restatement of the 33 diagnostics written from the
reference's formulas in its loop order and r4 / r8 mixing (scalar loops: the section is a few hundred cells).
"""
import os
import re

import numpy as np
import pytest

import nemo_files
import synth
from test_cli import read_nc, run, tools  # noqa: F401

f32, f64 = np.float32, np.float64
LAT, LONW, LONE, LONGS = 26.5, -80.1, 12.7, -77.4
ZLIM = [f32(x) for x in (0.0, 800.0, 1100.0, 3000.0, 5000.0, 10000.0)]


def gcdist(lona, lonb, lata, latb):
    c = np.pi / 180.0
    pds = (np.cos(lona * c) * np.cos(lata * c) * np.cos(lonb * c) * np.cos(latb * c)
           + np.sin(lona * c) * np.cos(lata * c) * np.sin(lonb * c) * np.cos(latb * c) + np.sin(lata * c) * np.sin(latb * c))
    return np.where(pds >= 1.0, 0.0, (6378.137 + 6356.7523) / 2.0 * np.arccos(np.minimum(pds, 1.0)))


def nearest(g, lon, lat):
    """1-based (i,j) of the F-point nearest to (lon,lat) among the inner points: what the block walk ends on when
    the distance has a single minimum."""
    d = gcdist(f64(lon), g["glamf"].astype(f64), f64(lat), g["gphif"].astype(f64))
    d[0, :] = d[-1, :] = d[:, 0] = d[:, -1] = np.inf
    j, i = np.unravel_index(np.argmin(d), d.shape)
    return i + 1, j + 1


def numpy_rapid(g, v3, t3, s3, taux, iiw, iie, iigs, ij):
    """v3, t3, s3 (npk,npj,npi), taux (npj,npi) of one record -> dict of the output variables."""
    npk = v3.shape[0]
    sl = slice(iiw - 1, iie)
    j = ij - 1
    itm = g["tmask"][:, j, sl].astype(np.int32)                    # (npk, ni)
    v = (v3[:, j, sl] * itm.astype(f32)).astype(f32)
    t = (t3[:, j, sl] * itm.astype(f32)).astype(f32)
    s = (s3[:, j, sl] * itm.astype(f32)).astype(f32)
    tx = taux[j, sl].astype(f64) * itm[0]
    e3 = g["e3v_0"][:, j, sl].astype(f32).copy()
    e3[e3 == 0] = f32(0.1)
    de1 = g["e1v"][j, sl].astype(f64)
    gdepw, gdept = g["gdepw_1d"].astype(f32), g["gdept_1d"].astype(f32)
    ni, ngs = iie - iiw + 1, iigs - iiw + 1
    out = {}

    def face(i, k):                                               # v*e1*e3*1.d0
        return f64(v[k, i]) * de1[i] * f64(e3[k, i])

    amoc = np.zeros(npk)
    for k in range(1, npk):
        a = amoc[k - 1]
        for i in range(ni):
            a = a + face(i, k - 1)
        amoc[k] = a
    out["amoc_rapid"] = (amoc / 1e6).astype(f32)
    out["Total_max_amoc_rapid"] = f32(amoc.max() / 1e6)

    def tsum(i1, i2, fld=None, k2=npk):
        a = 0.0
        for i in range(i1, i2):
            for k in range(k2):
                a = a + (face(i, k) if fld is None else f64(f32(v[k, i] * fld[k, i])) * de1[i] * f64(e3[k, i]))
        return a
    out["tr_GS"] = f32(tsum(0, ngs) / 1e6)
    zalpha = f32(0)
    for c, nm in enumerate(("tr_THERM", "tr_AIW", "tr_UNADW", "tr_LNADW", "tr_BW")):
        zmin, zmax = ZLIM[c], ZLIM[c + 1]
        a = 0.0
        for i in range(ngs, ni):
            for k in range(npk):
                zbot = f32(gdepw[k] + e3[k, i])
                if gdepw[k] >= zmin and zbot <= zmax:
                    zalpha = f32(1)
                if gdepw[k] >= zmax or zbot <= zmin:
                    zalpha = f32(0)
                if gdepw[k] <= zmin and zbot >= zmin:
                    zalpha = f32(f32(zbot - zmin) / e3[k, i])
                if gdepw[k] <= zmax and zbot >= zmax:
                    zalpha = f32(f32(zmax - gdepw[k]) / e3[k, i])
                a = a + face(i, k) * f64(zalpha)
        out[nm] = f32(a / 1e6)
    zpi = f32(4) * np.arctan(f32(1))
    zf = f32(f32(f32(f32(f32(2) * f32(2)) * zpi) / f32(86400)) * np.sin(f32(f32(f32(LAT) * zpi) / f32(180))))
    a = 0.0
    for i in range(ni):
        a = a - tx[i] * de1[i] / f64(f32(f32(1020) * zf))
    trek = a / 1e6
    out["tr_EKMAN"] = f32(a / 1e6)
    out["tr_TOTAL"] = f32(tsum(0, ni) / 1e6)

    def wmean(fld, i1, i2, k2=npk):
        a, ds = 0.0, 0.0
        for i in range(i1, i2):
            for k in range(k2):
                w = de1[i] * f64(e3[k, i]) * f64(itm[k, i])
                ds = ds + w
                a = a + f64(fld[k, i]) * w
        return a, ds
    for nm, fld in (("v", v), ("T", t), ("S", s)):
        for reg, (i1, i2) in (("total", (0, ni)), ("Florida", (0, ngs)), ("MidOcean", (ngs, ni))):
            a, ds = wmean(fld, i1, i2)
            out["mean_%s_%s_section" % (nm, reg)] = f32(a / ds)
    for nm, fld in (("T", t), ("S", s)):
        dmv, dmt = np.zeros(npk), np.zeros(npk)
        for k in range(npk):
            ds = 0.0
            for i in range(ngs, ni):
                dmv[k] = dmv[k] + f64(v[k, i]) * de1[i]
                dmt[k] = dmt[k] + f64(fld[k, i]) * de1[i]
                ds = ds + de1[i] * f64(itm[k, i])
            if ds != 0:
                dmv[k] /= ds
                dmt[k] /= ds
        a = 0.0
        for k in range(npk):
            for i in range(ngs, ni):
                a = a + dmv[k] * dmt[k] * de1[i] * f64(e3[k, i])
        out["MO_meanVtimesmean" + nm] = f32(a / 1e6)
    k100 = int(np.nonzero(gdept >= 100.0)[0].min()) + 1 if (gdept >= 100.0).any() else 0
    for nm, fld, sfx in (("temp", t, "T"), ("salt", s, "S")):
        out["Total_%s_transport" % nm] = f32(tsum(0, ni, fld) / 1e6)
        out["Florida_%s_transport" % nm] = f32(tsum(0, ngs, fld) / 1e6)
        out["MidOcean_%s_transport" % nm] = f32(tsum(ngs, ni, fld) / 1e6)
        for key, k2 in (("Ekman_%s_transport_SS%s" % (nm, sfx), 1), ("Ekman_%s_transport_%s100" % (nm, sfx), k100)):
            a, ds = 0.0, 0.0
            for i in range(ni):
                for k in range(k2):
                    a = a + f64(fld[k, i]) * de1[i] * f64(e3[k, i])
                    ds = ds + de1[i] * f64(e3[k, i]) * f64(itm[k, i])
            out[key] = f32(trek * a / ds)
    ds = mv = mt = ms = 0.0
    for i in range(ni):
        for k in range(k100):
            w = de1[i] * f64(e3[k, i]) * f64(itm[k, i])
            ds, mv, mt, ms = ds + w, mv + f64(v[k, i]) * w, mt + f64(t[k, i]) * w, ms + f64(s[k, i]) * w
    out["Total_meanVtimesmeanT"] = f32(mv * mt / ds / 1e6)
    out["Total_meanVtimesmeanS"] = f32(mv * ms / ds / 1e6)
    return out


def make_rapid_case(d, npi=124, npj=62, npk=9, npt=2, seed=11):
    """A lon-lat grid fine enough for cdf_findij (cell size consistent with the coordinates), every file -rapid reads."""
    g = synth.make_grid(npi, npj, npk, seed=seed)
    rng = np.random.default_rng(seed)
    dlat = (89.9 + 78.0) / (npj - 1) * 111.2e3
    for nm in ("e1f", "e2f", "e1v", "e2v"):
        g[nm] = (dlat * (0.9 + 0.2 * rng.random((npj, npi), dtype=f32))).astype(f32)
    nemo_files.write_mesh(d, g)
    v = synth.velocity(g, npt, seed + 1)
    u = synth.velocity(g, npt, seed + 2, which="u")
    t = ((-2 + 32 * rng.random((npt, npk, npj, npi), dtype=f32)) * g["tmask"][None]).astype(f32)
    s = ((30 + 8 * rng.random((npt, npk, npj, npi), dtype=f32)) * g["tmask"][None]).astype(f32)
    taux = (0.2 * (rng.random((npt, npj, npi), dtype=f32) - f32(0.5))).astype(f32)
    nemo_files.write_data(os.path.join(d, "V.nc"), g, "depthv", {"vomecrty": v})
    nemo_files.write_data(os.path.join(d, "U.nc"), g, "depthu", {"vozocrtx": u, "sozotaux": taux})
    nemo_files.write_data(os.path.join(d, "T.nc"), g, "deptht", {"votemper": t, "vosaline": s})
    return g, v, t, s, taux


def test_cdfmoc_rapid_end_to_end(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    g, v, t, s, taux = make_rapid_case(d)
    iiw, ij = nearest(g, LONW, LAT)
    iie, ij2 = nearest(g, LONE, LAT)
    iigs, _ = nearest(g, LONGS, LAT)
    assert ij == ij2 and iiw < iigs < iie
    rc, out, err = run("cdfmoc", ["-v", "V.nc", "-t", "T.nc", "-u", "U.nc", "-rapid"], d)
    assert rc == 99, out + err                                   # the reference ends -rapid with STOP 99 (:255)
    assert "Total maximum AMOC" in out and re.search(r"idvar =\s+33\b", out)
    var, att, dims = read_nc(os.path.join(d, "rapid_moc.nc"))
    assert dims["x"] == 1 and dims["y"] == 1 and dims["depthw"] == g["tmask"].shape[0] and len(var) >= 33 + 4
    assert np.allclose(var["nav_lat"].ravel(), g["gphiv"][ij - 1, iiw - 1])
    assert att["tr_GS"]["units"] == b"Sverdrup" and att["mean_T_total_section"]["units"] == b"deg. Celsius"
    assert att["Ekman_temp_transport_SST"]["short_name"] == b"Total_temp_transport_SST"
    wind = ("tr_EKMAN", "Ekman_temp_transport_SST", "Ekman_temp_transport_T100", "Ekman_salt_transport_SSS", "Ekman_salt_transport_S100")
    for rec in range(2):
        ref = numpy_rapid(g, v[rec], t[rec], s[rec], taux[rec], iiw, iie, iigs, ij)
        assert len(ref) == 33
        for nm, want in ref.items():
            got = var[nm][rec].ravel()
            if nm in wind:                                       # through an r4 SIN of the compiler's library
                assert np.allclose(got, want, rtol=2e-6, atol=0), nm
            else:
                assert np.array_equal(got, np.atleast_1d(want)), (nm, got, want)
    # the depth classes are not all empty, nor all the same
    assert len({float(var[n][0].ravel()[0]) for n in ("tr_THERM", "tr_AIW", "tr_UNADW", "tr_LNADW", "tr_BW")}) > 1


def test_cdfmoc_rapid_full_step_and_missing_files(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    g, v, t, s, taux = make_rapid_case(d, npi=100, npj=50, npk=7, npt=1, seed=5)
    rc, out, _ = run("cdfmoc", ["-v", "V.nc", "-t", "T.nc", "-u", "nothere.nc", "-rapid"], d)
    assert rc == 99 and "missing" in out and not os.path.exists(os.path.join(d, "rapid_moc.nc"))
    rc, out, err = run("cdfmoc", ["-v", "V.nc", "-t", "T.nc", "-s", "T.nc", "-u", "U.nc", "-rapid", "-full", "-o", "m.nc"], d)
    assert rc == 99 and "Total maximum AMOC" in out, out + err
    var, _, _ = read_nc(os.path.join(d, "rapid_m.nc"))
    iiw, ij = nearest(g, LONW, LAT)
    iie, _ = nearest(g, LONE, LAT)
    iigs, _ = nearest(g, LONGS, LAT)
    g2 = dict(g, e3v_0=np.broadcast_to(g["e3t_1d"][:, None, None], g["e3v_0"].shape).astype(f32))
    ref = numpy_rapid(g2, v[0], t[0], s[0], taux[0], iiw, iie, iigs, ij)
    for nm in ("amoc_rapid", "tr_GS", "tr_TOTAL", "mean_S_MidOcean_section", "Total_meanVtimesmeanT"):
        assert np.array_equal(var[nm][0].ravel(), np.atleast_1d(ref[nm])), nm


# ---------------------------------------------------------------------------------------------- the pin of numpy_rapid
import lift_cases as lc  # noqa: E402

WIND = ("tr_EKMAN", "Ekman_temp_transport_SST", "Ekman_temp_transport_T100", "Ekman_salt_transport_SSS", "Ekman_salt_transport_S100")


@pytest.mark.parametrize("name", [c[0] for c in lc.RAPID_CASES])
def test_numpy_rapid_equals_the_reference_statements(name):
    """numpy_rapid above is what the cdfmoc -rapid program is held to; here it is held itself to the reference's own
    statements: SUBROUTINE rapid_amoc of src/cdfmoc.f90 (:598-1004), lifted by oracle/ref_lift.py into a harness whose outputs
    on a seeded section are tests/golden/lift_cdfmoc_rapid.npz.  Bit for bit, except the five wind-driven numbers, which go
    through an r4 SIN / ATAN of the Fortran runtime (2e-6)."""
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lift_cdfmoc_rapid.npz"))[name]   # (npt,npk,33)
    g, v, t, s, taux, (iiw, iie, iigs, ij), full = lc.rapid_case(name)
    if full:                                                        # -full: e3v(:,jk) = e31d(jk) on the section
        g = dict(g)
        g["e3v_0"] = np.broadcast_to(g["e3t_1d"][:, None, None], g["e3v_0"].shape).astype(f32).copy()
    met = set()
    for rec in range(2):
        ref = numpy_rapid(g, v[rec], t[rec], s[rec], taux[rec], iiw, iie, iigs, ij)
        assert list(ref) and set(ref) == set(lc.RAPID_NAMES)
        for n, nm in enumerate(lc.RAPID_NAMES):
            want = gold[rec, :, n] if nm == "amoc_rapid" else gold[rec, 0, n]
            got = ref[nm]
            if nm in WIND:
                assert np.allclose(got, want, rtol=2e-6, atol=0), (nm, got, want)
            else:
                assert np.array_equal(np.asarray(got, f32), np.asarray(want, f32)), (nm, got, want)
            if nm.startswith("tr_") and float(np.ravel(want)[0]) != 0.0:
                met.add(nm)
    assert {"tr_THERM", "tr_AIW", "tr_UNADW", "tr_LNADW"} <= met      # the depth classes of the section are not empty


@pytest.mark.parametrize("case", __import__("prog_cases").RAPID_CASES, ids=lambda c: c[0])
def test_cdfmoc_rapid_against_the_reference_program(tools, tmp_path, case):  # noqa: F811
    """The program differential of tests/test_prog_diff.py for -rapid: what the reference's own cdfmoc (unmodified, built in the build
    container over this repository's NetCDF layer) left behind on these files -- exit status 99, every line of stdout, rapid_moc.nc with its
    33 variables, their attributes in file order and their values -- against this build's.  Host code on both sides: no GPU needed."""
    import prog_cases as pc
    from test_prog_diff import BIN, GOLD, diff_case, explained
    d = str(tmp_path)
    make_rapid_case(d)
    ref, ref_arr = pc.load_golden(os.path.join(GOLD, f"prog_{case[0]}.npz"))
    got, got_arr = pc.run_case(BIN, case, d)
    got.pop("stderr", None)
    assert ref["rc"] == 99
    bad = [x for x in diff_case(case, ref, ref_arr, got, got_arr) if not explained(case[0], x[0], x[1])]
    assert not bad, bad[:20]

