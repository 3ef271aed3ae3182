"""-vvl through the PROGRAMS: every tool that parses the option runs on files whose vertical metric differs from record to
record (`e3t` / `e3u` / `e3v` in the data files, the `cn_ve3?vvl` names of src/modcdfnames.F90:49-50), and its output is
compared with the oracle called with the metric of each record -- the arithmetic tests/test_lift_vvl.py pins to the
reference's own statements under lg_vvl.  Where -vvl takes the metric from is the reference's choice per tool
(src/cdfmoc.f90:247-250 V file; cdfheatc.f90:150-153 T file; cdfmhst.f90:219-222 and cdfvhst.f90:107-110 VT file;
cdfpsi.f90:178-183 and cdfvtrp.f90:131-136 U and V files; cdfmean.f90:394, cdfsum.f90:270, cdfvertmean.f90:208,
cdfvint.f90:157-160 the input file; cdfmocsig.f90:231-234 V file).  Each test also checks that the run without -vvl gives
another answer, i.e. that the option is not a no-op."""
import os

import numpy as np
import pytest

import lift_cases as lc
import nemo_files
import oracle_lib
import oracle_tools
import synth
from test_cli import read_nc, run, tools  # noqa: F401  (tools is a fixture)

f32 = np.float32


@pytest.fixture(scope="module")
def case(tools, tmp_path_factory):  # noqa: F811
    d = str(tmp_path_factory.mktemp("vvl"))
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    e3 = {p: lc.e3_records(g[f"e3{p}_0"], 2, 8000 + i) for i, p in enumerate("tuv")}
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridV.nc"), g, "depthv", {"vomecrty": rec["v"], "e3v": e3["v"]})
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridU.nc"), g, "depthu", {"vozocrtx": rec["u"], "e3u": e3["u"]})
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridT.nc"), g, "deptht", {"votemper": rec["t"], "vosaline": rec["s"], "e3t": e3["t"]},
                          fill=0.0)
    vt = {"vomevt": (rec["t"] * rec["v"]).astype(f32), "vomevs": (rec["s"] * rec["v"]).astype(f32),
          "vozout": (rec["t"] * rec["u"]).astype(f32), "vozous": (rec["s"] * rec["u"]).astype(f32)}
    nemo_files.write_data(os.path.join(d, "VVL_vt.nc"), g, "deptht", dict(vt, e3u=e3["u"], e3v=e3["v"]))
    return d, g, rec, e3, vt


def differs(d, a, b, names):
    va, _, _ = read_nc(os.path.join(d, a))
    vb, _, _ = read_nc(os.path.join(d, b))
    return any(not np.array_equal(va[n], vb[n]) for n in names)


@pytest.mark.gpu
@pytest.mark.parametrize("gmap", [None, "0,0"])
def test_cdfmoc_vvl(case, gmap):
    d, g, rec, e3, _ = case
    env = {} if gmap is None else {"CDFT_GPU_MAP": gmap}
    rc, out, err = run("cdfmoc", ["-v", "VVL_y0001_gridV.nc", "-vvl", "-o", "moc_vvl.nc"], d, env=env)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "moc_vvl.nc"))
    e1v, _, ib = synth.moc_inputs(g, nbasins=5)
    vm = g["vmask"].astype(f32)
    for t in range(2):
        ref = oracle_lib.moc(e1v, (e3["v"][t] * vm).astype(f32), ib, rec["v"][t])[0]
        for b, nm in enumerate(["zomsfglo", "zomsfatl", "zomsfinp", "zomsfind", "zomsfpac"]):
            want = ref[..., b].astype(f32)
            assert np.all(np.abs(var[nm][t, :, :, 0] - want) <= np.spacing(np.abs(want))), (nm, t)
    rc, _, _ = run("cdfmoc", ["-v", "VVL_y0001_gridV.nc", "-o", "moc_static.nc"], d)
    assert rc == 0 and differs(d, "moc_vvl.nc", "moc_static.nc", ["zomsfglo"])


@pytest.mark.gpu
def test_cdfmoc_decomp_vvl(case):
    from test_moc_decomp import decomp_inputs
    d, g, rec, e3, _ = case
    npk, npj, npi = g["tmask"].shape
    rc, out, err = run("cdfmoc", ["-v", "VVL_y0001_gridV.nc", "-t", "VVL_y0001_gridT.nc", "-decomp", "-vvl", "-o", "dec_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "dec_vvl.nc"))
    a = decomp_inputs(npi, npj, npk, 5, seed=1)
    e1v, _, ib = synth.moc_inputs(g, nbasins=5)
    vm = g["vmask"].astype(f32)
    sh = None
    for t in range(2):
        ref = oracle_lib.moc_decomp(e1v, (e3["v"][t] * vm).astype(f32), ib, rec["v"][t], rec["t"][t], rec["s"][t], g["tmask"].astype(np.uint8),
                                    g["umask"].astype(np.uint8), g["e1u"], a["zcoef"], g["gdept_1d"], dmoc_sh=sh)
        sh = ref["sh"]
        for b, nm in enumerate(["zomsfglo", "zomsfatl"]):
            for suffix, key in (("", "total"), ("_sh", "sh"), ("_bt", "bt"), ("_ag", "ag")):
                tol = 2e-6 * np.abs(ref[key]).max() + 1e-30
                assert np.all(np.abs(var[nm + suffix][t, :, :, 0] - ref[key][..., b]) <= tol), (nm + suffix, t)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"CDFT_HEATC_SUM": "r8"}, {"CDFT_GPU_MAP": "0,0,0"}])
def test_cdfheatc_vvl(case, env):
    d, g, rec, e3, _ = case
    rc, out, err = run("cdfheatc", ["-f", "VVL_y0001_gridT.nc", "-vvl", "-o", "heatc_vvl.nc"], d, env=env)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "heatc_vvl.nc"))
    scale = np.float64(f32(1020.0) * f32(4000.0))
    r8 = "CDFT_HEATC_SUM" in env
    for t in range(2):
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], rec["t"][t], e3t=e3["t"][t], r4sum=not r8)
        if r8:
            assert np.allclose(var["heatc2d"][t, :, 0, 0], scale * ref[:, 2], rtol=1e-12, atol=0)
        else:                                               # the reference's sequential r4 sums, bit for bit
            assert np.array_equal(var["heatc2d"][t, :, 0, 0], scale * ref[:, 2])
            assert var["heatc3d"][t, 0, 0] == scale * np.add.accumulate(ref[:, 2])[-1]
    rc, _, _ = run("cdfheatc", ["-f", "VVL_y0001_gridT.nc", "-o", "heatc_static.nc"], d)
    assert rc == 0 and differs(d, "heatc_vvl.nc", "heatc_static.nc", ["heatc2d"])


def mhst_check(g, mv, e3v_t, rec_kw):
    masks = np.stack([g["vmask"][0], g["tmaskatl"], g["tmaskpac"], g["tmaskind"]]).astype(f32)

    def col(x, b):
        return {"glo": x[:, 0], "atl": x[:, 1], "inp": x[:, 3] + x[:, 2], "ind": x[:, 3], "pac": x[:, 2], "inp0": x[:, 0] - x[:, 1]}[b]
    for t in range(2):
        kw = {k: v[t] for k, v in rec_kw.items()}
        heat, salt = oracle_tools.mhst(g["e1v"], masks, e3v=e3v_t[t], **kw)
        bh, bs = oracle_tools.mhst(g["e1v"], masks, e3v=e3v_t[t], **{k: np.abs(v) for k, v in kw.items()})
        for b in ("glo", "atl", "inp", "ind", "pac", "inp0"):
            for nm, x, bound, div in (("zomht_", heat, bh, 1e15), ("zomst_", salt, bs, 1e6)):
                want = col(x, b) / div
                want = np.where(want == 0, 9999.99, want)
                tol = 1e-13 * bound.sum(axis=1) / div + np.spacing(np.abs(want).astype(f32))
                assert np.all(np.abs(mv[nm + b][t, :, 0] - want) <= tol), (nm + b, t)


@pytest.mark.gpu
def test_cdfmhst_vvl(case):
    d, g, rec, e3, vt = case
    rc, out, err = run("cdfmhst", ["-vt", "VVL_vt.nc", "-MST", "-vvl", "-o", "mhst_vvl.nc"], d)
    assert rc == 0, out + err
    mv, _, _ = read_nc(os.path.join(d, "mhst_vvl.nc"))
    mhst_check(g, mv, e3["v"], dict(vt=vt["vomevt"], vs=vt["vomevs"]))
    rc, _, _ = run("cdfmhst", ["-vt", "VVL_vt.nc", "-MST", "-o", "mhst_static.nc"], d)
    assert rc == 0 and differs(d, "mhst_vvl.nc", "mhst_static.nc", ["zomht_glo"])
    # V, T, S files: the metric comes from the V file (cf_vtfil = cf_vfil, src/cdfmhst.f90:205, :220)
    rc, out, err = run("cdfmhst", ["-v", "VVL_y0001_gridV.nc", "-t", "VVL_y0001_gridT.nc", "-MST", "-vvl", "-o", "mhst_sep_vvl.nc"], d)
    assert rc == 0, out + err
    mv, _, _ = read_nc(os.path.join(d, "mhst_sep_vvl.nc"))
    mhst_check(g, mv, e3["v"], dict(zv=rec["v"], zt=rec["t"], zs=rec["s"]))


@pytest.mark.gpu
@pytest.mark.parametrize("gmap", [None, "0,0"])
def test_cdfpsi_vvl(case, gmap):
    d, g, rec, e3, _ = case
    npk = g["tmask"].shape[0]
    env = {} if gmap is None else {"CDFT_GPU_MAP": gmap}
    for args, kw in ((["-o", "p1v.nc"], dict(zu=True)), (["-V", "-o", "p2v.nc"], dict(zv=True)),
                     (["-mean", "-o", "p3v.nc"], dict(zu=True, zv=True)), (["-open", "-o", "p4v.nc"], dict(opn=True))):
        rc, out, err = run("cdfpsi", ["-u", "VVL_y0001_gridU.nc", "-v", "VVL_y0001_gridV.nc", "-vvl"] + args, d, env=env)
        assert rc == 0, out + err
        var, _, _ = read_nc(os.path.join(d, args[-1]))
        for t in range(2):
            if kw.get("opn"):
                want, _ = oracle_tools.psi_open(npk, rec["u"][t], g["e2u"], e3["u"][t], rec["v"][t], g["e1v"], e3["v"][t])
            else:
                o = {}
                if kw.get("zu"):
                    o.update(zu=rec["u"][t], e2u=g["e2u"], e3u=e3["u"][t])
                if kw.get("zv"):
                    o.update(zv=rec["v"][t], e1v=g["e1v"], e3v=e3["v"][t])
                want, _ = oracle_tools.psi(npk, **o)
            assert np.array_equal(var["sobarstf"][t], want), (args, t)
    rc, _, _ = run("cdfpsi", ["-u", "VVL_y0001_gridU.nc", "-v", "VVL_y0001_gridV.nc", "-o", "p1s.nc"], d)
    assert rc == 0 and differs(d, "p1v.nc", "p1s.nc", ["sobarstf"])


@pytest.mark.gpu
def test_cdfvhst_and_cdfvtrp_vvl(case):
    d, g, rec, e3, vt = case
    rc, out, err = run("cdfvhst", ["-f", "VVL_vt.nc", "-vvl", "-o", "vh_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "vh_vvl.nc"))
    for t in range(2):
        h, s = oracle_tools.vhst_transport(vt["vomevt"][t], vt["vomevs"][t], g["e1v"], e3=e3["v"][t])
        assert np.array_equal(var["somevt"][t], h.astype(f32)) and np.array_equal(var["somevs"][t], s.astype(f32))
        h, s = oracle_tools.vhst_transport(vt["vozout"][t], vt["vozous"][t], g["e2u"], e3=e3["u"][t])
        assert np.array_equal(var["sozout"][t], h.astype(f32)) and np.array_equal(var["sozous"][t], s.astype(f32))
    rc, _, _ = run("cdfvhst", ["-f", "VVL_vt.nc", "-o", "vh_static.nc"], d)
    assert rc == 0 and differs(d, "vh_vvl.nc", "vh_static.nc", ["somevt"])
    rc, out, err = run("cdfvtrp", ["-u", "VVL_y0001_gridU.nc", "-v", "VVL_y0001_gridV.nc", "-vvl", "-o", "tr_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "tr_vvl.nc"))
    for t in range(2):
        du = oracle_tools.vint_transport(rec["u"][t], g["e2u"], e3=e3["u"][t])
        dv = oracle_tools.vint_transport(rec["v"][t], g["e1v"], e3=e3["v"][t])
        assert np.array_equal(var["sozoutrp"][t], du.astype(f32)) and np.array_equal(var["somevtrp"][t], dv.astype(f32))
    # a file without the metrics: the reference's check never fires (lchkvar starts .FALSE. and is .AND.-ed, src/cdfvhst.f90:56,
    # :112-115), so the first read of the missing variable ends the program through ERR_HDL, STOP 98 (src/cdfio.F90:2937)
    rc, out, _ = run("cdfvhst", ["-f", "SYN_y0001_gridT.nc", "-vvl"], d)
    assert rc == 98 and "no vertical metrics" not in out


@pytest.mark.gpu
def test_cdfmean_and_cdfsum_vvl(case):
    d, g, rec, e3, _ = case
    npk, npj, npi = g["tmask"].shape
    tm = g["tmask"].astype(f32)
    one = np.ones((npj, npi, 1), np.int16)
    rc, out, err = run("cdfmean", ["-f", "VVL_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-var", "-S", "-vvl", "-o", "mean_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "mean_vvl.nc"))
    for t in range(2):
        s = oracle_tools.mean(g["e1t"], g["e2t"], tm, one, rec["t"][t], e3=e3["t"][t])[:, 0, :]
        ok = s[:, 1] != 0
        mean = np.where(ok, s[:, 2] / np.where(ok, s[:, 1], 1), 99999.0)
        assert np.allclose(var["mean_votemper"][t].ravel(), mean, rtol=2e-7, atol=0)
        assert np.allclose(var["sum_votemper"][t].ravel(), s[:, 2], rtol=2e-7, atol=0)
        assert np.allclose(var["mean_3Dvotemper"][t].ravel(), s[:, 2].sum() / s[:, 1].sum(), rtol=2e-7)
    rc, _, _ = run("cdfmean", ["-f", "VVL_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-var", "-S", "-o", "mean_static.nc"], d)
    assert rc == 0 and differs(d, "mean_vvl.nc", "mean_static.nc", ["sum_votemper"])
    # the V point takes e3v of the V file (cn_ve3vvvl, src/cdfmean.f90:332)
    rc, out, err = run("cdfmean", ["-f", "VVL_y0001_gridV.nc", "-v", "vomecrty", "-p", "V", "-S", "-vvl", "-o", "meanv_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "meanv_vvl.nc"))
    s = oracle_tools.mean(g["e1v"], g["e2v"], g["vmask"].astype(f32), one, rec["v"][1], e3=e3["v"][1])[:, 0, :]
    assert np.allclose(var["sum_vomecrty"][1].ravel(), s[:, 2], rtol=2e-7, atol=1e-30)

    rc, out, err = run("cdfsum", ["-f", "VVL_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-vvl", "-o", "sum_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "sum_vvl.nc"))
    for t in range(2):
        s = oracle_tools.mean(g["e1t"], g["e2t"], tm, one, rec["t"][t], e3=e3["t"][t], form=1)[:, 0, :]
        assert np.allclose(var["sum_votemper"][t].ravel(), np.where(s[:, 1] != 0, s[:, 2], 99999.0), rtol=2e-7, atol=0)
    # (no "differs" check for cdfsum: only its volume sum, which decides between a value and 99999., takes e3 -- the sum that
    # is written is zv*e1*e2, src/cdfsum.f90:319-323; the r8 level sums with the record's e3 are in tests/test_lift_vvl.py)


@pytest.mark.gpu
def test_cdfvint_and_cdfvertmean_vvl(case):
    d, g, rec, e3, _ = case
    tm = g["tmask"].astype(f32)
    sclf = f32(f32(1020.) * f32(4000.) / f32(1.e6))
    rc, out, err = run("cdfvint", ["-f", "VVL_y0001_gridT.nc", "-vvl", "-o", "vint_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "vint_vvl.nc"))
    gdepo = np.append(g["gdepw_1d"][1:], f32(6000.0)).astype(f32)
    for t in range(2):
        assert np.array_equal(var["voheatc"][t], oracle_tools.vint_cumul(rec["t"][t], tm, g["e3t_1d"], gdepo, sclf=sclf, e3t=e3["t"][t]))
    rc, _, _ = run("cdfvint", ["-f", "VVL_y0001_gridT.nc", "-o", "vint_static.nc"], d)
    assert rc == 0 and differs(d, "vint_vvl.nc", "vint_static.nc", ["voheatc"])
    rc, out, err = run("cdfvertmean", ["-f", "VVL_y0001_gridT.nc", "-l", "votemper", "-p", "T", "-zlim", "15", "120", "-vvl", "-o", "vm_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "vm_vvl.nc"))
    gdep = g["gdepw_1d"]
    ik1, ik2 = int(np.nonzero(gdep <= 15.0)[0].max()), int(np.nonzero(gdep <= 120.0)[0].max())
    sl = slice(ik1, ik2 + 1)
    for t in range(2):
        ref = oracle_tools.vertmean(rec["t"][t][sl], tm[sl], gdep[ik1], gdep[ik2], 15.0, 120.0, e3=e3["t"][t][sl])
        assert np.array_equal(var["votemper_vert_mean"][t], ref.astype(f32))
    rc, _, _ = run("cdfvertmean", ["-f", "VVL_y0001_gridT.nc", "-l", "votemper", "-p", "T", "-zlim", "15", "120", "-o", "vm_static.nc"], d)
    assert rc == 0 and differs(d, "vm_vvl.nc", "vm_static.nc", ["votemper_vert_mean"])


@pytest.mark.gpu
def test_cdfmocsig_vvl(case):
    d, g, rec, e3, _ = case
    e1v, _, ib = synth.moc_inputs(g, nbasins=5)
    rc, out, err = run("cdfmocsig", ["-v", "VVL_y0001_gridV.nc", "-t", "VVL_y0001_gridT.nc", "-r", "2000", "-vvl", "-o", "ms_vvl.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "ms_vvl.nc"))
    for t in range(2):
        ref = oracle_lib.mocsig(rec["v"][t], rec["t"][t], rec["s"][t], e1v, ib, 158, 30.0, 0.05, pref=2000.0, e3v=e3["v"][t])
        for b, nm in enumerate(["glo", "atl", "inp", "ind", "pac"]):
            assert np.allclose(var["zomsf" + nm][t, :, :, 0], ref[:, :, b].T, rtol=0, atol=2e-7 * np.abs(ref).max()), (nm, t)
    rc, _, _ = run("cdfmocsig", ["-v", "VVL_y0001_gridV.nc", "-t", "VVL_y0001_gridT.nc", "-r", "2000", "-o", "ms_static.nc"], d)
    assert rc == 0 and differs(d, "ms_vvl.nc", "ms_static.nc", ["zomsfglo"])


def test_cdfmoc_rapid_vvl(tools, tmp_path):  # noqa: F811
    """Host code (no GPU): -rapid reads the section's e3v of every record from the V file (src/cdfmoc.f90:669-672, :710-724)."""
    from test_cli_rapid import LAT, LONE, LONGS, LONW, WIND, make_rapid_case, nearest, numpy_rapid
    d = str(tmp_path)
    g, v, t, s, taux = make_rapid_case(d)
    e3v_t = lc.e3_records(g["e3v_0"], 2, 8010)
    nemo_files.write_data(os.path.join(d, "Vvvl.nc"), g, "depthv", {"vomecrty": v, "e3v": e3v_t})
    iiw, ij = nearest(g, LONW, LAT)
    iie, _ = nearest(g, LONE, LAT)
    iigs, _ = nearest(g, LONGS, LAT)
    rc, out, err = run("cdfmoc", ["-v", "Vvvl.nc", "-t", "T.nc", "-u", "U.nc", "-rapid", "-vvl"], d)
    assert rc == 99 and "Total maximum AMOC" in out, out + err
    var, _, _ = read_nc(os.path.join(d, "rapid_moc.nc"))
    for rec in range(2):
        ref = numpy_rapid(dict(g, e3v_0=e3v_t[rec]), v[rec], t[rec], s[rec], taux[rec], iiw, iie, iigs, ij)
        for nm, want in ref.items():
            got = var[nm][rec].ravel()
            if nm in WIND:
                assert np.allclose(got, want, rtol=2e-6, atol=0), nm
            else:
                assert np.array_equal(got, np.atleast_1d(want)), (nm, rec)
