"""End-to-end runs of the Fortran drop-in programs other than cdfmoc (cdfheatc, cdfsig0, cdfsigi,
cdfcurl, cdfpsi, cdfvT, cdfmhst) on NEMO-style NetCDF files, on the GPU: files are written and read
back with scipy (independent of the bundled backend), results are checked against the oracle."""
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

import nemo_files
import oracle_lib
import oracle_tools
from test_cli import read_nc, run, tools  # noqa: F401  (tools is a fixture)


@pytest.fixture(scope="module")
def case(tools, tmp_path_factory):  # noqa: F811
    d = tmp_path_factory.mktemp("nemo")
    g, rec = nemo_files.make_case(str(d), npi=26, npj=19, npk=6, npt=2)
    return d, g, rec


def read_global(path):
    with netcdf_file(path, "r", mmap=False) as f:
        return dict(f._attributes)


@pytest.mark.gpu
def test_cdfheatc_end_to_end(case):
    d, g, rec = case
    rc, out, err = run("cdfheatc", ["-f", "SYN_y0001_gridT.nc"], d)
    assert rc == 0, out + err
    assert "Total Heat content" in out and "Heat Content  at level" in out
    var, att, dims = read_nc(os.path.join(d, "heatc.nc"))
    scale = np.float64(np.float32(1020.0) * np.float32(4000.0))
    for t in range(2):
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], rec["t"][t], e3t=g["e3t_0"])
        assert np.allclose(var["heatc2d"][t, :, 0, 0], scale * ref[:, 2], rtol=1e-12, atol=0)
        assert np.isclose(var["heatc3d"][t, 0, 0], scale * ref[:, 2].sum(), rtol=1e-12)
        assert np.isclose(var["heatc3dpervol"][t, 0, 0], scale * ref[:, 2].sum() / ref[:, 1].sum(), rtol=1e-12)
    assert var["heatc2d"].dtype == np.dtype(">f8") and att["heatc3d"]["units"] == b"Joules"
    # zoom window + full step
    rc, out, _ = run("cdfheatc", ["-f", "SYN_y0001_gridT.nc", "-zoom", "3", "12", "2", "9", "2", "4", "-full", "-o", "hz.nc"], d)
    assert rc == 0, out
    vz, _, dz = read_nc(os.path.join(d, "hz.nc"))
    ref = oracle_tools.heatc(g["e1t"][1:9, 2:12], g["e2t"][1:9, 2:12], g["tmask"][1:4, 1:9, 2:12], rec["t"][0][1:4, 1:9, 2:12],
                             e31d=g["e3t_1d"][1:4])
    assert dz["depthw"] == 3 and np.allclose(vz["heatc2d"][0, :, 0, 0], scale * ref[:, 2], rtol=1e-12)


@pytest.mark.gpu
def test_cdfsig0_and_cdfsigi_end_to_end(case):
    d, g, rec = case
    rc, out, err = run("cdfsig0", ["-t", "SYN_y0001_gridT.nc"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "sig0.nc"))
    want = oracle_lib.sig_tool(rec["t"], rec["s"], 0.0, 0.0)
    assert np.array_equal(var["vosigma0"], want)                       # bit-identical r4
    assert att["vosigma0"]["long_name"] == b"Potential_density:sigma-0"
    assert np.array_equal(var["nav_lon"], g["glamf"]) and np.array_equal(var["deptht"], g["gdept_1d"])
    rc, out, err = run("cdfsigi", ["-t", "SYN_y0001_gridT.nc", "-r", "2000", "-teos10"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "sigi.nc"))
    assert np.array_equal(var["vosigmai"], oracle_lib.sig_tool(rec["t"], rec["s"], 2000.0, 0.0, teos10=True))
    assert att["vosigmai"]["long_name"] == b"Potential_density:refered to 2000 m"


@pytest.mark.gpu
def test_cdfcurl_end_to_end(case):
    d, g, rec = case
    # BASELINE config 0 command line
    rc, out, err = run("cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "1"], d)
    assert rc == 0, out + err
    var, att, dims = read_nc(os.path.join(d, "curl.nc"))
    want = np.stack([oracle_tools.curl(rec["u"][t, 0], rec["v"][t, 0], g["e1u"], g["e2v"], g["e1f"], g["e2f"], True)
                     for t in range(2)]).astype(np.float32)
    assert var["socurl"].ndim == 3                                     # one level: createvar makes it (t,y,x)
    assert np.array_equal(var["socurl"], want) and att["socurl"]["units"] == b"s-1"
    assert np.count_nonzero(want) > 10
    rc, out, err = run("cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "2-4",
                                   "-T", "-8", "-o", "ct.nc"], d)
    assert rc == 0, out + err
    var, _, dims = read_nc(os.path.join(d, "ct.nc"))
    want = np.stack([[oracle_tools.curl(rec["u"][t, k], rec["v"][t, k], g["e1u"], g["e2v"], g["e1f"], g["e2f"], True, True)
                      for k in (1, 2, 3)] for t in range(2)])
    assert var["socurlt"].dtype == np.dtype(">f8") and np.array_equal(var["socurlt"], want)
    assert np.array_equal(var["depthu"], g["gdept_1d"][1:4])


@pytest.mark.gpu
def test_cdfpsi_end_to_end(case):
    d, g, rec = case
    npk = g["tmask"].shape[0]
    for args, kw in ((["-o", "p1.nc"], dict(zu=True)), (["-V", "-o", "p2.nc"], dict(zv=True)),
                     (["-mean", "-mask", "-ref", "5", "4", "-o", "p3.nc"], dict(zu=True, zv=True, mask=True, ref=(5, 4)))):
        rc, out, err = run("cdfpsi", ["-u", "SYN_y0001_gridU.nc", "-v", "SYN_y0001_gridV.nc"] + args, d)
        assert rc == 0, out + err
        var, att, _ = read_nc(os.path.join(d, args[-1]))
        for t in range(2):
            o = {}
            if kw.get("zu"):
                o.update(zu=rec["u"][t], e2u=g["e2u"], e3u=g["e3u_0"])
            if kw.get("zv"):
                o.update(zv=rec["v"][t], e1v=g["e1v"], e3v=g["e3v_0"])
            if kw.get("mask"):
                o.update(fmask=np.minimum(g["fmask"][0], 1).astype(np.float32))
            if kw.get("ref"):
                o.update(iiref=kw["ref"][0], ijref=kw["ref"][1])
            want, _ = oracle_tools.psi(npk, **o)
            assert np.array_equal(var["sobarstf"][t], want), args
    assert b"cdfpsi" in read_global(os.path.join(d, "p1.nc"))["history"]


@pytest.mark.gpu
def test_cdfvT_then_cdfmhst_end_to_end(case):
    d, g, rec = case
    rc, out, err = run("cdfvT", ["-c", "SYN", "-l", "y0001"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "vt.nc"))
    npk = g["tmask"].shape[0]
    for k in range(npk):
        ref = oracle_tools.vt_level(rec["u"][:, k], rec["v"][:, k], rec["t"][:, k], rec["s"][:, k])
        for nm, key in (("vomevt", "vt"), ("vomevs", "vs"), ("vozout", "ut"), ("vozous", "us")):
            assert np.array_equal(var[nm][0, k], ref[key].astype(np.float32)), (nm, k)
    assert att["vomevt"]["iweight"] == 2 and var["time_counter"][0] == 86400.0 * 45
    rc, out, err = run("cdfmhst", ["-vt", "vt.nc", "-MST"], d)
    assert rc == 0, out + err
    mv, ma, md = read_nc(os.path.join(d, "mhst.nc"))
    masks = np.stack([g["vmask"][0], g["tmaskatl"], g["tmaskpac"], g["tmaskind"]]).astype(np.float32)
    heat, salt = oracle_tools.mhst(g["e1v"], masks, e3v=g["e3v_0"], vt=var["vomevt"][0], vs=var["vomevs"][0])
    bh, bs = oracle_tools.mhst(g["e1v"], masks, e3v=g["e3v_0"], vt=np.abs(var["vomevt"][0]), vs=np.abs(var["vomevs"][0]))

    def col(x, b):
        return {"glo": x[:, 0], "atl": x[:, 1], "inp": x[:, 3] + x[:, 2], "ind": x[:, 3], "pac": x[:, 2], "inp0": x[:, 0] - x[:, 1]}[b]

    for b in ("glo", "atl", "inp", "ind", "pac", "inp0"):
        for nm, x, bound, div in (("zomht_", heat, bh, 1e15), ("zomst_", salt, bs, 1e6)):
            want = col(x, b) / div
            want = np.where(want == 0, 9999.99, want)
            got = mv[nm + b][0, :, 0]                                    # npko = 1: (t,y,x)
            tol = 1e-13 * bound.sum(axis=1) / div + np.spacing(np.abs(want).astype(np.float32))
            assert np.all(np.abs(got - want) <= tol), nm + b
    assert ma["zomht_glo"]["units"] == b"PW" and md["depthv"] == 1
    txt = open(os.path.join(d, "zonal_heat_trp.dat")).read().splitlines()
    assert len(txt) == 3 + g["e1v"].shape[0] and "Atlantic" in txt[1]
