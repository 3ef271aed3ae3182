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
        # the program returns the reference's own r4 sequential SUMs (src/cdfheatc.f90:225-229), bit for bit
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], rec["t"][t], e3t=g["e3t_0"], r4sum=True)
        dvol = dsum = 0.0
        for k in range(ref.shape[0]):
            dvol, dsum = dvol + ref[k, 1], dsum + ref[k, 2]
        assert np.array_equal(var["heatc2d"][t, :, 0, 0], scale * ref[:, 2])
        assert var["heatc3d"][t, 0, 0] == scale * dsum and var["heatc3dpervol"][t, 0, 0] == scale * dsum / dvol
    rc, out, err = run("cdfheatc", ["-f", "SYN_y0001_gridT.nc", "-o", "heatc_r8.nc"], d, env={"CDFT_HEATC_SUM": "r8"})
    assert rc == 0, out + err
    v8, _, _ = read_nc(os.path.join(d, "heatc_r8.nc"))
    for t in range(2):                                   # CDFT_HEATC_SUM=r8: r8 sums of the same r4 products
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], rec["t"][t], e3t=g["e3t_0"])
        assert np.allclose(v8["heatc2d"][t, :, 0, 0], scale * ref[:, 2], rtol=1e-12, atol=0)
        assert np.isclose(v8["heatc3d"][t, 0, 0], scale * ref[:, 2].sum(), rtol=1e-12)
        assert np.isclose(v8["heatc3dpervol"][t, 0, 0], scale * ref[:, 2].sum() / ref[:, 1].sum(), rtol=1e-12)
    assert var["heatc2d"].dtype == np.dtype(">f8") and att["heatc3d"]["units"] == b"Joules"
    # zoom window + full step
    rc, out, _ = run("cdfheatc", ["-f", "SYN_y0001_gridT.nc", "-zoom", "3", "12", "2", "9", "2", "4", "-full", "-o", "hz.nc"], d)
    assert rc == 0, out
    vz, _, dz = read_nc(os.path.join(d, "hz.nc"))
    c = np.ascontiguousarray
    ref = oracle_tools.heatc(c(g["e1t"][1:9, 2:12]), c(g["e2t"][1:9, 2:12]), c(g["tmask"][1:4, 1:9, 2:12]), c(rec["t"][0][1:4, 1:9, 2:12]),
                             e31d=g["e3t_1d"][1:4], r4sum=True)
    assert dz["depthw"] == 3 and np.array_equal(vz["heatc2d"][0, :, 0, 0], scale * ref[:, 2])


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args", [("cdfheatc", ["-f", "SYN_y0001_gridT.nc", "-o", "ht.nc"]), ("cdfsig0", ["-t", "SYN_y0001_gridT.nc", "-o", "s0t.nc"]),
                                       ("cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "1", "-o", "ct.nc"])])
def test_timing_switch_reports_on_stderr_only(case, tool, args):
    """CDFT_TIMING=1 (a phase-time switch that must not change stdout): cdfio's read / write totals and the library's
    per-entry wall times go to stderr; stdout and the output file are those of a run without the switch."""
    d, g, rec = case
    rc0, out0, err0 = run(tool, args, d)
    before = open(os.path.join(d, args[-1]), "rb").read()
    rc1, out1, err1 = run(tool, args, d, env={"CDFT_TIMING": "1"})
    assert rc0 == 0 and rc1 == 0, out1 + err1
    assert out1 == out0 and "CDFT_TIMING" not in err0
    assert "CDFT_TIMING cdfio (s): reads" in err1 and "CDFT_TIMING library: cdft_" in err1, err1
    var0, var1 = read_nc_bytes(before, d), read_nc(os.path.join(d, args[-1]))[0]
    assert all(np.array_equal(var0[k], var1[k], equal_nan=True) for k in var0)


def read_nc_bytes(blob, d):
    path = os.path.join(d, "_before.nc")
    open(path, "wb").write(blob)
    return read_nc(path)[0]


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
    # the reference names the depth by its option parser's scratch string (src/cdfsigi.f90:98, :166): the value of -r only when
    # no value-less option follows it -- kept file for file (the reference's own program writes the same:
    # tests/golden/prog_cdfsigi_teos10.npz)
    assert att["vosigmai"]["long_name"] == b"Potential_density:refered to -teos10 m"
    rc, out, err = run("cdfsigi", ["-teos10", "-o", "sigi_b.nc", "-t", "SYN_y0001_gridT.nc", "-r", "2000"], d)
    assert rc == 0, out + err
    _, att, _ = read_nc(os.path.join(d, "sigi_b.nc"))
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


@pytest.mark.gpu
def test_cdfpsi_open_ssh_and_lev_options(case):
    """-open [-mean], --ssh-file (with the reference's zv-for-zu operand, src/cdfpsi.f90:299) and -lev
    (first record, one-level file as the reference creates it, :311-315 and :497)."""
    d, g, rec = case
    npk, npj, npi = g["tmask"].shape
    rng = np.random.default_rng(77)
    ssh = (rng.standard_normal((2, npj, npi)) * g["tmask"][0]).astype(np.float32)
    nemo_files.write_data(os.path.join(d, "SYN_y0001_grid2D.nc"), g, "deptht", {"sossheig": ssh})
    uv = ["-u", "SYN_y0001_gridU.nc", "-v", "SYN_y0001_gridV.nc"]
    fm = np.minimum(g["fmask"][0], 1).astype(np.float32)

    def transports(t):
        return dict(zu=rec["u"][t], e2u=g["e2u"], e3u=g["e3u_0"], zv=rec["v"][t], e1v=g["e1v"], e3v=g["e3v_0"])

    for extra, lmean in ((["-open"], False), (["-open", "-mean"], True)):
        rc, out, err = run("cdfpsi", uv + extra + ["-ref", "5", "4", "-mask", "-o", "po.nc"], d)
        assert rc == 0, out + err
        var, _, _ = read_nc(os.path.join(d, "po.nc"))
        for t in range(2):
            want, _ = oracle_tools.psi_open(npk, **transports(t), lmean=lmean, iiref=5, ijref=4, fmask=fm)
            assert np.array_equal(var["sobarstf"][t], want), extra
    # --ssh-file with -mean: three variables
    rc, out, err = run("cdfpsi", uv + ["-mean", "--ssh-file", "SYN_y0001_grid2D.nc", "-o", "ps.nc"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "ps.nc"))
    assert att["sobarstfssh"]["long_name"] == b"Barotropic_Stream_Function SSH contribution"
    for t in range(2):
        sshu, sshv = np.zeros((npj, npi), np.float32), np.zeros((npj, npi), np.float32)
        sshu[:, :-1] = np.float32(0.5) * (ssh[t][:, 1:] + ssh[t][:, :-1])
        sshv[:-1, :] = np.float32(0.5) * (ssh[t][1:, :] + ssh[t][:-1, :])
        v1 = rec["v"][t][0:1]                                             # level-1 v multiplies BOTH parts (:280, :299)
        _, d8 = oracle_tools.psi(npk, **transports(t))
        p4s, d8s = oracle_tools.psi(1, zu=v1, e2u=g["e2u"], e3u=sshu[None], zv=v1, e1v=g["e1v"], e3v=sshv[None])
        # the SSH part is not referenced: undo the oracle's subtraction of its value at (npi,npj)
        _, d8s0 = oracle_tools.psi(1, zu=v1, e2u=g["e2u"], e3u=sshu[None], zv=v1, e1v=g["e1v"], e3v=sshv[None], iiref=npi, ijref=1)
        assert np.array_equal(var["sobarstf"][t], d8.astype(np.float32))
        got_ssh = var["sobarstfssh"][t]
        # reconstruct the unreferenced field from two differently referenced oracle runs: both are dpsi - const
        assert np.allclose(got_ssh - got_ssh[-1, -1], d8s.astype(np.float32), rtol=0, atol=np.abs(d8s).max() * 3e-7)
        assert np.allclose(var["sobarstftotal"][t], got_ssh.astype(np.float64) + d8, rtol=0, atol=np.abs(d8).max() * 2e-7)
        assert np.abs(got_ssh).max() > 0
    # without the V component the reference's zv is never assigned: the SSH part is 0 there and here
    rc, out, err = run("cdfpsi", uv + ["--ssh-file", "SYN_y0001_grid2D.nc", "-o", "ps0.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "ps0.nc"))
    assert np.all(var["sobarstfssh"] == 0) and np.array_equal(var["sobarstftotal"], var["sobarstf"])
    # -lev
    rc, out, err = run("cdfpsi", uv + ["-lev", "-o", "pl.nc"], d)
    assert rc == 0 and "cdfpsi_level completed." in out, out + err
    var, att, dims = read_nc(os.path.join(d, "pl.nc"))
    assert att["vobarstf"]["long_name"] == b"Barotropic_Stream_Function_per_level"
    lev = var["vobarstf"]
    lev = lev.reshape(lev.shape[0], -1, npj, npi)                      # (t, 1, y, x): the file holds one level
    assert lev.shape[1] == 1
    _, want = oracle_tools.psi(1, zu=rec["u"][0][0:1], e2u=g["e2u"], e3u=g["e3u_0"][0:1], iiref=1, ijref=1)
    assert np.array_equal(lev[0, 0], want.astype(np.float32))
    rc, out, _ = run("cdfpsi", uv + ["-lev", "-open"], d)
    assert rc == 99 and "cannot use both -lev and -open" in out


@pytest.mark.gpu
def test_cdfcurl_level_0_forcing_fields_on_the_a_grid(case):
    """-l 0: the fields are taken as A-grid forcing, averaged to U / V points (src/cdfcurl.f90:250-258), then
    the same stencil.  As in the reference the test ':154' comes first: a file without depth stops with
    'Use -surf option', so level 0 is only reachable on a file that has a depth dimension."""
    d, g, rec = case
    npk, npj, npi = g["tmask"].shape
    rc, out, err = run("cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "0",
                                   "-o", "cf.nc"], d)
    assert rc == 0 and "npk =0, assume 1" in out, out + err
    var, _, _ = read_nc(os.path.join(d, "cf.nc"))
    for t in range(2):
        tx, ty = rec["u"][t, 0], rec["v"][t, 0]
        un, vn = tx.copy(), ty.copy()
        un[:, :-1] = np.float32(0.5) * (tx[:, :-1] + tx[:, 1:])
        vn[:-1, :] = np.float32(0.5) * (ty[:-1, :] + ty[1:, :])
        want = oracle_tools.curl(un, vn, g["e1u"], g["e2v"], g["e1f"], g["e2f"], True).astype(np.float32)
        assert np.array_equal(var["socurl"][t].reshape(npj, npi), want)
        assert np.count_nonzero(want) > 20
    with netcdf_file(os.path.join(d, "flx.nc"), "w", version=2) as f:       # no depth dimension at all
        f.createDimension("time_counter", None)
        f.createDimension("y", npj)
        f.createDimension("x", npi)
        for nm in ("sozotaux", "sometauy"):
            v = f.createVariable(nm, "f4", ("time_counter", "y", "x"))
            v[:] = np.zeros((1, npj, npi), np.float32)
        v = f.createVariable("time_counter", "f8", ("time_counter",))
        v[:] = [86400.0]
    rc, out, _ = run("cdfcurl", ["-u", "flx.nc", "sozotaux", "-v", "flx.nc", "sometauy", "-l", "0"], d)
    assert rc == 99 and "Use -surf option" in out
