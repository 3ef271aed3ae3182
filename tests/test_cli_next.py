"""The first 'next' tools of SURVEY.md 8f rank 2, vertically integrated transports per grid cell:
cdfvhst (src/cdfvhst.f90) and cdfvtrp (src/cdfvtrp.f90).  Kernel parity through the C ABI (bit-exact:
r4 products as written, r8 sums in level order) and whole runs of the Fortran programs."""
import os

import numpy as np
import pytest

import nemo_files
import oracle_tools
import synth
from test_cli import read_nc, run, tools  # noqa: F401
from test_transport import tracer_fields


def test_oracle_vint_and_vhst_match_numpy_restatement_bitwise():
    g = synth.make_grid(19, 13, 5, seed=4)
    t, s, u, v = tracer_fields(g, 1)
    want = np.zeros(g["e2u"].shape)
    for k in range(5):
        want += ((u[k] * g["e2u"]) * g["e3u_0"][k]).astype(np.float64)
    assert np.array_equal(oracle_tools.vint_transport(u, g["e2u"], e3=g["e3u_0"]), want)
    wh, ws = np.zeros(g["e1v"].shape), np.zeros(g["e1v"].shape)
    for k in range(5):
        em = (g["e3v_0"][k] * g["e1v"]).astype(np.float32)
        wh += (((t[k] * em).astype(np.float32) * np.float32(1000)).astype(np.float32) * np.float32(4000)).astype(np.float64)
        ws += (s[k] * em).astype(np.float64)
    h, sl = oracle_tools.vhst_transport(t, s, g["e1v"], e3=g["e3v_0"])
    assert np.array_equal(h, wh) and np.array_equal(sl, ws)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk", [(24, 18, 7), (182, 149, 31), (1, 1, 1), (5, 3, 2), (1023, 7, 3)])
def test_hip_vint_and_vhst_equal_oracle_bitwise(gpu, npi, npj, npk):
    import torch

    g = synth.make_grid(npi, npj, npk, seed=npi + 2)
    t, s, u, v = tracer_fields(g, 3)
    assert np.array_equal(gpu.vint_transport(u, g["e2u"], e3=g["e3u_0"]), oracle_tools.vint_transport(u, g["e2u"], e3=g["e3u_0"]))
    assert np.array_equal(gpu.vint_transport(v, g["e1v"], e31d=g["e3t_1d"]), oracle_tools.vint_transport(v, g["e1v"], e31d=g["e3t_1d"]))
    for kw in (dict(e3=g["e3v_0"]), dict(e31d=g["e3t_1d"])):
        gh, gs = gpu.vhst_transport(t, s, g["e1v"], **kw)
        rh, rs = oracle_tools.vhst_transport(t, s, g["e1v"], **kw)
        assert np.array_equal(gh, rh) and np.array_equal(gs, rs)
    dev = [torch.from_numpy(x).cuda() for x in (t, s, g["e1v"], g["e3v_0"])]
    dh, ds = gpu.vhst_transport(dev[0], dev[1], dev[2], e3=dev[3])            # resident arrays
    rh, rs = oracle_tools.vhst_transport(t, s, g["e1v"], e3=g["e3v_0"])
    assert np.array_equal(dh.cpu().numpy(), rh) and np.array_equal(ds.cpu().numpy(), rs)


@pytest.mark.parametrize("tool,flag", [("cdfvhst", "-f"), ("cdfvtrp", "-u")])
def test_usage_and_unknown_option(tools, tmp_path, tool, flag):  # noqa: F811
    rc, out, _ = run(tool, [], tmp_path)
    assert rc == 0 and "usage :" in out.splitlines()[0] and tool in out.splitlines()[0]
    rc, out, _ = run(tool, ["-thisisnotanoption"], tmp_path)
    assert rc == 99 and "unknown option" in out
    if tool == "cdfvtrp":
        rc, out, _ = run(tool, ["-u", "U.nc", "-v", "V.nc"], tmp_path)
        assert rc == 99 and "missing" in out


@pytest.mark.gpu
def test_cdfvhst_and_cdfvtrp_end_to_end(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    npk = 6
    rc, out, err = run("cdfvT", ["-c", "SYN", "-l", "y0001"], d)
    assert rc == 0, out + err
    vt, _, _ = read_nc(os.path.join(d, "vt.nc"))
    rc, out, err = run("cdfvhst", ["-f", "vt.nc", "-o", "vh.nc"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "vh.nc"))
    h, s = oracle_tools.vhst_transport(vt["vomevt"][0], vt["vomevs"][0], g["e1v"], e3=g["e3v_0"])
    assert np.array_equal(var["somevt"][0], h.astype(np.float32)) and np.array_equal(var["somevs"][0], s.astype(np.float32))
    h, s = oracle_tools.vhst_transport(vt["vozout"][0], vt["vozous"][0], g["e2u"], e3=g["e3u_0"])
    assert np.array_equal(var["sozout"][0], h.astype(np.float32)) and np.array_equal(var["sozous"][0], s.astype(np.float32))
    assert att["somevt"]["units"] == b"W" and att["sozous"]["long_name"] == b"Zonal_salt_transport"
    rc, out, err = run("cdfvhst", ["-f", "vt.nc", "-full", "-o", "vhf.nc"], d)
    assert rc == 0, out + err
    varf, _, _ = read_nc(os.path.join(d, "vhf.nc"))
    h, _ = oracle_tools.vhst_transport(vt["vomevt"][0], vt["vomevs"][0], g["e1v"], e31d=g["e3t_1d"])
    assert np.array_equal(varf["somevt"][0], h.astype(np.float32))

    rc, out, err = run("cdfvtrp", ["-u", "SYN_y0001_gridU.nc", "-v", "SYN_y0001_gridV.nc", "-bathy", "-o", "tr.nc"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "tr.nc"))
    tm = g["tmask"][0].astype(np.float32)
    hdepw = (g["e3t_0"] * g["tmask"]).sum(axis=0).astype(np.float32)
    for t in range(2):
        du = oracle_tools.vint_transport(rec["u"][t], g["e2u"], e3=g["e3u_0"])
        dv = oracle_tools.vint_transport(rec["v"][t], g["e1v"], e3=g["e3v_0"])
        assert np.array_equal(var["sozoutrp"][t], du.astype(np.float32)) and np.array_equal(var["somevtrp"][t], dv.astype(np.float32))
        # -bathy (src/cdfvtrp.f90:224-259): rotation onto the bathymetric slope, through the host's libm
        wu, wv = np.zeros_like(du), np.zeros_like(dv)
        wu[:, 1:] = 0.5 * (du[:, 1:] + du[:, :-1])
        wu[:, 0] = wu[:, -2]
        wv[1:, :] = 0.5 * (dv[1:, :] + dv[:-1, :])
        dhdx, dhdy = np.zeros_like(tm), np.zeros_like(tm)
        dhdx[:, 1:-1] = (hdepw[:, 2:] - hdepw[:, :-2]) / (g["e1u"][:, 1:-1] + g["e1u"][:, :-2]) * tm[:, 1:-1]
        dhdy[1:-1, :] = (hdepw[2:, :] - hdepw[:-2, :]) / (g["e2v"][1:-1, :] + g["e2v"][:-2, :]) * tm[1:-1, :]
        al = (np.arctan2(dhdx, dhdy) * tm).astype(np.float32)
        along = (wu * np.cos(al) + wv * np.sin(al)) * tm
        cross = (-wu * np.sin(al) + wv * np.cos(al)) * tm
        scale = np.abs(along).max()
        assert np.allclose(var["soastrp"][t], along, rtol=0, atol=2e-6 * scale)
        assert np.allclose(var["socstrp"][t], cross, rtol=0, atol=2e-6 * scale)
    assert att["soastrp"]["long_name"] == b"Along_Slope_Barotropic_Transp"


# ------------------------------------------------------------------------------------------- cdfmean
def mean_inputs(npi, npj, npk, nb, seed):
    g = synth.make_grid(npi, npj, npk, seed=seed)
    rng = np.random.default_rng(seed + 3)
    zv = ((-2 + 30 * rng.random((npk, npj, npi), dtype=np.float32)) * g["tmask"]).astype(np.float32)
    zv[g["tmask"] == 0] = 1.0e20                                           # funny fill values on land (:410)
    ib = np.zeros((npj, npi, nb), np.int16)
    ib[:, :, 0] = 1
    for b, nm in zip(range(1, nb), ("tmaskatl", "tmaskpac", "tmaskind")):
        ib[:, :, b] = g[nm]
    return g, zv, ib


def test_oracle_mean_matches_numpy_restatement():
    g, zv, ib = mean_inputs(19, 13, 5, 3, seed=2)
    tm = g["tmask"].astype(np.float32)
    got = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, zv, e3=g["e3t_0"])
    f32 = np.float32
    zm = (zv * tm).astype(f32)
    for b in range(3):
        fb = ib[:, :, b].astype(f32)
        a = (g["e1t"] * g["e2t"]).astype(f32)
        want = np.stack([((a * tm).astype(f32) * fb).astype(np.float64).sum(axis=(1, 2)),
                         (((a * g["e3t_0"]).astype(f32) * tm).astype(f32) * fb).astype(np.float64).sum(axis=(1, 2)),
                         ((((zm * g["e1t"]).astype(f32) * g["e2t"]).astype(f32) * g["e3t_0"]).astype(f32) * fb).astype(np.float64).sum(axis=(1, 2)),
                         (((((zm * zm).astype(f32) * g["e1t"]).astype(f32) * g["e2t"]).astype(f32) * g["e3t_0"]).astype(f32) * fb)
                         .astype(np.float64).sum(axis=(1, 2))], axis=1)
        assert np.allclose(got[:, b, :], want, rtol=1e-13, atol=0)
    assert np.all(np.isfinite(got))


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 4), (182, 149, 31, 1), (1, 1, 1, 1), (5, 3, 2, 8), (1023, 7, 3, 3), (64, 64, 2, 2),
                                            (40, 9, 96, 4), (33, 5, 48, 8), (21, 4, 192, 2), (30, 6, 95, 4)])   # npk * nb = 384: the LDS budget's edge
def test_hip_mean_matches_oracle(gpu, monkeypatch, npi, npj, npk, nb):
    g, zv, ib = mean_inputs(npi, npj, npk, min(nb, 4), seed=npi + 1)
    if nb > 4:
        ib = np.concatenate([ib, ib[:, :, :nb - 4] * 2], axis=2)                      # values other than 0/1 are allowed
    tm = g["tmask"].astype(np.float32)
    plan = gpu.MeanPlan(npi, npj, npk, nb)
    for kw in (dict(e3=g["e3t_0"]), dict(e31d=g["e3t_1d"])):
        plan.set_mesh(g["e1t"], g["e2t"], tm, ib, **kw)
        got = plan.compute(zv)
        ref = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, zv, **kw)
        bound = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, np.abs(zv), **kw)
        assert np.all(np.abs(got - ref) <= 1e-13 * bound + 1e-300), float(np.abs(got - ref).max())
        assert np.array_equal(plan.compute(zv), got)                                    # run-to-run identical
        monkeypatch.setenv("CDFT_MEAN_R4MASK", "1")                                     # the 0/1 mask as r4 instead of bytes
        assert np.array_equal(plan.compute(zv), got)
        monkeypatch.delenv("CDFT_MEAN_R4MASK")
        # two vectors per lane instead of four: other runs of cells per lane and block, the r8 sums regrouped within the bound
        monkeypatch.setenv("CDFT_MEAN_VPT", "2")
        got2 = plan.compute(zv)
        assert np.all(np.abs(got2 - ref) <= 1e-13 * bound + 1e-300)
        for form in (1, 2, 3):                                                          # cdfsum's two forms, cdfmean without -var
            plan.set_form(form)
            a2 = plan.compute(zv)
            monkeypatch.setenv("CDFT_MEAN_VPT", "4")
            a4 = plan.compute(zv)
            monkeypatch.setenv("CDFT_MEAN_VPT", "2")
            assert np.all(np.abs(a2 - a4) <= 2e-13 * bound + 1e-300) and np.array_equal(a2[..., 3], 0 * a2[..., 3])
        plan.set_form(0)
        monkeypatch.delenv("CDFT_MEAN_VPT")
    tm2 = tm * np.float32(0.5)                                                          # not 0/1: never packed
    plan.set_mesh(g["e1t"], g["e2t"], tm2, ib, e3=g["e3t_0"])
    got = plan.compute(zv)
    ref = oracle_tools.mean(g["e1t"], g["e2t"], tm2, ib, zv, e3=g["e3t_0"])
    bound = oracle_tools.mean(g["e1t"], g["e2t"], tm2, ib, np.abs(zv), e3=g["e3t_0"])
    assert np.all(np.abs(got - ref) <= 1e-13 * bound + 1e-300)


@pytest.mark.gpu
def test_cdfmean_end_to_end(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    npk, npj, npi = g["tmask"].shape
    tm = g["tmask"].astype(np.float32)
    one = np.ones((npj, npi, 1), np.int16)

    def nc(name):
        return read_nc(os.path.join(d, name))

    rc, out, err = run("cdfmean", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-var", "-S"], d)
    assert rc == 0, out + err
    var, att, dims = nc("cdfmean.nc")
    assert dims["x"] == 1 and dims["y"] == 1
    for t in range(2):
        s = oracle_tools.mean(g["e1t"], g["e2t"], tm, one, rec["t"][t], e3=g["e3t_0"])[:, 0, :]
        mean = np.where(s[:, 1] != 0, s[:, 2] / np.where(s[:, 1] != 0, s[:, 1], 1), 99999.0)
        vari = np.where(s[:, 1] != 0, s[:, 3] / np.where(s[:, 1] != 0, s[:, 1], 1) - mean * mean, 99999.0)
        assert np.allclose(var["mean_votemper"][t].ravel(), mean, rtol=2e-7, atol=0)
        assert np.allclose(var["var_votemper"][t].ravel(), vari, rtol=1e-5, atol=1e-6)
        assert np.allclose(var["sum_votemper"][t].ravel(), s[:, 2], rtol=2e-7, atol=0)
        assert np.allclose(var["mean_3Dvotemper"][t].ravel(), s[:, 2].sum() / s[:, 1].sum(), rtol=2e-7)
        assert np.allclose(var["sum_3Dvotemper"][t].ravel(), s[:, 2].sum(), rtol=2e-7)
    assert "Mean value over the ocean:" in out and os.path.exists(os.path.join(d, "cdfvar.txt"))
    txt = open(os.path.join(d, "cdfmean.txt")).read().splitlines()
    assert len(txt) == 2 * (npk + 1) - 2 * int((g["tmask"][-1] == 0).all())          # a level without water writes no line
    # window, full step, sub-basins (followed by another option, as the reference's list parser needs)
    rc, out, err = run("cdfmean", ["-f", "SYN_y0001_gridT.nc", "-v", "vosaline", "-p", "T", "-w", "3", "20", "2", "15", "2", "4",
                                   "-B", "new_maskglo.nc", "tmaskatl", "tmaskpac", "-full", "-o", "mw.nc", "-P", "pre_"], d)
    assert rc == 0, out + err
    var, att, dims = nc("pre_mw.nc")
    ib = np.stack([g["tmaskatl"], g["tmaskpac"]], axis=2).astype(np.int16)[1:15, 2:20]
    s = oracle_tools.mean(g["e1t"][1:15, 2:20], g["e2t"][1:15, 2:20], tm[1:4, 1:15, 2:20], ib, rec["s"][0][1:4, 1:15, 2:20],
                          e31d=g["e3t_1d"][0:3])                                       # e31d(jk), jk = 1.. (:412)
    for b, nm in enumerate(("tmaskatl", "tmaskpac")):
        mean = np.where(s[:, b, 1] != 0, s[:, b, 2] / np.where(s[:, b, 1] != 0, s[:, b, 1], 1), 99999.0)
        assert np.allclose(var["mean_vosaline_" + nm][0].ravel(), mean, rtol=2e-7), nm
    assert os.path.exists(os.path.join(d, "pre_cdfmean.txt"))
    # another mask file and variable (-M), V point
    rc, out, err = run("cdfmean", ["-f", "SYN_y0001_gridV.nc", "-v", "vomecrty", "-p", "V", "-M", "mask.nc", "umask", "-o", "mm.nc"], d)
    assert rc == 0, out + err
    var, _, _ = nc("mm.nc")
    s = oracle_tools.mean(g["e1v"], g["e2v"], g["umask"].astype(np.float32), one, rec["v"][1], e3=g["e3v_0"])[:, 0, :]
    good = s[:, 1] != 0
    assert np.allclose(var["mean_vomecrty"][1].ravel()[good], (s[:, 2] / np.where(good, s[:, 1], 1))[good], rtol=5e-6, atol=1e-9)
    # -zeromean: the field minus the 3-D mean of each record, masked, for the global domain and per basin
    rc, out, err = run("cdfmean", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-zeromean", "-oz", "zm.nc",
                                   "-B", "new_maskglo.nc", "tmaskatl", "tmaskind", "-o", "mz.nc"], d)
    assert rc == 0, out + err
    var, att, dims = nc("zm.nc")
    mvar, _, _ = nc("mz.nc")
    assert dims["x"] == npi and dims["y"] == npj and dims["deptht"] == npk
    for nm in ("tmaskatl", "tmaskind"):
        ibm = g[nm].astype(np.float64)
        for t in range(2):
            s = oracle_tools.mean(g["e1t"], g["e2t"], tm, g[nm][:, :, None].astype(np.int16), rec["t"][t], e3=g["e3t_0"])[:, 0, :]
            m3 = s[:, 2].sum() / s[:, 1].sum()
            assert np.allclose(mvar["mean_3Dvotemper_" + nm][t].ravel(), m3, rtol=2e-7)
            want = ((rec["t"][t].astype(np.float64) - m3) * tm * ibm[None]).astype(np.float32)
            assert np.allclose(var["votemper_" + nm][t], want, rtol=0, atol=4e-6), nm
            assert np.abs(var["votemper_" + nm][t]).max() > 1
    rc, out, _ = run("cdfmean", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "Q"], d)
    assert rc == 99 and "not known" in out


# ------------------------------------------------------------------------------------------------ cdfsum
def test_oracle_sum_forms_match_numpy_restatement():
    g, zv, ib = mean_inputs(19, 13, 5, 1, seed=2)
    tm = (g["tmask"] * np.float32(0.5)).astype(np.float32)                 # a -M mask with values other than 0/1
    f32 = np.float32
    zm = (zv * tm).astype(f32)
    zee = ((zm * g["e1t"]).astype(f32) * g["e2t"]).astype(f32)
    want = {1: zee, 2: (zee * tm).astype(f32)}
    for form in (1, 2):
        got = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, zv, e3=g["e3t_0"], form=form)[:, 0, :]
        ref = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, zv, e3=g["e3t_0"])[:, 0, :]
        assert np.array_equal(got[:, :2], ref[:, :2]) and np.all(got[:, 3] == 0)
        assert np.allclose(got[:, 2], want[form].astype(np.float64).sum(axis=(1, 2)), rtol=1e-13, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk", [(24, 18, 7), (182, 149, 5), (1, 1, 1), (1023, 7, 3), (64, 64, 2)])
def test_hip_sum_forms_match_oracle(gpu, npi, npj, npk):
    g, zv, ib = mean_inputs(npi, npj, npk, 1, seed=npi + 4)
    tm = (g["tmask"] * np.float32(0.5)).astype(np.float32)
    plan = gpu.MeanPlan(npi, npj, npk, 1)
    plan.set_mesh(g["e1t"], g["e2t"], tm, ib, e3=g["e3t_0"])
    for form in (1, 2, 0):
        plan.set_form(form)
        got = plan.compute(zv)
        ref = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, zv, e3=g["e3t_0"], form=form)
        bound = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, np.abs(zv), e3=g["e3t_0"], form=form)
        assert np.all(np.abs(got - ref) <= 1e-13 * bound + 1e-300), form
    with pytest.raises(gpu.CdftError):
        plan.set_form(4)                                                  # (3 = cdfmean without the variance sum)


def test_cdfsum_usage_and_option_errors(tools, tmp_path):  # noqa: F811
    rc, out, _ = run("cdfsum", [], tmp_path)
    assert rc == 0 and "usage : cdfsum" in out.splitlines()[0]
    rc, out, _ = run("cdfsum", ["-full"], tmp_path)                        # announced by the usage, unknown to the parser
    assert rc == 99 and "unknown option" in out
    rc, out, _ = run("cdfsum", ["-f", "x.nc"], tmp_path)
    assert rc == 99 and "-v option" in out and "-p option" in out
    rc, out, _ = run("cdfsum", ["-f", "nothere.nc", "-v", "votemper", "-p", "T"], tmp_path)
    assert rc == 99 and "missing" in out


@pytest.mark.gpu
def test_cdfsum_end_to_end(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    npk, npj, npi = g["tmask"].shape
    tm = g["tmask"].astype(np.float32)
    one = np.ones((npj, npi, 1), np.int16)

    rc, out, err = run("cdfsum", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "T"], d)
    assert rc == 0, out + err
    var, att, dims = read_nc(os.path.join(d, "cdfsum.nc"))
    assert dims["x"] == 1 and dims["y"] == 1 and dims["deptht"] == npk
    cum = 0.0
    for t in range(2):
        s = oracle_tools.mean(g["e1t"], g["e2t"], tm, one, rec["t"][t], e3=g["e3t_0"], form=1)[:, 0, :]
        want = np.where(s[:, 1] != 0, s[:, 2], 99999.0)
        assert np.allclose(var["sum_votemper"][t].ravel(), want, rtol=2e-7, atol=0)
        cum += s[:, 2].sum()
        assert np.allclose(var["sum_3D_tvotemper"][t].ravel(), s[:, 2].sum(), rtol=2e-7)
        assert np.allclose(var["sum_3Dvotemper"][t].ravel(), cum, rtol=2e-7)
    assert att["sum_votemper"]["units"].endswith(b".m2") and att["sum_3Dvotemper"]["units"].endswith(b".m3")
    assert "Sum value over the ocean: (cumulated in time)" in out and " mean Sum over time " in out

    # a 3-D window, the U mask given with -M on the V point, unweighted
    rc, out, err = run("cdfsum", ["-f", "SYN_y0001_gridV.nc", "-v", "vomecrty", "-p", "V", "-w", "3", "20", "2", "15", "2", "4",
                                  "-M", "mask.nc", "umask", "-nowght", "-o", "w.nc"], d)
    assert rc == 0, out + err
    var, att, dims = read_nc(os.path.join(d, "w.nc"))
    assert dims["depthv"] == 3 and not att["sum_vomecrty"]["units"].endswith(b".m2")
    sl = (slice(1, 4), slice(1, 15), slice(2, 20))
    ones2 = np.ones((14, 18), np.float32)
    s = oracle_tools.mean(ones2, ones2, g["umask"].astype(np.float32)[sl], one[1:15, 2:20], rec["v"][1][sl],
                          e3=np.ones((3, 14, 18), np.float32), form=1)[:, 0, :]
    assert np.allclose(var["sum_vomecrty"][1].ravel(), np.where(s[:, 1] != 0, s[:, 2], 99999.0), rtol=2e-7, atol=1e-9)
    # -nomsk
    rc, out, err = run("cdfsum", ["-f", "SYN_y0001_gridT.nc", "-v", "vosaline", "-p", "T", "-nomsk", "-o", "n.nc"], d)
    assert rc == 0, out + err
    var, _, _ = read_nc(os.path.join(d, "n.nc"))
    s = oracle_tools.mean(g["e1t"], g["e2t"], np.ones_like(tm), one, rec["s"][0], e3=g["e3t_0"], form=1)[:, 0, :]
    assert np.allclose(var["sum_vosaline"][0].ravel(), s[:, 2], rtol=2e-7, atol=0)

    # a file without depth (a forcing field): the other expression, and cdfsum.txt
    flx = (100 * np.random.default_rng(3).random((2, npj, npi), dtype=np.float32)).astype(np.float32)
    with nemo_files._new(os.path.join(d, "flx.nc"), [("time_counter", None), ("y", npj), ("x", npi)]) as f:
        nemo_files._var(f, "nav_lon", "f4", ("y", "x"), g["glamf"])
        nemo_files._var(f, "nav_lat", "f4", ("y", "x"), g["gphif"])
        nemo_files._var(f, "time_counter", "f8", ("time_counter",), np.array([1.0, 2.0]), units="seconds since 1950-01-01 00:00:00",
                        calendar="noleap", time_origin="1950-JAN-01 00:00:00")
        nemo_files._var(f, "sohefldo", "f4", ("time_counter", "y", "x"), flx, units="W/m2")
    rc, out, err = run("cdfsum", ["-f", "flx.nc", "-v", "sohefldo", "-p", "T", "-o", "f.nc"], d)
    assert rc == 0, out + err
    assert "W A R N I N G : you used a forcing field" in out
    var, _, _ = read_nc(os.path.join(d, "f.nc"))
    for t in range(2):
        s = oracle_tools.mean(g["e1t"], g["e2t"], tm[:1], one, flx[t][None], e3=g["e3t_0"][:1], form=2)[:, 0, :]
        assert np.allclose(var["sum_sohefldo"][t].ravel(), s[0, 2], rtol=2e-7)
    lines = open(os.path.join(d, "cdfsum.txt")).read().splitlines()
    assert len(lines) == 2 and lines[0].split()[0] == "1"


# ------------------------------------------------------------------------------ cdfzonalsum / cdfzonalmean
def zonal_inputs(npi, npj, npk, nb, seed):
    g = synth.make_grid(npi, npj, npk, seed=seed)
    rng = np.random.default_rng(seed + 5)
    zv = ((-2 + 30 * rng.random((npk, npj, npi), dtype=np.float32)) * g["tmask"]).astype(np.float32)
    zm = np.zeros((npj, npi, nb), np.float32)
    zm[:, :, 0] = g["tmask"][0]
    if nb >= 5:
        zm[:, :, 1], zm[:, :, 3], zm[:, :, 4] = g["tmaskatl"], g["tmaskind"], g["tmaskpac"]
        zm[:, :, 2] = np.minimum(zm[:, :, 4] + zm[:, :, 3], 1)
    return g, zv, zm


def test_oracle_zonal_matches_numpy_restatement():
    g, zv, zm = zonal_inputs(19, 13, 5, 5, seed=8)
    tm = g["tmask"].astype(np.float32)
    surf = g["e1t"].astype(np.float64) * g["e2t"].astype(np.float64)
    s = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, zv)
    m = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, zv, mean_form=True)
    for b in range(5):
        d4 = ((zm[None, :, :, b] * tm).astype(np.float32) * zv).astype(np.float32).astype(np.float64)
        assert np.allclose(s["sum"][b], (surf[None] * d4).sum(axis=2), rtol=1e-13, atol=0)
        d8 = zm[None, :, :, b].astype(np.float64) * tm * zv
        assert np.allclose(m["sum"][b], (surf[None] * d8).sum(axis=2), rtol=1e-13, atol=0)
        assert np.allclose(m["area"][b], (surf[None] * zm[None, :, :, b] * tm).sum(axis=2), rtol=1e-13, atol=0)
        assert np.array_equal(m["max"][b], np.maximum(d8.max(axis=2), np.float32(-1e20)).astype(np.float32))
        mn = np.where(d8 != 0, d8, np.inf).min(axis=2)
        assert np.array_equal(m["min"][b], np.minimum(mn, np.float32(1e20)).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["class", "fast", "general"])
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 5), (182, 149, 31, 5), (1, 1, 1, 1), (300, 5, 2, 1), (1031, 4, 3, 5),
                                            (1537, 3, 2, 5), (4322, 2, 3, 5)])
def test_hip_zonal_matches_oracle(gpu, monkeypatch, npi, npj, npk, nb, kernel):
    """The three kernels behind cdft_zonal_compute: LDS class accumulators (0/1 masks, up to 8 basin patterns), the
    select kernel (0/1 masks) and the general one (any mask values); the environment hooks force the latter two."""
    if kernel != "class":
        monkeypatch.setenv("CDFT_ZONAL_GENERAL" if kernel == "general" else "CDFT_ZONAL_NOCLASS", "1")
    g, zv, zm = zonal_inputs(npi, npj, npk, nb, seed=npi + 9)
    tm = g["tmask"].astype(np.float32)
    plan = gpu.ZonalPlan(npi, npj, npk, nb)
    plan.set_mesh(g["e1t"], g["e2t"], tm, zm)
    for mean_form, minmax in ((False, False), (True, True), (True, False)):      # (-max: the class kernel too since round 5)
        got = plan.compute(zv, mean_form=mean_form, minmax=minmax)
        ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, zv, mean_form=mean_form)
        bound = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, np.abs(zv), mean_form=mean_form)["sum"]
        assert np.all(np.abs(got["sum"] - ref["sum"]) <= 1e-13 * bound + 1e-300)
        assert np.array_equal(plan.compute(zv, mean_form=mean_form, minmax=minmax)["sum"], got["sum"])   # run-to-run identical
        if kernel == "class":                     # the level masks as bytes (0/1 masks) against the r4 masks: same bits
            monkeypatch.setenv("CDFT_ZONAL_R4MASK", "1")
            r4 = plan.compute(zv, mean_form=mean_form, minmax=minmax)
            monkeypatch.delenv("CDFT_ZONAL_R4MASK")
            assert all(np.array_equal(r4[k], got[k]) for k in got)
        dev = plan.compute(zv, mean_form=mean_form, minmax=minmax, device_out=True)                       # results left in HBM
        assert sorted(dev) == sorted(got) and all(np.array_equal(dev[k].cpu().numpy(), got[k]) for k in got)
        if mean_form:
            assert np.all(np.abs(got["area"] - ref["area"]) <= 1e-13 * ref["area"] + 1e-300)
        if minmax:
            assert np.array_equal(got["max"], ref["max"]) and np.array_equal(got["min"], ref["min"])
    two = plan.compute(zv[:2] if npk > 1 else zv)                          # a variable with fewer levels
    ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm[:2], zm, zv[:2] if npk > 1 else zv)
    assert np.allclose(two["sum"], ref["sum"], rtol=1e-12, atol=1e-6)
    tm2 = (tm * np.float32(0.5)).astype(np.float32)                        # a level mask that is not 0/1 is never packed to bytes
    plan.set_mesh(g["e1t"], g["e2t"], tm2, zm)
    for mean_form in (False, True):
        got = plan.compute(zv, mean_form=mean_form)
        ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm2, zm, zv, mean_form=mean_form)
        bound = oracle_tools.zonal(g["e1t"], g["e2t"], tm2, zm, np.abs(zv), mean_form=mean_form)["sum"]
        assert np.all(np.abs(got["sum"] - ref["sum"]) <= 1e-13 * bound + 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 5), (1442, 9, 5, 5), (1031, 4, 3, 1), (4322, 3, 4, 5), (130, 70, 9, 8)])
def test_hip_zonal_max_on_the_class_kernel_equals_the_select_kernel(gpu, monkeypatch, npi, npj, npk, nb):
    """cdfzonalmean -max (src/cdfzonalmean.f90:316-326) on the class kernel (round 5: r4 extrema per lane and class in LDS,
    ds_max_f32 / ds_min_f32) against the select kernel (CDFT_ZONAL_MAX_SELECT) and the oracle: the same bits -- rows in one
    basin only, rows of land, values that round to the same r4, tiny values (the minimum skips an exact zero, not a denormal),
    negative zeros, and rows with a NaN or an Inf, which both hand to the general kernel"""
    g, zv, zm = zonal_inputs(npi, npj, npk, min(nb, 5), seed=npi + 31)
    if nb > 5:
        more = np.stack([(g["tmask"][0] * ((np.arange(npi) + b) % 3 == 0)[None, :]).astype(np.float32) for b in range(nb - 5)], axis=2)
        zm = np.concatenate([zm, more], axis=2)
    tm = g["tmask"].astype(np.float32)
    rng = np.random.default_rng(npi)
    zv = zv.copy()
    zv[0, 0, :] = 0.0                                                     # a row of zeros: max 0 (or -1e20 on land), min 1e20
    zv[-1, -1, ::3] = np.float32(1e-42)                                   # denormal r4 values
    zv[0, 1, ::2] = -0.0
    if npj > 2:
        zv[:, 2, :] = np.abs(zv[:, 2, :]) + 1                             # a strictly positive row: the 0 of a cell outside a basin is its minimum's rival, never taken
    plan = gpu.ZonalPlan(npi, npj, npk, nb)
    plan.set_mesh(g["e1t"], g["e2t"], tm, zm)
    for poison in (None, np.nan, np.inf):
        z = zv.copy()
        if poison is not None:
            z[npk // 2, npj // 2, int(rng.integers(npi))] = poison
            z[0, 0, 0] = poison
        got = plan.compute(z, mean_form=True, minmax=True)
        monkeypatch.setenv("CDFT_ZONAL_MAX_SELECT", "1")
        sel = plan.compute(z, mean_form=True, minmax=True)
        monkeypatch.delenv("CDFT_ZONAL_MAX_SELECT")
        for k in ("max", "min"):
            assert np.array_equal(got[k].view(np.uint32), sel[k].view(np.uint32)), (k, poison)
        assert np.array_equal(np.isnan(got["sum"]), np.isnan(sel["sum"]))
        if poison is None:
            ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, z, mean_form=True)
            assert np.array_equal(got["max"], ref["max"]) and np.array_equal(got["min"], ref["min"])
        assert all(np.array_equal(plan.compute(z, mean_form=True, minmax=True)[k], got[k], equal_nan=True) for k in got)


@pytest.mark.parametrize("tool", ["cdfzonalsum", "cdfzonalmean", "cdfmean"])
def test_usage_of_the_reduction_tools(tools, tmp_path, tool):  # noqa: F811
    rc, out, _ = run(tool, [], tmp_path)
    assert rc == 0 and "usage :" in out.splitlines()[0] and tool in out.splitlines()[0]
    rc, out, _ = run(tool, ["-thisisnotanoption"], tmp_path)
    assert rc == 99 and "unknown option" in out
    rc, out, _ = run(tool, ["-f", "nothere.nc", "-p", "T"] + (["-v", "votemper"] if tool == "cdfmean" else []), tmp_path)
    assert rc == 99 and "missing" in out


@pytest.mark.gpu
def test_cdfzonalsum_and_cdfzonalmean_end_to_end(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    npk, npj, npi = g["tmask"].shape
    tm = g["tmask"].astype(np.float32)
    zm = np.zeros((npj, npi, 5), np.float32)
    zm[:, :, 0] = g["tmask"][0]
    zm[:, :, 1], zm[:, :, 3], zm[:, :, 4] = g["tmaskatl"], g["tmaskind"], g["tmaskpac"]
    zm[:, :, 2] = np.minimum(zm[:, :, 4] + zm[:, :, 3], 1)
    sfx = ["_glo", "_atl", "_inp", "_ind", "_pac"]

    rc, out, err = run("cdfzonalsum", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-b", "new_maskglo.nc"], d)
    assert rc == 0, out + err
    var, att, dims = read_nc(os.path.join(d, "zonalsum.nc"))
    assert dims["x"] == 1 and dims["y"] == npj and att["zoitemper_atl"]["long_name"].startswith(b"Zonal_Integral_")
    for nm, key in (("zoitemper", "t"), ("zoisaline", "s")):
        for t in range(2):
            ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, rec[key][t])["sum"]
            for b in range(5):
                got = var[nm + sfx[b]][t, :, :, 0]
                assert np.allclose(got, ref[b], rtol=2e-7, atol=0), (nm, b)
    assert np.array_equal(var["deptht"], -g["gdept_1d"])
    # one variable, per degree of latitude, positive depths
    rc, out, err = run("cdfzonalsum", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-l", "vosaline", "-pdeg", "-pdep"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "zonalintdeg.nc"))
    assert [n for n in var if n.startswith("zoi")] == ["zoisaline_glo"] and np.array_equal(var["deptht"], g["gdept_1d"])
    alpha = (g["e2t"][:, 0] * np.float32(360.) / (np.float32(2.0) * np.arccos(np.float32(-1.))) / np.float32(6371229.)).astype(np.float32)
    ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm[:, :, :1], rec["s"][0])["sum"][0] / alpha[None, :]
    assert np.allclose(var["zoisaline_glo"][0, :, :, 0], ref, rtol=3e-7)

    rc, out, err = run("cdfzonalmean", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-b", "new_maskglo.nc", "-max", "-l", "votemper"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "zonalmean.nc"))
    assert att["zotemper_pac_max"]["long_name"].startswith(b"Zonal_Max_")
    for t in range(2):
        ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, rec["t"][t], mean_form=True)
        ok = ref["area"] != 0
        mean = np.where(ok, ref["sum"] / np.where(ok, ref["area"], 1), 0.0)
        for b in range(5):
            assert np.allclose(var["zotemper" + sfx[b]][t, :, :, 0], mean[b], rtol=2e-7, atol=0)
            assert np.array_equal(var["zotemper" + sfx[b] + "_max"][t, :, :, 0], np.where(ok[b], ref["max"][b], 0).astype(np.float32))
            assert np.array_equal(var["zotemper" + sfx[b] + "_min"][t, :, :, 0], np.where(ok[b], ref["min"][b], 0).astype(np.float32))


@pytest.mark.gpu
def test_cdfzonal_file_with_2d_and_3d_variables(tools, tmp_path):  # noqa: F811
    """A file that holds a 2-D variable beside a 3-D one: the results of a call are packed for the levels of THAT variable
    ((npjglo, 1, nbasins) for the 2-D one), and every basin of both must land in its own output variable."""
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    npk, npj, npi = g["tmask"].shape
    ssh = (np.random.default_rng(5).random((2, npj, npi), dtype=np.float32) - np.float32(0.5)) * g["tmask"][0]
    nemo_files.write_data(os.path.join(d, "mix.nc"), g, "deptht", {"sossheig": ssh.astype(np.float32), "votemper": rec["t"]}, fill=0.0)
    tm = g["tmask"].astype(np.float32)
    zm = np.zeros((npj, npi, 5), np.float32)
    zm[:, :, 0] = g["tmask"][0]
    zm[:, :, 1], zm[:, :, 3], zm[:, :, 4] = g["tmaskatl"], g["tmaskind"], g["tmaskpac"]
    zm[:, :, 2] = np.minimum(zm[:, :, 4] + zm[:, :, 3], 1)
    sfx = ["_glo", "_atl", "_inp", "_ind", "_pac"]
    for tool, out_name, pre in (("cdfzonalsum", "zonalsum.nc", "zoi"), ("cdfzonalmean", "zonalmean.nc", "zo")):
        rc, out, err = run(tool, ["-f", "mix.nc", "-p", "T", "-b", "new_maskglo.nc"], d)
        assert rc == 0, out + err
        var, _, _ = read_nc(os.path.join(d, out_name))
        for t in range(2):
            for nm, fld, msk in ((pre + "ssheig", ssh[t][None], tm[:1]), (pre + "temper", rec["t"][t], tm)):
                ref = oracle_tools.zonal(g["e1t"], g["e2t"], msk, zm, fld, mean_form=tool == "cdfzonalmean")
                want = ref["sum"]
                if tool == "cdfzonalmean":
                    ok = ref["area"] != 0
                    want = np.where(ok, ref["sum"] / np.where(ok, ref["area"], 1), 0.0)
                for b in range(5):
                    got = var[nm + sfx[b]][t]
                    got = got[:, :, 0] if got.ndim == 3 else got[None, :, 0]
                    assert np.allclose(got, want[b], rtol=2e-7, atol=1e-30), (tool, nm, b, t)


# ------------------------------------------------------------------------------------ cdfvint / cdfvertmean
def vint_schedule(e31d, gdepo, lstd):
    """The scalar bookkeeping of src/cdfvint.f90:266-311 in r4, as the Fortran host program does it."""
    npk = len(e31d)
    kout, hit = -np.ones(npk, np.int32), np.zeros(npk, np.int32)
    clip, rdep = np.zeros(npk, np.float32), np.zeros(npk, np.float32)
    iko, r1, r2, lout = 0, np.float32(0), np.float32(0), True
    for k in range(npk):
        if lstd:
            lout = False
        r1 = r2
        r2 = np.float32(r1 + np.float32(e31d[k]))
        rdep[k] = r2
        if iko < len(gdepo):
            if r2 >= np.float32(np.float32(gdepo[iko]) - np.float32(1.0)):
                lout, hit[k], clip[k] = True, 1, np.float32(np.float32(gdepo[iko]) - r1)
            if lout:
                kout[k] = iko
                iko += 1
    return kout, hit, clip, rdep


def column_inputs(npi, npj, npk, seed):
    g = synth.make_grid(npi, npj, npk, seed=seed)
    rng = np.random.default_rng(seed + 4)
    zt = ((-2 + 30 * rng.random((npk, npj, npi), dtype=np.float32)) * g["tmask"]).astype(np.float32)
    return g, zt, g["tmask"].astype(np.float32)


def test_oracle_vint_and_vertmean_match_numpy_restatement_bitwise():
    g, zt, tm = column_inputs(17, 11, 7, seed=3)
    e31d = g["e3t_1d"]
    gdepo = np.append(g["gdepw_1d"][1:], np.float32(6000.0)).astype(np.float32)
    sclf = np.float32(np.float32(1020.) * np.float32(4000.) / np.float32(1.e6))
    got = oracle_tools.vint_cumul(zt, tm, e31d, gdepo, sclf=sclf, e3t=g["e3t_0"])
    kout, hit, clip, _ = vint_schedule(e31d, gdepo, False)
    v1 = np.zeros(zt.shape[1:])
    want = np.zeros_like(got)
    for k in range(7):
        v2 = v1.copy()
        v1 = v1 + ((zt[k] * g["e3t_0"][k]).astype(np.float32) * tm[k]).astype(np.float64)
        if hit[k]:
            e = np.minimum(g["e3t_0"][k], clip[k]).astype(np.float32)
            v2 = v2 + ((zt[k] * e).astype(np.float32) * tm[k]).astype(np.float64)
        if kout[k] >= 0:
            want[kout[k]] = (v2 * np.float64(sclf)).astype(np.float32)
    assert np.array_equal(got, want) and np.abs(got).max() > 0
    # vertical mean between 15 m and 120 m
    gdep = g["gdepw_1d"]
    ik1 = int(np.nonzero(gdep <= 15.0)[0].max())
    ik2 = int(np.nonzero(gdep <= 120.0)[0].max())
    sl = slice(ik1, ik2 + 1)
    m = oracle_tools.vertmean(zt[sl], tm[sl], gdep[ik1], gdep[ik2], 15.0, 120.0, e3=g["e3t_0"][sl])
    dvol, ds = np.zeros(zt.shape[1:]), np.zeros(zt.shape[1:])
    for n, k in enumerate(range(ik1, ik2 + 1)):
        e = g["e3t_0"][k].copy()
        if n == 0:
            e = np.minimum(e, (np.float32(gdep[ik1]) + e).astype(np.float32) - np.float32(15.0)).astype(np.float32)
        if k == ik2:
            e = np.minimum(e, np.float32(np.float32(120.0) - np.float32(gdep[ik2]))).astype(np.float32)
        dvol = (e * tm[k]).astype(np.float64) + dvol
        ds = ((zt[k] * e).astype(np.float32) * tm[k]).astype(np.float64) + ds
    assert np.array_equal(m, np.where(dvol != 0, ds / np.where(dvol != 0, dvol, 1), ds))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["vec4", "scalar"])
@pytest.mark.parametrize("npi,npj,npk", [(24, 18, 7), (182, 149, 31), (1, 1, 2), (130, 5, 4), (3, 1, 3), (5, 1, 9), (7, 3, 5),
                                         (1027, 3, 6)])
def test_hip_vint_and_vertmean_equal_oracle_bitwise(gpu, monkeypatch, kernel, npi, npj, npk):
    # vec4: 4 cells per thread with the last vector anchored at nij-4 (nij % 4 of every kind, nij < 4 falls back);
    # scalar: one cell per thread
    monkeypatch.setenv("CDFT_COLUMN_SCALAR", "1" if kernel == "scalar" else "0")
    g, zt, tm = column_inputs(npi, npj, npk, seed=npi + 6)
    e31d = g["e3t_1d"]
    sclf = np.float32(np.float32(1020.) * np.float32(4000.) / np.float32(1.e6))
    for gdepo, lstd in ((np.append(g["gdepw_1d"][1:], np.float32(6000.0)).astype(np.float32), False),
                        (np.array([100., 300., 500., 700., 800., 2000., 6000.], np.float32), True),
                        (np.array([700., 2000., 6000.], np.float32), True)):
        kout, hit, clip, rdep = vint_schedule(e31d, gdepo, lstd)
        for mean in (False, True):
            for kw, okw in ((dict(e3t=g["e3t_0"]), dict(e3t=g["e3t_0"])), (dict(e31d=e31d), dict(full=True))):
                got = gpu.vint_cumul(zt, tm, kout, hit, clip, rdep, len(gdepo), sclf=sclf, mean=mean, **kw)
                ref = oracle_tools.vint_cumul(zt, tm, e31d, gdepo, lstd=lstd, sclf=sclf, mean=mean, **okw)
                assert np.array_equal(got, ref), (lstd, mean, list(kw))
    gdep = g["gdepw_1d"]
    for up, down in ((0.0, 50.0), (15.0, 120.0), (30.0, 30.0), (0.0, 1.0e4)):
        ik1 = int(np.nonzero(gdep <= up)[0].max())
        ik2 = int(np.nonzero(gdep <= down)[0].max())
        sl = slice(ik1, ik2 + 1)
        for kw in (dict(e3=g["e3t_0"][sl]), dict(e31d=e31d[sl])):
            got = gpu.vertmean(zt[sl], tm[sl], gdep[ik1], gdep[ik2], up, down, **kw)
            assert np.array_equal(got, oracle_tools.vertmean(zt[sl], tm[sl], gdep[ik1], gdep[ik2], up, down, **kw)), (up, down)


@pytest.mark.gpu
def test_cdfvint_and_cdfvertmean_end_to_end(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    tm = g["tmask"].astype(np.float32)
    e31d = g["e3t_1d"]
    sclf = np.float32(np.float32(1020.) * np.float32(4000.) / np.float32(1.e6))
    rc, out, err = run("cdfvint", ["-f", "SYN_y0001_gridT.nc"], d)
    assert rc == 0 and "using model depths" in out, out + err
    var, att, dims = read_nc(os.path.join(d, "voheatc.nc"))
    gdepo = np.append(g["gdepw_1d"][1:], np.float32(6000.0)).astype(np.float32)
    for t in range(2):
        assert np.array_equal(var["voheatc"][t], oracle_tools.vint_cumul(rec["t"][t], tm, e31d, gdepo, sclf=sclf, e3t=g["e3t_0"]))
    assert att["voheatc"]["units"] == b"10^6 J/m2" and np.array_equal(var["depthw"], gdepo)
    rc, out, err = run("cdfvint", ["-f", "SYN_y0001_gridT.nc", "-v", "vosaline", "-OCCI", "-smean", "-full", "-w", "3", "20", "2", "15",
                                   "-o", "sm.nc"], d)
    assert rc == 0, out + err
    var, att, dims = read_nc(os.path.join(d, "sm.nc"))
    occi = np.array([700., 2000., 6000.], np.float32)
    ref = oracle_tools.vint_cumul(rec["s"][1][:, 1:15, 2:20], tm[:, 1:15, 2:20], e31d, occi, lstd=True, mean=True, full=True)
    assert dims["depthw"] == 3 and np.array_equal(var["vosaline"][1], ref)
    rc, out, _ = run("cdfvint", ["-f", "SYN_y0001_gridT.nc", "-v", "sossheig"], d)
    assert rc == 99 and "not pre-defined" in out

    rc, out, err = run("cdfvertmean", ["-f", "SYN_y0001_gridT.nc", "-l", "votemper,vosaline", "-p", "T", "-zlim", "15", "120"], d)
    assert rc == 0, out + err
    var, att, _ = read_nc(os.path.join(d, "vertmean.nc"))
    gdep = g["gdepw_1d"]
    ik1 = int(np.nonzero(gdep <= 15.0)[0].max())
    ik2 = int(np.nonzero(gdep <= 120.0)[0].max())
    sl = slice(ik1, ik2 + 1)
    for nm, key in (("votemper", "t"), ("vosaline", "s")):
        for t in range(2):
            ref = oracle_tools.vertmean(rec[key][t][sl], tm[sl], gdep[ik1], gdep[ik2], 15.0, 120.0, e3=g["e3t_0"][sl])
            assert np.array_equal(var[nm + "_vert_mean"][t], ref.astype(np.float32)), nm
    rc, out, _ = run("cdfvertmean", ["-f", "SYN_y0001_gridT.nc", "-l", "votemper", "-p", "T", "-zlim", "100", "10"], d)
    assert rc == 99 and "increasing order" in out


# ------------------------------------------------------------------------------------------- cdfmocsig
import oracle_lib  # noqa: E402


def mocsig_inputs(npi, npj, npk, nb, seed):
    g = synth.make_grid(npi, npj, npk, seed=seed, nbasins=nb if nb > 1 else 5)
    e1v, e3v, ib = synth.moc_inputs(g, nbasins=nb)
    rng = np.random.default_rng(seed + 2)
    zv = ((rng.random((npk, npj, npi), dtype=np.float32) - np.float32(0.5)) * g["vmask"]).astype(np.float32)
    zt = ((np.float32(-2) + np.float32(30) * rng.random((npk, npj, npi), dtype=np.float32)) * g["tmask"]).astype(np.float32)
    zs = ((np.float32(33) + np.float32(4) * rng.random((npk, npj, npi), dtype=np.float32)) * g["tmask"]).astype(np.float32)
    return g, e1v, e3v, ib, zv, zt, zs


def test_oracle_mocsig_matches_numpy_restatement():
    g, e1v, e3v, ib, zv, zt, zs = mocsig_inputs(21, 12, 5, 5, seed=4)
    nbins, sigmin, sigstp = 52, np.float32(23.0), np.float32(0.1)
    gdep = -g["gdept_1d"]
    got, depi = oracle_lib.mocsig(zv, zt, zs, e1v, ib, nbins, sigmin, sigstp, e3v=e3v, gdep=gdep, isodep=True)
    npk, npj, npi = zv.shape
    raw = np.zeros((npj, nbins, 5))
    dd, ww = np.zeros((npj, nbins, 5)), np.zeros((npj, nbins, 5))
    for k in range(npk - 1):
        itm = (zs[k] != 0).astype(np.float32)                                      # zsps = 0: S = 0 is land
        dens = oracle_lib.sigmai(zt[k], zs[k], 0.0)
        ztt = (dens * itm).astype(np.float32)
        ibin = np.clip(((ztt - sigmin) / sigstp).astype(np.float32).astype(np.int64), 1, nbins) - 1
        zarea = (e1v * e3v[k]).astype(np.float32)
        term = -(zv[k] * zarea).astype(np.float32).astype(np.float64)
        tdep = ((np.float32(gdep[k]) * itm).astype(np.float32) * zarea).astype(np.float32).astype(np.float64)
        twei = (itm * zarea).astype(np.float32).astype(np.float64)
        for j in range(1, npj - 1):
            for i in range(1, npi - 1):
                raw[j, ibin[j, i]] += term[j, i] * ib[j, i]
                dd[j, ibin[j, i]] += tdep[j, i] * ib[j, i]
                ww[j, ibin[j, i]] += twei[j, i] * ib[j, i]
    want = np.zeros_like(raw)
    want[:, nbins - 1] = raw[:, nbins - 1] / 1e6
    for n in range(nbins - 2, -1, -1):
        want[:, n] = want[:, n + 1] + raw[:, n] / 1e6
    assert np.allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max()) and np.abs(want).max() > 0
    wd = np.where(ww != 0, dd / np.where(ww != 0, ww, 1), 99999.0)
    assert np.allclose(depi, wd, rtol=1e-12, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb,pref,teos", [(24, 18, 7, 5, 0.0, False), (182, 149, 31, 5, 2000.0, False),
                                                      (40, 1, 4, 1, 1000.0, True), (700, 6, 5, 5, 0.0, True)])
def test_hip_mocsig_matches_oracle(gpu, npi, npj, npk, nb, pref, teos):
    g, e1v, e3v, ib, zv, zt, zs = mocsig_inputs(npi, npj, npk, nb, seed=npi + 1)
    nbins, sigmin, sigstp = {0.0: (52, 23.0, 0.1), 1000.0: (88, 24.0, 0.1), 2000.0: (158, 30.0, 0.05)}[pref]
    gdep = -g["gdept_1d"]
    eiv = (zv * np.float32(0.1)).astype(np.float32)
    for kw in (dict(e3v=e3v), dict(e31d=g["e3t_1d"], zveiv=eiv)):
        ref, rdep = oracle_lib.mocsig(zv, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=pref, teos10=teos, gdep=gdep, isodep=True, **kw)
        got, gdp = gpu.mocsig(zv, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=pref, eos=gpu.EOS_TEOS10 if teos else gpu.EOS_80,
                              gdep=gdep, isodep=True, **kw)
        assert np.all(np.abs(got - ref) <= 1e-10 * np.abs(ref).max() + 1e-300), float(np.abs(got - ref).max())
        assert np.allclose(gdp, rdep, rtol=1e-10, atol=0)
        only = gpu.mocsig(zv, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=pref, eos=gpu.EOS_TEOS10 if teos else gpu.EOS_80, **kw)
        assert np.all(np.abs(only - ref) <= 1e-10 * np.abs(ref).max() + 1e-300)
        again = gpu.mocsig(zv, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=pref, eos=gpu.EOS_TEOS10 if teos else gpu.EOS_80, **kw)
        assert np.array_equal(only, again)                                   # per-wave histograms: run-to-run identical


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["class", "general"])
@pytest.mark.parametrize("npi,npj,npk,nb,smooth", [(24, 18, 7, 5, False), (182, 149, 9, 5, True), (1031, 5, 4, 5, True),
                                                   (2045, 4, 3, 8, False), (4322, 3, 3, 5, True), (5, 4, 3, 5, False)])
def test_hip_mocsig_kernels(gpu, monkeypatch, npi, npj, npk, nb, smooth, kernel):
    """The class kernel (0/1 masks, <= 8 bit patterns: vector loads, one LDS accumulator per pattern and bin, the cells
    of a vector merged in registers) and the general kernel (CDFT_MOCSIG_NOCLASS=1) against the oracle, with white-noise
    or smooth T,S (neighbours in one bin), host arrays and arrays resident in HBM, a NaN on a cell outside a basin."""
    import torch

    if kernel == "general":
        monkeypatch.setenv("CDFT_MOCSIG_NOCLASS", "1")
    g, e1v, e3v, ib, zv, zt, zs = mocsig_inputs(npi, npj, npk, nb, seed=npi + 3)
    if smooth:
        x = np.linspace(0, 1, npi, dtype=np.float32)[None, None, :]
        z = np.linspace(0, 1, npk, dtype=np.float32)[:, None, None]
        zt = ((np.float32(25) * (1 - z) ** 2 + np.float32(2) * x) * g["tmask"]).astype(np.float32)
        zs = ((np.float32(34) + z + np.float32(0.5) * x) * g["tmask"]).astype(np.float32)
    nbins, sigmin, sigstp = 158, 30.0, 0.05
    for poison in (False, True):
        v = zv.copy()
        if poison:
            out = np.argwhere(ib[1:-1, 1:-1, min(1, nb - 1)] == 0)
            if len(out) == 0:
                continue
            j, i = out[len(out) // 2] + 1
            v[0, j, i] = np.nan
        ref = oracle_lib.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v)
        bound = np.abs(oracle_lib.mocsig(np.abs(np.nan_to_num(v)), zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v))
        got = gpu.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v)
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and (not poison or np.isnan(ref).any())
        ok = ~np.isnan(ref)
        lim = 1e-13 * bound.max(axis=1, keepdims=True) + 1e-300          # the cumulated sums of a row share its largest
        assert np.all(np.abs(got - ref)[ok] <= np.broadcast_to(lim, ref.shape)[ok]), float(np.nanmax(np.abs(got - ref)))
        dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (v, zt, zs, e1v, ib, e3v)]
        res = gpu.mocsig(dev[0], dev[1], dev[2], dev[3], dev[4], nbins, sigmin, sigstp, pref=2000.0, e3v=dev[5])
        assert np.array_equal(res.cpu().numpy(), got, equal_nan=True)       # same kernel, same order: identical
    # -eiv: the eddy-induced velocity added in r4 after the missing-value select (a fifth stream of the class kernel);
    # a missing value in v still takes the eiv term (cdfmocsig.f90:374-378)
    eiv = (zv[::-1] * np.float32(0.3)).astype(np.float32)
    v = zv.copy()
    v[0, 1, 1:3] = np.float32(-9999.0)
    ref = oracle_lib.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v, zveiv=eiv, spv=(-9999.0, 0.0, 0.0))
    bound = np.abs(oracle_lib.mocsig(np.abs(v) + np.abs(eiv), zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v))
    got = gpu.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v, zveiv=eiv, spv=(-9999.0, 0.0, 0.0))
    assert np.all(np.abs(got - ref) <= 1e-13 * bound.max(axis=1, keepdims=True) + 1e-300), float(np.abs(got - ref).max())
    assert npi == 5 or not np.array_equal(got, gpu.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v, spv=(-9999.0, 0.0, 0.0)))
    # -full with -eiv: the level's e3t_1d in place of e3v (cdfmocsig.f90:389)
    ref = oracle_lib.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e31d=g["e3t_1d"], zveiv=eiv, spv=(-9999.0, 0.0, 0.0))
    bound = np.abs(oracle_lib.mocsig(np.abs(v) + np.abs(eiv), zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e31d=g["e3t_1d"]))
    got = gpu.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e31d=g["e3t_1d"], zveiv=eiv, spv=(-9999.0, 0.0, 0.0))
    assert np.all(np.abs(got - ref) <= 1e-13 * bound.max(axis=1, keepdims=True) + 1e-300), float(np.abs(got - ref).max())
    # -isodep: depth * weight and weight sums in two more histograms of the class kernel; a cell with v = 0 still counts
    gdep = -g["gdept_1d"]
    v[0, 2, 1:] = 0
    ref, rdep = oracle_lib.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v, zveiv=eiv, spv=(-9999.0, 0.0, 0.0),
                                  gdep=gdep, isodep=True)
    bound = np.abs(oracle_lib.mocsig(np.abs(v) + np.abs(eiv), zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v))
    got, gdp = gpu.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v, zveiv=eiv, spv=(-9999.0, 0.0, 0.0),
                          gdep=gdep, isodep=True)
    assert np.all(np.abs(got - ref) <= 1e-13 * bound.max(axis=1, keepdims=True) + 1e-300), float(np.abs(got - ref).max())
    assert np.array_equal(gdp == 99999.0, rdep == 99999.0) and np.allclose(gdp, rdep, rtol=1e-12, atol=0)
    assert npi == 5 or (rdep != 99999.0).any()
    got2, gdp2 = gpu.mocsig(v, zt, zs, e1v, ib, nbins, sigmin, sigstp, pref=2000.0, e3v=e3v, zveiv=eiv, spv=(-9999.0, 0.0, 0.0),
                            gdep=gdep, isodep=True)
    assert np.array_equal(got, got2) and np.array_equal(gdp, gdp2)            # per-wave histograms: run-to-run identical
    # -ntr: the neutral density in the same classes
    ref = oracle_lib.mocsig(zv, zt, zs, e1v, ib, 80, 18.0, 0.2, e3v=e3v, ntr=True)
    bound = np.abs(oracle_lib.mocsig(np.abs(zv), zt, zs, e1v, ib, 80, 18.0, 0.2, e3v=e3v, ntr=True))
    got = gpu.mocsig(zv, zt, zs, e1v, ib, 80, 18.0, 0.2, eos=gpu.EOS_NTR, e3v=e3v)
    assert np.all(np.abs(got - ref) <= 1e-13 * bound.max(axis=1, keepdims=True) + 1e-300)


@pytest.mark.gpu
def test_cdfmocsig_end_to_end(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    e1v, e3v, ib = synth.moc_inputs(g, nbasins=5)
    rc, out, err = run("cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-r", "2000", "-isodep"], d)
    assert rc == 0 and "-nbins 158" in out, out + err
    var, att, dims = read_nc(os.path.join(d, "mocsig.nc"))
    assert dims["sigma"] == 158 and att["zoisoatl"]["units"] == b"m"
    assert np.allclose(var["sigma"], np.float32(30.0) + (np.arange(1, 159) - 0.5).astype(np.float32) * np.float32(0.05), rtol=1e-6)
    gdep = -g["gdept_1d"]
    names = ["glo", "atl", "inp", "ind", "pac"]
    for t in range(2):
        ref, rdep = oracle_lib.mocsig(rec["v"][t], rec["t"][t], rec["s"][t], e1v, ib, 158, 30.0, 0.05, pref=2000.0, e3v=g["e3v_0"],
                                      gdep=gdep, isodep=True, spv=(0.0, 0.0, 0.0))
        for b, nm in enumerate(names):
            got = var["zomsf" + nm][t, :, :, 0]                               # (sigma, y)
            assert np.allclose(got, ref[:, :, b].T, rtol=0, atol=2e-7 * np.abs(ref).max()), nm
            assert np.allclose(var["zoiso" + nm][t, :, :, 0], rdep[:, :, b].T, rtol=3e-7, atol=0), nm
    assert np.abs(var["zomsfglo"]).max() > 0
    rc, out, _ = run("cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-r", "500"], d)
    assert rc == 99 and "not implemented as standard" in out
    # -eiv -full, by either route to the GPU: the eddy-induced velocity of the V file added, e3t_1d for the thickness
    eiv = (rec["v"][::-1] * np.float32(0.3)).astype(np.float32)
    nemo_files.write_data(os.path.join(d, "EIV_y0001_gridV.nc"), g, "depthv", {"vomecrty": rec["v"], "vomeeivv": eiv})
    for feed in ("0", "1"):
        rc, out, err = run("cdfmocsig", ["-v", "EIV_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-r", "1000", "-eiv", "-full", "-o", "eiv.nc"], d,
                           env={"CDFT_FEED": feed})
        assert rc == 0 and "-nbins  88" in out.replace("-nbins 88", "-nbins  88"), out + err
        var, att, dims = read_nc(os.path.join(d, "eiv.nc"))
        for t in range(2):
            ref = oracle_lib.mocsig(rec["v"][t], rec["t"][t], rec["s"][t], e1v, ib, 88, 24.0, 0.1, pref=1000.0, e31d=g["e3t_1d"], zveiv=eiv[t],
                                    spv=(0.0, 0.0, 0.0))
            for b, nm in enumerate(names):
                assert np.allclose(var["zomsf" + nm][t, :, :, 0], ref[:, :, b].T, rtol=0, atol=2e-7 * np.abs(ref).max()), (feed, nm)
        assert np.abs(var["zomsfglo"]).max() > 0
    # -ntr: the neutral density binned in classes the user gives (the reference's default ones start at 1023 while
    # sigmantr returns rho - 1000: kept, everything then falls in the first class)
    rc, out, err = run("cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-ntr", "-sigmin", "19", "-sigstp",
                                     "0.25", "-nbins", "60", "-o", "ntr.nc"], d)
    assert rc == 0 and "For Neutral density MOC" in out, out + err
    var, _, dims = read_nc(os.path.join(d, "ntr.nc"))
    assert dims["sigma"] == 60
    for t in range(2):
        ref = oracle_lib.mocsig(rec["v"][t], rec["t"][t], rec["s"][t], e1v, ib, 60, 19.0, 0.25, e3v=g["e3v_0"], ntr=True)
        for b, nm in enumerate(names):
            assert np.allclose(var["zomsf" + nm][t, :, :, 0], ref[:, :, b].T, rtol=0, atol=2e-7 * np.abs(ref).max()), nm
        assert len(np.unique(np.round(ref[:, :, 0], 6))) > 20                  # spread over many classes
    rc, out, err = run("cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-ntr", "-o", "ntr0.nc"], d)
    assert rc == 0 and "-nbins  52" in out, out + err                        # the default classes of -ntr (:267-270)
    rc, out, _ = run("cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc"], d)
    assert rc == 99 and "mandatory arguments missing" in out


@pytest.mark.gpu
def test_hip_mocsig_land_shortcut_is_bit_identical(gpu, monkeypatch):
    """cdfmocsig's class kernel skips the equation of state for a wave whose 256 cells all have an exactly zero transport
    (land in large pieces): the histogram is the one of the run that evaluates every cell (CDFT_MOCSIG_NOLAND=1), bit for bit
    -- with a NaN velocity inside the land (not a zero: it must be binned), and with land pieces that end inside a wave."""
    npi, npj, npk, nb = 1031, 6, 5, 5
    g, e1v, _, ib, zv, zt, zs = mocsig_inputs(npi, npj, npk, nb, seed=31)
    land = np.zeros((npj, npi), bool)
    land[:, 100:700] = True
    land[2, :] = True
    land[4, 900:] = True
    zv = zv.copy()
    zv[:, land] = 0.0
    zv[1, 3, 350] = np.nan
    zt[:, land] = 0.0
    zs[:, land] = 0.0
    kw = dict(pref=2000.0, e3v=g["e3v_0"])
    monkeypatch.delenv("CDFT_MOCSIG_NOLAND", raising=False)
    got = gpu.mocsig(zv, zt, zs, e1v, ib, 158, np.float32(30.0), np.float32(0.05), **kw)
    monkeypatch.setenv("CDFT_MOCSIG_NOLAND", "1")
    full = gpu.mocsig(zv, zt, zs, e1v, ib, 158, np.float32(30.0), np.float32(0.05), **kw)
    assert got.tobytes() == full.tobytes() and np.isnan(got[3]).any() and np.isfinite(got[0]).all()
    ref = oracle_lib.mocsig(zv, zt, zs, e1v, ib, 158, np.float32(30.0), np.float32(0.05), **kw)
    ok = np.isfinite(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.all(np.abs(got[ok] - ref[ok]) <= 1e-10 * np.abs(ref[ok]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("teos,pref", [(False, 2000.0), (True, 0.0), (False, 0.0)])
def test_hip_mocsig_zero_missing_values_form_is_bit_identical(gpu, monkeypatch, teos, pref):
    """With zspv = zspt = zsps = 0 (NEMO files) the class kernel leaves out the compares against the missing values and the
    second mask product (mocsig.hip, OPT = 3): the same histogram, bit for bit, as the general form (CDFT_MOCSIG_SPVAL_GENERAL=1)
    -- with -0.0 velocities and salinities, zeros in T and S, a NaN, and land."""
    npi, npj, npk, nb = 1031, 6, 5, 5
    g, e1v, _, ib, zv, zt, zs = mocsig_inputs(npi, npj, npk, nb, seed=41)
    zv, zt, zs = zv.copy(), zt.copy(), zs.copy()
    zv[:, 1, 200:260] = -0.0
    zs[:, 2, 300:420] = -0.0
    zs[:, 3, 500:600] = 0.0
    zt[:, 3, 550:650] = 0.0
    zt[:, 4, 10:40] = -0.0
    zv[1, 4, 700] = np.nan
    zs[2, 0, 800] = np.nan
    kw = dict(pref=pref, e3v=g["e3v_0"], eos=1 if teos else 0)
    monkeypatch.delenv("CDFT_MOCSIG_SPVAL_GENERAL", raising=False)
    got = gpu.mocsig(zv, zt, zs, e1v, ib, 158, np.float32(30.0 if pref else 20.0), np.float32(0.05), **kw)
    monkeypatch.setenv("CDFT_MOCSIG_SPVAL_GENERAL", "1")
    gen = gpu.mocsig(zv, zt, zs, e1v, ib, 158, np.float32(30.0 if pref else 20.0), np.float32(0.05), **kw)
    assert got.tobytes() == gen.tobytes()
    assert np.isnan(got).any() and np.isfinite(got).any()
