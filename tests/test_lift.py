"""Pins the C oracle -- and through it the HIP path -- to the reference's own statements.

tests/golden/lift_*.npz hold OUTPUTS of harnesses built around LITERAL statements of the reference programs
(oracle/ref_lift.py: the declaration blocks, ALLOCATEs and whole time loops of cdfmoc / cdfheatc / cdfmhst / cdfpsi /
cdfvT / cdfcurl, lifted by line range from /root/reference/src where it lies, compiled with amdflang -O1
-ffp-contract=off; only I/O statements are rewritten into plain array accesses).  They are not outputs of the
reference PROGRAMS (no cdfio, no files, no option parser) but of the reference's arithmetic as its compiler sees it.

  CPU : oracle == golden BIT FOR BIT for every tool, option and output the harness captures (r8 work arrays and the
        r4 values handed to putvar); and, where oracle/_ref/liblift_ref.so exists (the build container), oracle ==
        harness on fresh random cases that are not in the goldens.
  GPU : HIP (through the C ABI) against the same goldens: bit-exact for cdfpsi, cdfvT, cdfcurl; |d| <= 1e-13 *
        sum|terms| for the regrouped zonal sums of cdfmoc / cdfmhst and written floats within 1 ulp(r4); cdfheatc
        against the reference's r4 sums inside the rigorous bound of a sequential r4 sum, n * 2^-24 * sum|terms|.
"""
import os

import numpy as np
import pytest

import lift_cases as lc
import oracle_lib
import oracle_tools as ot
import ref_lift_lib as rl

f32 = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(tool):
    return np.load(os.path.join(GOLD, f"lift_{tool}.npz"))


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def ulp4(x):
    return np.spacing(np.abs(x).astype(f32)).astype(np.float64)


# ================================================================================================ cdfmoc
def moc_oracle(name):
    g, zv, nb, full, dec, teos = lc.moc_case(name)
    e1v, e3v, ib = moc_mesh(g, nb, full)
    return g, zv, nb, full, dec, teos, e1v, e3v, ib


def moc_mesh(g, nb, full):
    import synth
    e1v, e3v, ib = synth.moc_inputs(g, nb)
    if full:                                          # get_e3v with -full: e31d(jk) * vmask (src/cdfmoc.f90:589-594)
        e3v = (g["e3t_1d"][:, None, None] * g["vmask"].astype(f32)).astype(f32)
    return e1v, e3v, ib


def moc_vars(dmoc, nb):
    """What the output section hands to putvar (src/cdfmoc.f90:520-551): REAL(dmoc(basin,:,jk)), then glo - atl."""
    out = [dmoc[..., b].astype(f32) for b in range(nb)]
    if nb == 5:
        out.append((dmoc[..., 0] - dmoc[..., 1]).astype(f32))
    return np.stack(out, axis=1)                      # (npt,nvar,npk,npj)


def decomp_vars(tot, dec, nb):
    """-decomp order (:521-569): per basin total, sh, bt, ag; then the four glo - atl differences."""
    out = []
    for b in range(nb):
        out += [tot[..., b].astype(f32)] + [dec[:, c, ..., b].astype(f32) for c in range(3)]
    if nb == 5:
        out += [(tot[..., 0] - tot[..., 1]).astype(f32)] + [(dec[:, c, ..., 0] - dec[:, c, ..., 1]).astype(f32) for c in range(3)]
    return np.stack(out, axis=1)


def oracle_decomp_records(g, zv, dec, teos, e1v, e3v, ib):
    sh, tot, parts = None, [], []
    for t in range(zv.shape[0]):
        o = oracle_lib.moc_decomp(e1v=e1v, e3v=e3v, ibmask=ib, zv=zv[t], ztemp=dec["t"][t], zsal=dec["s"][t],
                                  tmask=g["tmask"].astype(np.uint8), umask=g["umask"].astype(np.uint8), e1u=g["e1u"],
                                  zcoef=lc.moc_zcoef(g), gdept=g["gdept_1d"].astype(f32), teos10=teos, dmoc_sh=sh)
        sh = o["sh"]                                  # the reference never clears dmoc_sh (:471)
        tot.append(o["total"])
        parts.append(np.stack([o["sh"], o["bt"], o["ag"]]))
    return np.stack(tot), np.stack(parts)


def test_oracle_decomp_on_the_program_case_equals_reference_statements(tmp_path):
    """cdfmoc -decomp on the 90 x 61 x 15 grid of lc.moc_prog_case (two records): the reference's statements formed zcoef from
    gphiv themselves with the r4 SIN of amdflang's runtime (src/cdfmoc.f90:427-433); the oracle is handed lc.moc_zcoef's."""
    g, rec = lc.moc_prog_case(str(tmp_path))
    e1v, e3v, ib = moc_mesh(g, 5, False)
    G = gold("cdfmoc_decomp_prog")
    tot, parts = oracle_decomp_records(g, rec["v"], dict(t=rec["t"], s=rec["s"]), False, e1v, e3v, ib)
    assert same(tot, G["dmoc"]) and same(parts, G["dec"])
    assert same(decomp_vars(tot, parts, 5), G["var"])


@pytest.mark.parametrize("name", lc.MOC_CASES)
def test_oracle_moc_equals_reference_statements(name):
    g, zv, nb, full, dec, teos, e1v, e3v, ib = moc_oracle(name)
    G = gold("cdfmoc")
    if dec is None:
        ref = oracle_lib.moc(e1v, e3v, ib, zv)
        assert same(ref, G[f"{name}.dmoc"])
        assert same(moc_vars(ref, nb), G[f"{name}.var"])
    else:
        tot, parts = oracle_decomp_records(g, zv, dec, teos, e1v, e3v, ib)
        assert same(tot, G[f"{name}.dmoc"]) and same(parts, G[f"{name}.dec"])
        assert same(decomp_vars(tot, parts, nb), G[f"{name}.var"])


# ================================================================================================ cdfheatc
def heatc_oracle(name, r4sum):
    g, temp, full, mxlopt, mxl = lc.heatc_case(name)
    kw = dict(e31d=g["e3t_1d"]) if full else dict(e3t=g["e3t_0"])
    return np.stack([ot.heatc(g["e1t"], g["e2t"], g["tmask"], temp[t], mxl=None if mxl is None else mxl[t],
                              gdepw=g["gdepw_1d"], mxlopt=mxlopt, r4sum=r4sum, **kw) for t in range(temp.shape[0])])


def heatc_vars(sums):
    """src/cdfheatc.f90:227-247: dvol, dsum r8 running totals of the level values; heatc2d = pprho0*ppcp*dsum2d with the
    r4 constants multiplied first (left to right), heatc3d likewise, heatc3dpervol = heatc3d / dvol."""
    npt, npk, _ = sums.shape
    var = np.zeros((npt, 3, npk))
    c = np.float64(f32(1020.0) * f32(4000.0))          # pprho0*ppcp: r4 * r4, exact here
    for t in range(npt):
        dvol = dsum = 0.0
        for k in range(npk):
            dvol, dsum = dvol + sums[t, k, 1], dsum + sums[t, k, 2]
            var[t, 1, k] = c * sums[t, k, 2]
        var[t, 0, 0] = c * dsum
        var[t, 2, 0] = var[t, 0, 0] / dvol
    return var


@pytest.mark.parametrize("name", lc.HEATC_CASES)
def test_oracle_heatc_r4sum_equals_reference_statements(name):
    """The reference's SUM is r4 (:225-229); amdflang -O1 inlines it as a sequential r4 loop, like gfortran -O.  The
    oracle's r4sum mode reproduces it bit for bit; its default mode (r4 products, r8 sums -- the comparator of the fast
    HIP path) stays inside the rigorous bound of that r4 sum."""
    G = gold("cdfheatc")
    o4 = heatc_oracle(name, True)
    assert same(o4, G[f"{name}.sums"])
    assert same(heatc_vars(o4), G[f"{name}.var"])
    o8 = heatc_oracle(name, False)
    g, temp, full, mxlopt, mxl = lc.heatc_case(name)
    kw = dict(e31d=g["e3t_1d"]) if full else dict(e3t=g["e3t_0"])
    n = g["e1t"].size
    for t in range(temp.shape[0]):
        absb = ot.heatc(g["e1t"], g["e2t"], g["tmask"], np.abs(temp[t]), mxl=None if mxl is None else mxl[t], gdepw=g["gdepw_1d"],
                        mxlopt=mxlopt, **kw)
        assert np.all(np.abs(o8[t] - G[f"{name}.sums"][t]) <= n * 2.0 ** -24 * absb + 1e-300)


# ================================================================================================ cdfmhst
def mhst_masks5(g, nb):
    m = np.zeros((4 if nb == 5 else 1,) + g["e1v"].shape, f32)
    m[0] = g["vmask"][0]
    if nb == 5:
        m[1], m[2], m[3] = g["tmaskatl"], g["tmaskpac"], g["tmaskind"]
    return m


def mhst_vars(heat, salt, nb, zdim, npk):
    """src/cdfmhst.f90:394-441: /1.d15 (PW) and /1.d6, exact zeros -> ppspval, REAL(); order glo atl inp ind pac inp0."""
    hh = heat if zdim else heat[None]
    ss = salt if zdim else salt[None]

    def out(x, div):
        d = x / div
        return np.where(d == 0, np.float64(f32(9999.99)), d).astype(f32)
    res = np.zeros((12, npk, heat.shape[-2]), f32)
    for base, x, div in ((0, hh, 1e15), (6, ss, 1e6)):
        lst = [out(x[..., 0], div)]
        if nb == 5:        # oracle basin order: glo atl pac ind
            lst += [out(x[..., 1], div), out(x[..., 3] + x[..., 2], div), out(x[..., 3], div), out(x[..., 2], div),
                    out(x[..., 0] - x[..., 1], div)]
        else:
            base = 0 if base == 0 else 1
        for i, a in enumerate(lst):
            res[base + i, : a.shape[0]] = a
    return res


def mhst_oracle_args(g, nb, kw):
    a = {k: kw[k] for k in ("vt", "vs", "zv", "zt", "zs") if k in kw}
    mesh = dict(e31d=g["e3t_1d"]) if kw["full"] else dict(e3v=g["e3v_0"])
    return a, mesh


@pytest.mark.parametrize("name", lc.MHST_CASES)
def test_oracle_mhst_equals_reference_statements(name):
    g, nb, kw = lc.mhst_case(name)
    G = gold("cdfmhst")
    a, mesh = mhst_oracle_args(g, nb, kw)
    m = mhst_masks5(g, nb)
    npt = next(iter(a.values())).shape[0]
    npk = g["tmask"].shape[0]
    for t in range(npt):
        h, s = ot.mhst(g["e1v"], m, zdim=kw["zdim"], **mesh, **{k: v[t] for k, v in a.items()})
        last_h, last_s = (h[-1], s[-1]) if kw["zdim"] else (h, s)
        nz = m.shape[0]
        assert same(last_h.T, G[f"{name}.zon"][t, 0:nz]) and same(last_s.T, G[f"{name}.zon"][t, 4:4 + nz])
        assert same(mhst_vars(h, s, nb, kw["zdim"], npk), G[f"{name}.var"][t])


# ================================================================================================ cdfpsi
def psi_by(psi, psi_open, name, case=lc.psi_case):
    """The record loop of src/cdfpsi.f90:256-439 written over a psi(...) / psi_open(...) pair (the oracle's or the
    HIP path's): -> var (npt,3,nlo,npj,npi) r4 and dpsi (npt,npj,npi) r8 laid out like the harness outputs.
    With `e3u_t` / `e3v_t` in the case (-vvl) record t takes the metrics of record t (:257-262, :275, :294)."""
    g, kw = case(name)
    npk, npj, npi = g["tmask"].shape
    zu, zv = kw.get("zu"), kw.get("zv")
    npt = (zu if zu is not None else zv).shape[0]
    lev = kw.get("lev", False)
    var = np.zeros((npt, 3, npk if lev else 1, npj, npi), f32)
    dpsi = np.zeros((npt, npj, npi))
    fm = g["fmask"][0].astype(f32) if kw.get("mask") else None
    e3 = dict(e31d=g["e3t_1d"]) if kw.get("full") else {}
    for t in range(npt):
        a = {}
        if zu is not None:
            a.update(zu=zu[t], e2u=g["e2u"])
            if not kw.get("full"):
                a["e3u"] = g["e3u_0"] if kw.get("e3u_t") is None else kw["e3u_t"][t]
        if zv is not None:
            a.update(zv=zv[t], e1v=g["e1v"])
            if not kw.get("full"):
                a["e3v"] = g["e3v_0"] if kw.get("e3v_t") is None else kw["e3v_t"][t]
        if lev:                                   # -lev: one scan per level, no reference point, first record only (:311-315)
            if t > 0:
                break
            for k in range(npk):
                b = {x: (y[k:k + 1] if y.ndim == 3 else y) for x, y in a.items()}
                var[t, 0, k] = psi(1, iiref=npi, ijref=1, **b)[0]
            continue
        if kw.get("opn"):
            p4, d8 = psi_open(npk, a["zu"], a["e2u"], a["e3u"], a["zv"], a["e1v"], a["e3v"], lmean=kw.get("mean", False),
                              iiref=kw.get("iiref"), ijref=kw.get("ijref"), fmask=fm)
        else:
            p4, d8 = psi(npk, iiref=kw.get("iiref"), ijref=kw.get("ijref"), fmask=fm, **a, **e3)
        var[t, 0, 0], dpsi[t] = p4, d8
        if kw.get("ssh") is not None:             # :264-266, :280, :299: the level-1 v times the SSH at U / V points
            ssh = kw["ssh"][t]
            sshu, sshv = np.zeros((npj, npi), f32), np.zeros((npj, npi), f32)
            sshu[:, :-1] = f32(0.5) * (ssh[:, 1:] + ssh[:, :-1])
            sshv[:-1, :] = f32(0.5) * (ssh[1:, :] + ssh[:-1, :])
            v1 = zv[t][0:1]
            _, s8 = psi(1, zu=v1, e2u=g["e2u"], e3u=sshu[None], zv=v1, e1v=g["e1v"], e3v=sshv[None], iiref=npi, ijref=1)
            var[t, 1, 0] = s8.astype(f32)                       # SNGL(dpsissh): not normalised (:423 touches dpsi only)
            var[t, 2, 0] = (s8 + d8).astype(f32)                # SNGL(dpsissh + dpsi)
    return var, dpsi


@pytest.mark.parametrize("name", lc.PSI_CASES)
def test_oracle_psi_equals_reference_statements(name):
    G = gold("cdfpsi")
    var, dpsi = psi_by(lambda npk, **k: ot.psi(npk, **k), ot.psi_open, name)
    assert same(var, G[f"{name}.var"])
    if f"{name}.dpsi" in G:
        assert same(dpsi, G[f"{name}.dpsi"])


# ================================================================================================ cdfvT / cdfcurl
def vt_oracle(name):
    u, v, t, s, e3u, e3v = lc.vt_case(name)
    fl = lambda x: None if x is None else x.reshape((-1,) + x.shape[2:])     # noqa: E731  tags x frames -> frames, file order
    u, v, t, s, e3u, e3v = map(fl, (u, v, t, s, e3u, e3v))
    npk = u.shape[1]
    lv = [ot.vt_level(u[:, k], v[:, k], t[:, k], s[:, k], None if e3u is None else e3u[:, k], None if e3v is None else e3v[:, k])
          for k in range(npk)]
    out = np.zeros((6, npk) + u.shape[2:])
    for i, key in enumerate(("vt", "vs", "ut", "us", "e3u", "e3v")):
        if key in lv[0]:
            out[i] = np.stack([x[key] for x in lv])
    return out


@pytest.mark.parametrize("name", lc.VT_CASES)
def test_oracle_vt_equals_reference_statements(name):
    assert same(vt_oracle(name), gold("cdfvt")[name])


def curl_inputs(name):
    g, u, v, forcing, tp, overf = lc.curl_case(name)
    npi = g["e1u"].shape[1]
    lperio = bool(g["glamf"][0, 0] == g["glamf"][0, npi - 2])               # src/cdfcurl.f90:238
    if forcing:                                                             # :250-258, A grid -> U / V points
        un, vn = np.zeros_like(u), np.zeros_like(v)
        un[..., :-1] = f32(0.5) * (u[..., :-1] + u[..., 1:])
        vn[..., :-1, :] = f32(0.5) * (v[..., :-1, :] + v[..., 1:, :])
        u, v = un, vn
    return g, u, v, lperio, tp, lc.curl_ff(g) if overf else None


@pytest.mark.parametrize("name", lc.CURL_CASES)
def test_oracle_curl_equals_reference_statements(name):
    g, u, v, lperio, tp, ff = curl_inputs(name)
    ref = np.stack([np.stack([ot.curl(u[t, k], v[t, k], g["e1u"], g["e2v"], g["e1f"], g["e2f"], lperio, tp, ff)
                              for k in range(u.shape[1])]) for t in range(u.shape[0])])
    want = gold("cdfcurl")[name]
    assert same(ref.astype(want.dtype), want)


# ================================================================================================ live pin
needs_ref = pytest.mark.skipif(not rl.available(), reason="oracle/_ref/liblift_ref.so is only built where /root/reference exists")


@needs_ref
def test_goldens_are_what_the_harness_returns_now():
    """The committed goldens are reproduced by the harness built in this container (guards against a stale file)."""
    g, zv, nb, full, dec, teos = lc.moc_case("small5")
    assert same(rl.cdfmoc(g, zv, nbasins=nb)["dmoc"], gold("cdfmoc")["small5.dmoc"])
    g, kw = lc.psi_case("open_mean")
    assert same(rl.cdfpsi(g, **kw)["dpsi"], gold("cdfpsi")["open_mean.dpsi"])
    assert same(rl.cdfvt(*lc.vt_case("vvl")), gold("cdfvt")["vvl"])


@needs_ref
@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_equals_harness_on_fresh_cases(seed):
    """Random grids that are in no golden file: cdfmoc, cdfmhst, cdfpsi, cdfcurl, cdfheatc against the live harness."""
    import synth
    rng = np.random.default_rng(seed)
    npi, npj, npk = int(rng.integers(5, 60)), int(rng.integers(4, 40)), int(rng.integers(2, 12))
    g = synth.make_grid(npi, npj, npk, seed=1000 + seed)
    t, s, u, v = lc.tracers(g, 2, 70 + seed)
    e1v, e3v, ib = synth.moc_inputs(g)
    assert same(oracle_lib.moc(e1v, e3v, ib, v), rl.cdfmoc(g, v)["dmoc"])
    m = mhst_masks5(g, 5)
    r = rl.cdfmhst(g, zv=v, zt=t, zs=s)
    for k in range(2):
        h, sa = ot.mhst(g["e1v"], m, e3v=g["e3v_0"], zv=v[k], zt=t[k], zs=s[k])
        assert same(h.T, r["zon"][k, 0:4]) and same(sa.T, r["zon"][k, 4:8])
    r = rl.cdfpsi(g, zu=u, zv=v, mean=True)
    for k in range(2):
        p4, d8 = ot.psi(npk, zu=u[k], e2u=g["e2u"], e3u=g["e3u_0"], zv=v[k], e1v=g["e1v"], e3v=g["e3v_0"])
        assert same(d8, r["dpsi"][k]) and same(p4, r["var"][k, 0, 0])
    lperio = bool(g["glamf"][0, 0] == g["glamf"][0, npi - 2])
    r = rl.cdfcurl(g, u[:, :2], v[:, :2], tpoint=True)
    ref = np.stack([np.stack([ot.curl(u[k, l], v[k, l], g["e1u"], g["e2v"], g["e1f"], g["e2f"], lperio, True) for l in range(2)])
                    for k in range(2)])
    assert same(ref, r)
    r = rl.cdfheatc(g, t)
    for k in range(2):
        assert same(ot.heatc(g["e1t"], g["e2t"], g["tmask"], t[k], e3t=g["e3t_0"], r4sum=True), r["sums"][k])


# ================================================================================================ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.MOC_CASES)
def test_hip_moc_against_reference_statements(gpu, name):
    g, zv, nb, full, dec, teos, e1v, e3v, ib = moc_oracle(name)
    G = gold("cdfmoc")
    npk, npj, npi = e3v.shape
    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    bound = -oracle_lib.moc(e1v, e3v, ib, np.abs(zv))
    if dec is None:
        got = plan.compute(zv)
        assert np.all(np.abs(got - G[f"{name}.dmoc"]) <= 1e-13 * bound + 1e-300)
        assert np.all(np.abs(moc_vars(got, nb).astype(np.float64) - G[f"{name}.var"]) <= ulp4(G[f"{name}.var"]))
        return
    from test_moc_decomp import decomp_bounds, within_bounds
    plan.decomp_set_mesh(e3v, g["e1u"], lc.moc_zcoef(g), g["gdept_1d"], g["tmask"], g["umask"])
    sh = shb = None
    for t in range(zv.shape[0]):
        o = plan.decomp(zv[t], dec["t"][t], dec["s"][t], dmoc_sh=sh, eos=gpu.EOS_TEOS10 if teos else gpu.EOS_80)
        ref = {"total": G[f"{name}.dmoc"][t], "sh": G[f"{name}.dec"][t, 0], "bt": G[f"{name}.dec"][t, 1], "ag": G[f"{name}.dec"][t, 2]}
        a = dict(e1v=e1v, e3v=e3v, ibmask=ib, zv=zv[t], ztemp=dec["t"][t], zsal=dec["s"][t], tmask=g["tmask"].astype(np.uint8),
                 umask=g["umask"].astype(np.uint8), e1u=g["e1u"], zcoef=lc.moc_zcoef(g), gdept=g["gdept_1d"].astype(np.float32))
        # element by element, 1e-13 of the sum of the absolute values of its terms (tests/test_moc_decomp.py); the shear part
        # is accumulated onto the one of the record before (:471) -- the GPU's onto its own, the reference's onto its own:
        # that one's terms count too
        b = decomp_bounds(a, teos, prior_bound=shb)
        within_bounds(o, ref, b)
        sh, shb = o["sh"], b["sh"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.HEATC_CASES)
def test_hip_heatc_against_reference_statements(gpu, name):
    g, temp, full, mxlopt, mxl = lc.heatc_case(name)
    G = gold("cdfheatc")
    npk, npj, npi = g["tmask"].shape
    plan = gpu.HeatcPlan(npi, npj, npk)
    hmask = g["tmask"] if name == "rmask" else g["tmask"].astype(np.uint8)      # "rmask": an r4 mask that is not 0/1 (-M)
    plan.set_mesh(g["e1t"], g["e2t"], hmask, **(dict(e31d=g["e3t_1d"]) if full else dict(e3t=g["e3t_0"])))
    n = npi * npj
    for t in range(temp.shape[0]):
        got = plan.compute(temp[t], rmxldep=None if mxl is None else mxl[t], gdepw=g["gdepw_1d"], mxloption=mxlopt)
        kw = dict(e31d=g["e3t_1d"]) if full else dict(e3t=g["e3t_0"])
        absb = ot.heatc(g["e1t"], g["e2t"], g["tmask"], np.abs(temp[t]), mxl=None if mxl is None else mxl[t], gdepw=g["gdepw_1d"],
                        mxlopt=mxlopt, **kw)
        assert np.all(np.abs(got - G[f"{name}.sums"][t]) <= n * 2.0 ** -24 * absb + 1e-300)
    # the reference-order mode (sequential r4 sums): what the reference's statements return, bit for bit -- on one
    # device and chained through 2 and 3 row blocks
    for devices in (None, [0, 0], [0, 0, 0]):
        seq = gpu.HeatcPlan(npi, npj, npk, devices=devices)
        seq.set_mesh(g["e1t"], g["e2t"], hmask, **(dict(e31d=g["e3t_1d"]) if full else dict(e3t=g["e3t_0"])))
        seq.set_sum_mode(True)
        got = np.stack([seq.compute(temp[t], rmxldep=None if mxl is None else mxl[t], gdepw=g["gdepw_1d"], mxloption=mxlopt)
                        for t in range(temp.shape[0])])
        assert same(got, G[f"{name}.sums"]), devices
        assert same(heatc_vars(got), G[f"{name}.var"]), devices


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.MHST_CASES)
def test_hip_mhst_against_reference_statements(gpu, name):
    g, nb, kw = lc.mhst_case(name)
    G = gold("cdfmhst")
    a, mesh = mhst_oracle_args(g, nb, kw)
    m = mhst_masks5(g, nb)
    npk, npj, npi = g["tmask"].shape
    plan = gpu.MhstPlan(npi, npj, npk, m.shape[0])
    plan.set_mesh(g["e1v"], m, **mesh)
    for t in range(next(iter(a.values())).shape[0]):
        rec = {k: v[t] for k, v in a.items()}
        h, s = plan.compute(zdim=kw["zdim"], **rec)
        bh, bs = ot.mhst(g["e1v"], m, zdim=kw["zdim"], **mesh, **{k: np.abs(v) for k, v in rec.items()})
        lh, ls, lbh, lbs = (h[-1], s[-1], bh[-1], bs[-1]) if kw["zdim"] else (h, s, bh, bs)
        nz = m.shape[0]
        assert np.all(np.abs(lh.T - G[f"{name}.zon"][t, 0:nz]) <= 1e-13 * lbh.T + 1e-300)
        assert np.all(np.abs(ls.T - G[f"{name}.zon"][t, 4:4 + nz]) <= 1e-13 * lbs.T + 1e-300)
        want = G[f"{name}.var"][t]
        got = mhst_vars(h, s, nb, kw["zdim"], npk)
        assert np.all(np.abs(got.astype(np.float64) - want) <= 2 * ulp4(want))       # glo - atl: two roundings


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.PSI_CASES)
def test_hip_psi_against_reference_statements(gpu, name):
    g, _ = lc.psi_case(name)
    npk, npj, npi = g["tmask"].shape
    plans = {}

    def plan(n):
        if n not in plans:
            plans[n] = gpu.PsiPlan(npi, npj, n)
        return plans[n]

    def psi(n, **k):
        return plan(n).compute(**k)

    def psi_open(n, zu, e2u, e3u, zv, e1v, e3v, lmean=False, iiref=None, ijref=None, fmask=None):
        p = plan(n)
        p.integrate(zu=zu, e2u=e2u, e3u=e3u, zv=zv, e1v=e1v, e3v=e3v)
        p.scan_open(lmean)
        return p.finish(None, npi if iiref is None else iiref, npj if ijref is None else ijref, fmask)
    G = gold("cdfpsi")
    var, dpsi = psi_by(psi, psi_open, name)
    assert same(var, G[f"{name}.var"])
    if f"{name}.dpsi" in G:
        assert same(dpsi, G[f"{name}.dpsi"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.VT_CASES)
def test_hip_vt_against_reference_statements(gpu, name):
    u, v, t, s, e3u, e3v = lc.vt_case(name)
    fl = lambda x: None if x is None else np.ascontiguousarray(x.reshape((-1,) + x.shape[2:]))     # noqa: E731
    o = gpu.vt_compute(*map(fl, (u, v, t, s)), e3u=fl(e3u), e3v=fl(e3v))
    want = gold("cdfvt")[name]
    for i, key in enumerate(("vt", "vs", "ut", "us", "e3u", "e3v")):
        if key in o:
            assert same(o[key], want[i]), key


@pytest.mark.gpu
@pytest.mark.parametrize("name", lc.CURL_CASES)
def test_hip_curl_against_reference_statements(gpu, name):
    g, u, v, lperio, tp, ff = curl_inputs(name)
    npj, npi = g["e1u"].shape
    plan = gpu.CurlPlan(npi, npj)
    plan.set_mesh(g["e1u"], g["e2v"], g["e1f"], g["e2f"], ff=ff, lperio=lperio)
    want = gold("cdfcurl")[name]
    got = np.stack([plan.compute(u[t], v[t], ltpoint=tp) for t in range(u.shape[0])])
    assert same(got.astype(want.dtype), want)
