"""cdfmhst (src/cdfmhst.f90:308-367) and cdfpsi (src/cdfpsi.f90:271-438).

Stated tolerances:
  cdfpsi  : the vertical sums and both scans are sequential in the reference's order -> the HIP result
            must equal the oracle BIT FOR BIT (r8 field and the r4 output).
  cdfmhst : vertical sums bit-exact; zonal sums in a fixed tree order instead of sequentially:
            |d| <= 1e-13 * sum|terms| per output.
"""
import numpy as np
import pytest

import oracle_tools
import synth


def tracer_fields(g, seed):
    rng = np.random.default_rng(seed)
    npk, npj, npi = g["tmask"].shape
    t = (-2 + 32 * rng.random((npk, npj, npi), dtype=np.float32)).astype(np.float32) * g["tmask"]
    s = (30 + 8 * rng.random((npk, npj, npi), dtype=np.float32)).astype(np.float32) * g["tmask"]
    v = synth.velocity(g, 1, seed + 1)[0]
    u = synth.velocity(g, 1, seed + 2, which="u")[0]
    return t.astype(np.float32), s.astype(np.float32), u, v


def mhst_masks(g, nb):
    m = np.zeros((nb,) + g["e1v"].shape, np.float32)
    m[0] = g["vmask"][0]
    if nb >= 4:
        m[1], m[2], m[3] = g["tmaskatl"], g["tmaskpac"], g["tmaskind"]
    return m


# ----------------------------------------------------------------------------- CPU
def test_oracle_psi_matches_numpy_restatement_bitwise():
    g = synth.make_grid(19, 13, 5, seed=1)
    _, _, u, v = tracer_fields(g, 3)
    _, d8u = oracle_tools.psi(5, zu=u, e2u=g["e2u"], e3u=g["e3u_0"])
    tr = np.cumsum(((u * g["e2u"][None]).astype(np.float32) * g["e3u_0"]).astype(np.float32).astype(np.float64), axis=0)[-1]
    want = np.zeros_like(tr)
    want[1:] = np.cumsum(-tr[1:], axis=0)
    want = want - want[-1, -1]
    assert np.array_equal(d8u, want)
    _, d8v = oracle_tools.psi(5, zv=v, e1v=g["e1v"], e3v=g["e3v_0"])
    trv = np.cumsum(((v * g["e1v"][None]).astype(np.float32) * g["e3v_0"]).astype(np.float32).astype(np.float64), axis=0)[-1]
    wv = np.zeros_like(trv)
    wv[:, :-1] = np.cumsum(-trv[:, -2::-1], axis=1)[:, ::-1]
    assert np.array_equal(d8v, wv - wv[-1, -1])


def test_oracle_mhst_sep_equals_vt_path_fed_with_its_own_products():
    g = synth.make_grid(17, 11, 4, seed=2)
    t, s, _, v = tracer_fields(g, 5)
    vt = np.zeros_like(t)
    vs = np.zeros_like(s)
    vt[:, :-1] = (np.float32(0.5) * (t[:, :-1] + t[:, 1:])).astype(np.float32) * v[:, :-1]
    vs[:, :-1] = (np.float32(0.5) * (s[:, :-1] + s[:, 1:])).astype(np.float32) * v[:, :-1]
    m = mhst_masks(g, 4)
    a = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], vt=vt, vs=vs)
    b = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], zv=v, zt=t, zs=s)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    z = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], vt=vt, vs=vs, zdim=True)
    assert np.array_equal(z[0][-1], a[0]) and np.array_equal(z[1][-1], a[1])     # last level = total


# ----------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("scan", ["ring", "workgroup", "wave"])
@pytest.mark.parametrize("npi,npj,npk", [(24, 18, 7), (182, 149, 31), (1, 1, 1), (5, 3, 2), (130, 67, 4), (64, 64, 3),
                                         (70, 300, 2), (129, 257, 1), (65, 128, 1), (3, 129, 2), (17, 1100, 1), (33, 513, 2),
                                         (40, 120, 1), (40, 121, 2), (31, 239, 1), (32, 240, 1), (33, 241, 1), (100, 481, 1)])
def test_hip_psi_equals_oracle_bitwise(gpu, monkeypatch, scan, npi, npj, npk):
    # the j scan: one chain wave fed through a ring of LDS slots by 15 others (what runs), 16 waves that stage, walk and store 128
    # rows in turn (round 3), or one wave per 64 columns
    monkeypatch.setenv("CDFT_PSI_SCAN_WAVE", {"ring": "0", "wave": "1", "workgroup": "2"}[scan])
    g = synth.make_grid(npi, npj, npk, seed=npi)
    _, _, u, v = tracer_fields(g, 7)
    plan = gpu.PsiPlan(npi, npj, npk)
    cases = [
        dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"]),                                         # default
        dict(zv=v, e1v=g["e1v"], e3v=g["e3v_0"]),                                         # -V
        dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], zv=v, e1v=g["e1v"], e3v=g["e3v_0"]),     # -mean
        dict(zu=u, e2u=g["e2u"], e31d=g["e3t_1d"]),                                       # -full
        dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], fmask=g["fmask"][0].astype(np.float32),
             iiref=max(1, npi // 2), ijref=max(1, npj // 3)),                             # -mask -ref
    ]
    for kw in cases:
        p4, d8 = plan.compute(**kw)
        r4, r8 = oracle_tools.psi(npk, **kw)
        assert np.array_equal(d8, r8), list(kw)
        assert np.array_equal(p4, r4), list(kw)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,blocks", [(24, 18, 7, 2), (182, 149, 5, 3), (70, 9, 2, 8), (5, 2, 1, 2), (66, 700, 1, 3)])
def test_hip_psi_row_blocks_chained_equal_the_whole_domain_bitwise(gpu, npi, npj, npk, blocks):
    """The staged entry points (integrate / scan_u with a carry / value / finish with a given
    reference value) used by cdftools_amd.shard.psi_chain: any number of row blocks, chained in
    order, reproduces the one-plan result bit for bit."""
    from cdftools_amd import shard

    g = synth.make_grid(npi, npj, npk, seed=npi + 3)
    _, _, u, v = tracer_fields(g, 5)
    fm = np.minimum(g["fmask"][0], 1).astype(np.float32)
    iiref, ijref = max(1, npi // 2), max(1, (2 * npj) // 3)
    for lu, lv in ((True, False), (False, True), (True, True)):
        kw = {}
        if lu:
            kw.update(zu=u, e2u=g["e2u"], e3u=g["e3u_0"])
        if lv:
            kw.update(zv=v, e1v=g["e1v"], e3v=g["e3v_0"])
        w4, w8 = gpu.PsiPlan(npi, npj, npk).compute(iiref=iiref, ijref=ijref, fmask=fm, **kw)
        plans, carry, ref = [], None, None
        for r in range(blocks):
            j0, j1 = shard.row_block(npj, blocks, r)
            p = gpu.PsiPlan(npi, j1 - j0, npk)
            p.integrate(**{k: (x[:, j0:j1] if x.ndim == 3 else x[j0:j1]) for k, x in kw.items()})
            carry = p.scan_u(carry, first_block=(r == 0))
            if j0 < ijref <= j1:
                ref = p.value(iiref, ijref - j0)
            plans.append((p, j0, j1))
        got4, got8 = np.empty_like(w4), np.empty_like(w8)
        for p, j0, j1 in plans:
            got4[j0:j1], got8[j0:j1] = p.finish(ref, fmask=fm[j0:j1])
        assert np.array_equal(got8, w8) and np.array_equal(got4, w4), (lu, lv)
    with pytest.raises(gpu.CdftError):
        gpu.PsiPlan(4, 4, 2).finish(0.0)                                # nothing integrated yet


def test_oracle_psi_open_matches_numpy_restatement_bitwise():
    g = synth.make_grid(24, 18, 5, seed=6)
    _, _, u, v = tracer_fields(g, 2)
    npk, npj, npi = u.shape
    dtu, dtv = np.zeros((npj, npi)), np.zeros((npj, npi))
    for k in range(npk):
        dtu += ((u[k] * g["e2u"]) * g["e3u_0"][k]).astype(np.float64)
        dtv += ((v[k] * g["e1v"]) * g["e3v_0"][k]).astype(np.float64)
    J, I = npj - 3, npi - 3
    pu = np.zeros((npj, npi))
    pu[J] = np.cumsum(dtv[J])                                   # sequential prefix along i (np.cumsum adds in order)
    for j in range(J - 1, -1, -1):
        pu[j] = pu[j + 1] + dtu[j + 1]
    pv = np.zeros((npj, npi))
    acc = dtu[npj - 1, I]
    pv[npj - 1, I] = acc
    for j in range(npj - 2, -1, -1):
        acc = acc + dtu[j + 1, I]
        pv[j, I] = acc
    for i in range(I - 1, -1, -1):
        pv[:, i] = pv[:, i + 1] - dtv[:, i + 1]
    kw = dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], zv=v, e1v=g["e1v"], e3v=g["e3v_0"])
    _, d8 = oracle_tools.psi_open(npk, **kw, noref=True)
    assert np.array_equal(d8, pu)
    _, d8m = oracle_tools.psi_open(npk, **kw, lmean=True, noref=True)
    assert np.array_equal(d8m, 0.5 * (pu + pv))
    p4, d8r = oracle_tools.psi_open(npk, **kw, iiref=4, ijref=3)
    assert np.array_equal(d8r, pu - pu[2, 3]) and np.array_equal(p4, (pu - pu[2, 3]).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk", [(24, 18, 7), (182, 149, 5), (3, 3, 2), (130, 67, 3)])
def test_hip_psi_open_equals_oracle_bitwise(gpu, npi, npj, npk):
    g = synth.make_grid(npi, npj, npk, seed=npi + 11)
    _, _, u, v = tracer_fields(g, 3)
    fm = np.minimum(g["fmask"][0], 1).astype(np.float32)
    kw = dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], zv=v, e1v=g["e1v"], e3v=g["e3v_0"])
    plan = gpu.PsiPlan(npi, npj, npk)
    for lmean in (False, True):
        plan.integrate(**kw)
        plan.scan_open(lmean)
        g4, g8 = plan.finish(None, max(1, npi // 2), max(1, npj // 3), fm)
        r4, r8 = oracle_tools.psi_open(npk, **kw, lmean=lmean, iiref=max(1, npi // 2), ijref=max(1, npj // 3), fmask=fm)
        assert np.array_equal(g8, r8) and np.array_equal(g4, r4), lmean
        plan.integrate(**kw)
        plan.scan_open(lmean)
        g4, g8 = plan.finish(0.0)                                # the SSH part: no reference point
        r4, r8 = oracle_tools.psi_open(npk, **kw, lmean=lmean, noref=True)
        assert np.array_equal(g8, r8) and np.array_equal(g4, r4), lmean
    only_u = gpu.PsiPlan(npi, npj, npk)
    only_u.integrate(zu=u, e2u=g["e2u"], e3u=g["e3u_0"])
    with pytest.raises(gpu.CdftError):
        only_u.scan_open()


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb", [(24, 18, 7, 4), (182, 149, 31, 4), (182, 149, 31, 1), (1, 2, 1, 1),
                                            (7, 3, 2, 4), (1442, 9, 3, 4), (333, 5, 4, 8)])
def test_hip_mhst_matches_oracle(gpu, npi, npj, npk, nb):
    import torch

    g = synth.make_grid(npi, npj, npk, seed=npi + 1)
    t, s, _, v = tracer_fields(g, 9)
    m = mhst_masks(g, nb)
    vt = ((np.float32(0.5) * (t + np.roll(t, -1, axis=1))).astype(np.float32) * v).astype(np.float32)
    vs = ((np.float32(0.5) * (s + np.roll(s, -1, axis=1))).astype(np.float32) * v).astype(np.float32)
    plan = gpu.MhstPlan(npi, npj, npk, nb)
    plan.set_mesh(g["e1v"], m, e3v=g["e3v_0"])

    def check(got, ref, bound):
        for k in (0, 1):
            assert np.all(np.abs(got[k] - ref[k]) <= 1e-13 * bound[k] + 1e-300)

    for zdim in (False, True):
        ref = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], vt=vt, vs=vs, zdim=zdim)
        bound = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], vt=np.abs(vt), vs=np.abs(vs), zdim=zdim)
        check(plan.compute(vt=vt, vs=vs, zdim=zdim), ref, bound)
    # separate V, T, S files; arrays resident in HBM
    ref = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], zv=v, zt=t, zs=s)
    bound = oracle_tools.mhst(g["e1v"], m, e3v=g["e3v_0"], zv=np.abs(v), zt=np.abs(t), zs=np.abs(s))
    dv, dt, ds = (torch.from_numpy(x).cuda() for x in (v, t, s))
    check(plan.compute(zv=dv, zt=dt, zs=ds), ref, bound)
    # -full
    plan.set_mesh(g["e1v"], m, e31d=g["e3t_1d"])
    ref = oracle_tools.mhst(g["e1v"], m, e31d=g["e3t_1d"], vt=vt, vs=vs)
    bound = oracle_tools.mhst(g["e1v"], m, e31d=g["e3t_1d"], vt=np.abs(vt), vs=np.abs(vs))
    check(plan.compute(vt=vt, vs=vs), ref, bound)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb", [(2000, 5, 4, 5), (4322, 4, 3, 5), (1537, 3, 2, 1), (300, 7, 5, 8), (4700, 3, 2, 2)])
def test_hip_mhst_zdim_one_kernel_against_the_level_by_level_launches(gpu, monkeypatch, npi, npj, npk, nb):
    """-Zdim from a VT file runs as ONE kernel (a row per workgroup of 256 or 512 threads, the running sums in registers;
    rows beyond 4608 cells keep the launch pair per level).  Against the oracle, against the level-by-level path
    (CDFT_MHST_ZDIM_LEVELS=1: the zonal sums are grouped differently, the same bound), with e3v and with -full."""
    g = synth.make_grid(npi, npj, npk, seed=npi + 5)
    t, s, _, v = tracer_fields(g, 3)
    m = mhst_masks(g, nb)
    vt = ((np.float32(0.5) * (t + np.roll(t, -1, axis=1))).astype(np.float32) * v).astype(np.float32)
    vs = ((np.float32(0.5) * (s + np.roll(s, -1, axis=1))).astype(np.float32) * v).astype(np.float32)
    plan = gpu.MhstPlan(npi, npj, npk, nb)
    for mesh in (dict(e3v=g["e3v_0"]), dict(e31d=g["e3t_1d"])):
        plan.set_mesh(g["e1v"], m, **mesh)
        ref = oracle_tools.mhst(g["e1v"], m, vt=vt, vs=vs, zdim=True, **mesh)
        bound = oracle_tools.mhst(g["e1v"], m, vt=np.abs(vt), vs=np.abs(vs), zdim=True, **mesh)
        got = plan.compute(vt=vt, vs=vs, zdim=True)
        monkeypatch.setenv("CDFT_MHST_ZDIM_LEVELS", "1")
        lev = plan.compute(vt=vt, vs=vs, zdim=True)
        monkeypatch.delenv("CDFT_MHST_ZDIM_LEVELS")
        for k in (0, 1):
            assert np.all(np.abs(got[k] - ref[k]) <= 1e-13 * bound[k] + 1e-300)
            assert np.all(np.abs(lev[k] - ref[k]) <= 1e-13 * bound[k] + 1e-300)
            assert np.array_equal(plan.compute(vt=vt, vs=vs, zdim=True)[k], got[k])       # run-to-run identical
        flat = plan.compute(vt=vt, vs=vs)                                               # the last level of -Zdim = the totals
        for k in (0, 1):
            assert np.all(np.abs(got[k][-1] - flat[k]) <= 1e-13 * bound[k][-1] + 1e-300)


@pytest.mark.gpu
def test_hip_mhst_packed_masks_equal_the_r4_masks_bitwise(gpu, monkeypatch):
    """0/1 basin masks are packed to a byte per cell for the zonal sums; CDFT_MHST_R4MASKS=1 keeps the r4 masks.  Same
    products (x*1.d0, x*0.d0), same order: identical bits, a NaN under a zero mask included (0*NaN = NaN in the reference's
    masked sum too).  Masks with other values are never packed."""
    npi, npj, npk, nb = 130, 21, 4, 5
    g = synth.make_grid(npi, npj, npk, seed=12)
    t, s, _, v = tracer_fields(g, 3)
    m = mhst_masks(g, nb)
    vt = (t * v).astype(np.float32)
    vs = (s * v).astype(np.float32)
    vt[1, 7, 9] = np.nan                                     # wherever it falls: every basin sum of row 7 sees it
    plan = gpu.MhstPlan(npi, npj, npk, nb)
    plan.set_mesh(g["e1v"], m, e3v=g["e3v_0"])
    a = plan.compute(vt=vt, vs=vs)
    monkeypatch.setenv("CDFT_MHST_R4MASKS", "1")
    b = plan.compute(vt=vt, vs=vs)
    monkeypatch.delenv("CDFT_MHST_R4MASKS")
    for k in (0, 1):
        assert np.array_equal(a[k], b[k], equal_nan=True)
    assert np.isnan(a[0]).sum() == nb                        # row 7 of every basin, nothing else
    m2 = m.copy()
    m2[1] *= np.float32(0.5)                                 # not a 0/1 mask: the r4 values are used
    plan.set_mesh(g["e1v"], m2, e3v=g["e3v_0"])
    vt[1, 7, 9] = 0.0
    got = plan.compute(vt=vt, vs=vs)
    ref = oracle_tools.mhst(g["e1v"], m2, e3v=g["e3v_0"], vt=vt, vs=vs)
    bound = oracle_tools.mhst(g["e1v"], m2, e3v=g["e3v_0"], vt=np.abs(vt), vs=np.abs(vs))
    for k in (0, 1):
        assert np.all(np.abs(got[k] - ref[k]) <= 1e-13 * bound[k] + 1e-300)


@pytest.mark.gpu
def test_hip_psi_and_mhst_at_eorca025_size(gpu):
    """Full eORCA025 (1442x1207x75) resident in HBM: psi bit-exact on sampled columns/rows against
    the oracle run on the sub-grids that determine them; mhst rows against the oracle."""
    import torch

    npi, npj, npk = 1442, 1207, 75
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(5)
    e2u = 2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen)
    e3u = torch.from_numpy(synth.e3_profile(npk)).to(dev)[:, None, None] * (0.2 + 0.8 * torch.rand((npk, npj, npi), device=dev, generator=gen))
    u = torch.rand((npk, npj, npi), device=dev, generator=gen) - 0.5
    plan = gpu.PsiPlan(npi, npj, npk)
    p4, d8 = plan.compute(zu=u, e2u=e2u, e3u=e3u, iiref=1, ijref=1)   # dpsiu(:,1) = 0: no shift
    # the U path scans along j column by column: a block of columns is an independent sub-problem
    cols = slice(700, 732)
    r4, r8 = oracle_tools.psi(npk, zu=u[:, :, cols].cpu().numpy(), e2u=e2u[:, cols].cpu().numpy(),
                              e3u=e3u[:, :, cols].cpu().numpy(), iiref=1, ijref=1)
    assert np.array_equal(d8[:, cols], r8) and np.array_equal(p4[:, cols], r4)
    a4, a8 = plan.compute(zu=u, e2u=e2u, e3u=e3u, iiref=1, ijref=1)
    assert np.array_equal(a8, d8) and np.array_equal(a4, p4)               # deterministic
    # mhst on the same arrays (vt := u, vs := e3u as arbitrary fields)
    m = (torch.rand((2, npj, npi), device=dev, generator=gen) > 0.4).float()
    mp = gpu.MhstPlan(npi, npj, npk, 2)
    mp.set_mesh(e2u, m, e3v=e3u)
    heat, salt = mp.compute(vt=u, vs=e3u)
    rows = slice(600, 606)
    rh, rs = oracle_tools.mhst(e2u[rows].cpu().numpy(), m[:, rows].cpu().numpy(), e3v=e3u[:, rows].cpu().numpy(),
                               vt=u[:, rows].cpu().numpy(), vs=e3u[:, rows].cpu().numpy())
    bh, bs = oracle_tools.mhst(e2u[rows].cpu().numpy(), m[:, rows].cpu().numpy(), e3v=e3u[:, rows].cpu().numpy(),
                               vt=np.abs(u[:, rows].cpu().numpy()), vs=e3u[:, rows].cpu().numpy())
    assert np.all(np.abs(heat[rows] - rh) <= 1e-13 * bh) and np.all(np.abs(salt[rows] - rs) <= 1e-13 * bs)
