"""BASELINE.json configs 4 and 5 at their full sizes on one MI355X (288 GB of HBM).

  config 4  cdfvT then cdfmhst on synthetic eORCA12 (4322 x 3606 x 75), row-block sharded on 2 and 4 GPUs:
            the whole grid with 6 frames resident (112 GB): cdfvT in one plan -- sampled rows against the oracle (bit
            for bit), properties -- and cdfmhst on its r4 output (what vt.nc holds); then the PRODUCT's sharded path
            (cdft_vt_compute_sharded, a sharded cdfmhst plan) on 2 and 4 row blocks, all mapped to this box's one GPU:
            bit-identical to the one-plan result.
  config 5  cdfmoc + cdfpsi on an eORCA36-scale grid (12960 x 10842 x 75) over 8 GPUs: ONE GPU's row block
            (12960 x 1356 x 75) at full size -- cdfmoc, 12 records, sampled rows against the oracle; cdfpsi on that
            block in one plan and as 8 chained blocks (the sharded plan: the carry row handed from block to block),
            bit-identical, and sampled COLUMNS against the oracle (the U-path scan runs along j inside a column);
            plus cdfpsi on a 12960 x 64 x 75 domain, 8 blocks of 8 rows, against the oracle on every cell;
            and the WHOLE 12960 x 10842 x 75 grid resident on the one GPU (records of more than 2^31 elements):
            cdfmoc with 2 records, cdfcurl, cdfpsi and cdfheatc, sampled rows / columns against the oracle.

Inputs are generated in HBM; comparisons of full fields are made there (torch.equal)."""
import numpy as np
import pytest

import oracle_lib
import oracle_tools
import synth

f32 = np.float32


def bits_checksum(x):
    """order-independent 64-bit checksum of the bit patterns of a tensor (wrap-around sum of its 32-bit words)"""
    import torch
    return int(x.contiguous().view(torch.int32).to(torch.int64).sum().item())


@pytest.mark.gpu
def test_config4_vt_then_mhst_at_eorca12(gpu):
    import torch
    npi, npj, npk, nfr = 4322, 3606, 75, 6
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(4)
    mb = torch.where(torch.rand((npj, npi), device=dev, generator=gen) < 0.35, 0.0,
                     1.0 + torch.floor((npk - 1) * torch.rand((npj, npi), device=dev, generator=gen))).clamp(max=npk - 1)
    k = torch.arange(npk, device=dev, dtype=torch.float32)[:, None, None]
    tmask = (k < mb[None])
    del mb
    fields = []
    for lo, hi, masked in ((-0.5, 0.5, True), (-0.5, 0.5, True), (-2.0, 30.0, True), (30.0, 38.0, True)):
        x = torch.empty((nfr, npk, npj, npi), dtype=torch.float32, device=dev)
        for f in range(nfr):
            x[f].uniform_(lo, hi, generator=gen)
            x[f] *= tmask
        fields.append(x)
    u, v, t, s = fields
    del fields

    # ---- cdfvT, the whole grid in one plan (6 frames, 112 GB resident)
    o = gpu.vt_compute(u, v, t, s, out_r4=True)
    torch.cuda.synchronize()
    rng = np.random.default_rng(0)
    for lev in (0, 36, 74):
        for j in rng.choice(npj - 1, 3, replace=False):
            ref = oracle_tools.vt_level(*(x[:, lev, j:j + 2].cpu().numpy() for x in (u, v, t, s)))
            for key in ("vt", "vs", "ut", "us"):
                assert np.array_equal(o[key][lev, j].cpu().numpy(), ref[key][0].astype(f32)), (lev, j, key)
    for key in ("vt", "vs", "ut", "us"):                     # the reference leaves the last row and column at 0 (:200-206)
        assert not o[key][:, npj - 1].any() and not o[key][:, :, npi - 1].any()
    sums = {key: bits_checksum(o[key]) for key in o}
    # frames are accumulated in file order: scaling all velocities by 2 is exact
    o2 = gpu.vt_compute(u * 2.0, v, t, s, out_r4=True)
    assert torch.equal(o2["ut"], o["ut"] * 2.0) and torch.equal(o2["vt"], o["vt"])
    del o2
    torch.cuda.empty_cache()

    # ---- cdfmhst on vT's r4 output (what vt.nc holds), whole grid, one plan
    e1v = 2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen)
    e3v = (1.0 + 100.0 * torch.rand((npk, npj, npi), device=dev, generator=gen)) * tmask
    band = (torch.arange(npi, device=dev) * 3) // npi
    m = torch.stack([tmask[0].float()] + [tmask[0].float() * (band == b)[None] for b in range(3)])
    vt, vs = o["vt"].clone(), o["vs"].clone()
    del o
    plan = gpu.MhstPlan(npi, npj, npk, 4)
    plan.set_mesh(e1v, m, e3v=e3v)
    heat, salt = plan.compute(vt=vt, vs=vs)
    rows = np.sort(rng.choice(npj, 8, replace=False))
    r = torch.from_numpy(rows).to(dev)
    a = dict(e3v=e3v[:, r].cpu().numpy(), vt=vt[:, r].cpu().numpy(), vs=vs[:, r].cpu().numpy())
    rh, rs = oracle_tools.mhst(e1v[r].cpu().numpy(), m[:, r].cpu().numpy(), **a)
    bh, bs = oracle_tools.mhst(e1v[r].cpu().numpy(), m[:, r].cpu().numpy(), e3v=a["e3v"], vt=np.abs(a["vt"]), vs=np.abs(a["vs"]))
    assert np.all(np.abs(heat[rows] - rh) <= 1e-13 * bh) and np.all(np.abs(salt[rows] - rs) <= 1e-13 * bs)

    # ---- the sharded product path on 2 and 4 row blocks (3 of the frames: the blocks' staging copies double the footprint)
    nsub = 3
    torch.cuda.empty_cache()                                   # the library allocates outside torch's cache
    us, vs_, ts, ss = (x[:nsub] for x in (u, v, t, s))
    ref = gpu.vt_compute(us, vs_, ts, ss, out_r4=True)
    ref_sums = {key: bits_checksum(ref[key]) for key in ref}
    for devices in ([0, 0], [0, 0, 0, 0]):
        got = gpu.vt_compute(us, vs_, ts, ss, out_r4=True, devices=devices)
        for key in ref:
            assert torch.equal(got[key], ref[key]), (devices, key)
            assert bits_checksum(got[key]) == ref_sums[key]
        del got
        sh = gpu.MhstPlan(npi, npj, npk, 4, devices=devices)
        sh.set_mesh(e1v, m, e3v=e3v)
        h2, s2 = sh.compute(vt=vt, vs=vs)
        assert np.array_equal(h2, heat) and np.array_equal(s2, salt), devices
        h3, s3 = sh.compute(zv=v[0], zt=t[0], zs=s[0])                 # V, T, S form: a halo row between the blocks
        h1, s1 = plan.compute(zv=v[0], zt=t[0], zs=s[0])
        assert np.array_equal(h3, h1) and np.array_equal(s3, s1), devices
        del sh
    assert sums["vt"] != 0


@pytest.mark.gpu
def test_config5_moc_and_psi_on_one_eighth_of_eorca36(gpu):
    import torch
    npi, npj, npk, npt, nb = 12960, 1356, 75, 12, 5
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(5)
    mb = torch.where(torch.rand((npj + 1, npi), device=dev, generator=gen) < 0.35, 0.0,
                     1.0 + torch.floor((npk - 1) * torch.rand((npj + 1, npi), device=dev, generator=gen))).clamp(max=npk - 1)
    k = torch.arange(npk, device=dev, dtype=torch.float32)[:, None, None]
    tm = (k < mb[None])
    vmask = tm[:, :-1] & tm[:, 1:]
    umask = torch.zeros_like(vmask)
    umask[:, :, :-1] = tm[:, :-1, :-1] & tm[:, :-1, 1:]
    t1 = tm[0, :-1].to(torch.int16)
    del mb
    e1v = 2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen)
    e3 = torch.from_numpy(synth.e3_profile(npk)).to(dev)[:, None, None]
    e3v = (e3 * (0.2 + 0.8 * torch.rand((npk, npj, npi), device=dev, generator=gen))) * vmask
    band = (torch.arange(npi, device=dev) * 3) // npi
    ib = torch.zeros((npj, npi, nb), dtype=torch.int16, device=dev)
    ib[:, :, 0] = vmask[0].to(torch.int16)
    ib[:, :, 1] = t1 * (band == 0)[None]
    ib[:, :, 3] = t1 * (band == 2)[None]
    ib[:, :, 4] = t1 * (band == 1)[None]
    ib[:, :, 2] = (ib[:, :, 3] + ib[:, :, 4]).clamp(max=1)
    ib[:, 0, 0] = 0
    ib[:, -1, 0] = 0
    zv = torch.empty((npt, npk, npj, npi), dtype=torch.float32, device=dev)
    for t in range(npt):
        zv[t].uniform_(-0.5, 0.5, generator=gen)
        zv[t] *= vmask

    # ---- cdfmoc: one GPU's share of the eORCA36-scale grid, 12 records
    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    d1 = plan.compute(zv)
    assert torch.equal(plan.compute(zv), d1)
    rows = np.sort(np.random.default_rng(1).choice(npj, 6, replace=False))
    r = torch.from_numpy(rows).to(dev)
    a = (e1v[r].cpu().numpy(), e3v[:, r].cpu().numpy(), ib[r].cpu().numpy(), zv[:, :, r].cpu().numpy())
    ref = oracle_lib.moc(*a)
    bound = -oracle_lib.moc(a[0], a[1], a[2], np.abs(a[3]))
    assert np.all(np.abs(d1[:, :, r].cpu().numpy() - ref) <= 1e-13 * bound + 1e-300)
    # the same block as 2 row blocks of a sharded plan: bit-identical
    sh = gpu.MocPlan(npi, npj, npk, nb, devices=[0, 0])
    sh.set_mesh(e1v, e3v, ib)
    assert torch.equal(sh.compute(zv[:4]), d1[:4])
    del sh, plan, d1

    # ---- cdfpsi on the same block: one plan, and 8 chained blocks (the sharded plan's carry from block to block)
    e2u = 2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen)
    e3u = (e3 * (0.2 + 0.8 * torch.rand((npk, npj, npi), device=dev, generator=gen))) * umask
    zu = (torch.rand((npk, npj, npi), device=dev, generator=gen) - 0.5) * umask
    one = gpu.PsiPlan(npi, npj, npk)
    p4, d8 = one.compute(zu=zu, e2u=e2u, e3u=e3u, host_out=False)
    sh = gpu.PsiPlan(npi, npj, npk, devices=[0] * 8)
    q4, q8 = sh.compute(zu=zu, e2u=e2u, e3u=e3u, host_out=False)
    assert torch.equal(q4, p4) and torch.equal(q8, d8)
    q4, q8 = sh.compute(zu=zu, e2u=e2u, e3u=e3u, zv=zv[0], e1v=e1v, e3v=e3v, iiref=77, ijref=1000, host_out=False)      # -mean, -ref
    p4, p8 = one.compute(zu=zu, e2u=e2u, e3u=e3u, zv=zv[0], e1v=e1v, e3v=e3v, iiref=77, ijref=1000, host_out=False)
    assert torch.equal(q4, p4) and torch.equal(q8, p8)
    # sampled columns against the oracle: the U-path scan runs along j inside a column; reference point = last column
    # of the sample
    cols = np.sort(np.random.default_rng(2).choice(npi, 32, replace=False))
    c = torch.from_numpy(cols).to(dev)
    p4, d8 = one.compute(zu=zu, e2u=e2u, e3u=e3u, iiref=int(cols[-1]) + 1, ijref=npj, host_out=False)
    r4, r8 = oracle_tools.psi(npk, zu=zu[:, :, c].cpu().numpy(), e2u=e2u[:, c].cpu().numpy(), e3u=e3u[:, :, c].cpu().numpy())
    assert np.array_equal(d8[:, c].cpu().numpy(), r8) and np.array_equal(p4[:, c].cpu().numpy(), r4)
    del one, sh


@pytest.mark.gpu
def test_config5_psi_eight_blocks_against_the_oracle_on_every_cell(gpu):
    npi, npj, npk = 12960, 64, 75
    g = synth.make_grid(npi, npj, npk, seed=36)
    u = synth.velocity(g, 1, seed=3, which="u")[0]
    v = synth.velocity(g, 1, seed=4)[0]
    sh = gpu.PsiPlan(npi, npj, npk, devices=[0] * 8)
    for kw in (dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"]),
               dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], zv=v, e1v=g["e1v"], e3v=g["e3v_0"], iiref=6000, ijref=33)):
        g4, g8 = sh.compute(**kw)
        r4, r8 = oracle_tools.psi(npk, **kw)
        assert np.array_equal(g8, r8) and np.array_equal(g4, r4)


@pytest.mark.gpu
def test_config5_whole_eorca36_domain_resident_on_one_gpu(gpu):
    """The WHOLE eORCA36-scale grid (12960 x 10842 x 75 = 1.05e10 cells, 42 GB per r4 field: more than 2^31 elements
    in ONE record) on one MI355X: cdfmoc with 2 records (zv 84 GB + the plan's 42 GB of weights), then cdfpsi, cdfcurl
    and cdfheatc on the same fields.  Rows / columns sampled over the whole j range, the last row included, against
    the oracle; full-field identities where the arithmetic allows them."""
    import torch
    npi, npj, npk, npt, nb = 12960, 10842, 75, 2, 5
    dev = "cuda"
    import gc
    gc.collect()
    torch.cuda.empty_cache()                          # what earlier tests of this process left in torch's cache
    free, total = torch.cuda.mem_get_info()
    if free < 230 * 2**30:
        pytest.skip(f"needs 230 GB of free HBM, this device has {free / 2**30:.0f}")
    gen = torch.Generator(device=dev).manual_seed(36)
    mb = torch.where(torch.rand((npj + 1, npi), device=dev, generator=gen) < 0.35, 0.0,
                     1.0 + torch.floor((npk - 1) * torch.rand((npj + 1, npi), device=dev, generator=gen))).clamp(max=npk - 1)
    e3p = torch.from_numpy(synth.e3_profile(npk)).to(dev)

    def level_mask(k, which):                        # tmask / vmask / umask of level k, built when needed (bool)
        tm = k < mb
        if which == "t":
            return tm[:-1]
        if which == "v":
            return tm[:-1] & tm[1:]
        um = torch.zeros((npj, npi), dtype=torch.bool, device=dev)
        um[:, :-1] = tm[:-1, :-1] & tm[:-1, 1:]
        return um

    def field(lo, hi, which, scale=None):            # (npk,npj,npi) r4, uniform in [lo,hi) [x e3_1d(k)] x mask, level by level
        x = torch.empty((npk, npj, npi), dtype=torch.float32, device=dev)
        for k in range(npk):
            x[k].uniform_(lo, hi, generator=gen)
            if scale is not None:
                x[k] *= scale[k]
            x[k] *= level_mask(k, which)
        return x

    e1v = 2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen)
    t1 = level_mask(0, "t").to(torch.int16)
    band = (torch.arange(npi, device=dev) * 3) // npi
    ib = torch.zeros((npj, npi, nb), dtype=torch.int16, device=dev)
    ib[:, :, 0] = level_mask(0, "v").to(torch.int16)
    ib[:, :, 1] = t1 * (band == 0)[None]
    ib[:, :, 3] = t1 * (band == 2)[None]
    ib[:, :, 4] = t1 * (band == 1)[None]
    ib[:, :, 2] = (ib[:, :, 3] + ib[:, :, 4]).clamp(max=1)
    ib[:, 0, 0] = 0
    ib[:, -1, 0] = 0
    rows = np.unique(np.concatenate([[0, 1, npj // 2, npj - 2, npj - 1], np.random.default_rng(36).choice(npj, 5, replace=False)]))
    r = torch.from_numpy(rows).to(dev)

    # ---- cdfmoc, 2 records
    e3v = field(0.2, 1.0, "v", scale=e3p)
    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    e3v_rows = e3v[:, r].cpu().numpy()
    e3v_last = e3v[:, -2:].clone()
    del e3v
    torch.cuda.empty_cache()
    zv = torch.empty((npt, npk, npj, npi), dtype=torch.float32, device=dev)
    for t in range(npt):
        for k in range(npk):
            zv[t, k].uniform_(-0.5, 0.5, generator=gen)
            zv[t, k] *= level_mask(k, "v")
    d = plan.compute(zv)
    assert torch.equal(plan.compute(zv), d)
    a = (e1v[r].cpu().numpy(), e3v_rows, ib[r].cpu().numpy(), zv[:, :, r].cpu().numpy())
    ref = oracle_lib.moc(*a)
    bound = -oracle_lib.moc(a[0], a[1], a[2], np.abs(a[3]))
    assert np.all(np.abs(d[:, :, r].cpu().numpy() - ref) <= 1e-13 * bound + 1e-300)
    assert torch.equal(plan.compute(zv[1:]), d[1:])                       # record 2 alone: an offset of 1.05e10 elements
    moc_desc = plan.describe()
    del plan, d
    torch.cuda.empty_cache()

    # ---- cdfcurl on (zu, zv[0]) at every level: pointwise, sampled rows bit for bit (the row above a sampled one is needed)
    zu = field(-0.5, 0.5, "u")
    e1u, e2v, e1f, e2f = (2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen) for _ in range(4))
    cp = gpu.CurlPlan(npi, npj)
    cp.set_mesh(e1u, e2v, e1f, e2f)
    curl = cp.compute(zu, zv[0], out_r8=False)
    lev = [0, 37, npk - 1]
    for j in (0, npj // 3, npj - 2):
        sl = slice(j, j + 2)
        cs = gpu.CurlPlan(npi, 2)
        cs.set_mesh(e1u[sl].contiguous(), e2v[sl].contiguous(), e1f[sl].contiguous(), e2f[sl].contiguous())
        small = cs.compute(zu[lev, sl].contiguous(), zv[0][lev, sl].contiguous(), out_r8=False)
        assert torch.equal(small[:, 0], curl[lev, j])                      # row j of a 2-row domain = row j of the big one
        m = [x[sl].cpu().numpy() for x in (e1u, e2v, e1f, e2f)]
        for n, k in enumerate(lev):
            ref = oracle_tools.curl(zu[k, sl].cpu().numpy(), zv[0, k, sl].cpu().numpy(), *m)
            assert np.array_equal(small[n, 0].cpu().numpy(), ref[0].astype(f32))
    assert not torch.any(curl[:, -1] != 0)                                  # the last row stays 0 (:262-270)
    del curl, cp
    torch.cuda.empty_cache()

    # ---- cdfpsi, U path on the whole domain: sampled columns against the oracle (the scan runs along j in a column)
    e2u = 2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen)
    e3u = field(0.2, 1.0, "u", scale=e3p)
    one = gpu.PsiPlan(npi, npj, npk)
    cols = np.sort(np.random.default_rng(3).choice(npi, 24, replace=False))
    c = torch.from_numpy(cols).to(dev)
    p4, d8 = one.compute(zu=zu, e2u=e2u, e3u=e3u, iiref=int(cols[-1]) + 1, ijref=npj, host_out=False)
    r4, r8 = oracle_tools.psi(npk, zu=zu[:, :, c].cpu().numpy(), e2u=e2u[:, c].cpu().numpy(), e3u=e3u[:, :, c].cpu().numpy())
    assert np.array_equal(d8[:, c].cpu().numpy(), r8) and np.array_equal(p4[:, c].cpu().numpy(), r4)
    del one, p4, d8, e3u, zu, zv
    torch.cuda.empty_cache()

    # ---- cdfheatc: the level sums of the whole domain = those of its two halves added (r8 sums: regrouped, 1e-12)
    temp = field(-2.0, 30.0, "t")
    e3t = field(0.2, 1.0, "t", scale=e3p)
    tmask = torch.empty((npk, npj, npi), dtype=torch.uint8, device=dev)                # 0/1 bytes (10.5 GB)
    for k in range(npk):
        tmask[k] = level_mask(k, "t")
    e1t, e2t = (2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen) for _ in range(2))

    def heatc(j0, j1):
        hp = gpu.HeatcPlan(npi, j1 - j0, npk)
        hp.set_sum_mode(False)
        hp.set_mesh(e1t[j0:j1].contiguous(), e2t[j0:j1].contiguous(), tmask[:, j0:j1].contiguous(), e3t=e3t[:, j0:j1].contiguous())
        return np.asarray(hp.compute(temp[:, j0:j1].contiguous()))

    whole = gpu.HeatcPlan(npi, npj, npk)
    whole.set_sum_mode(False)
    whole.set_mesh(e1t, e2t, tmask, e3t=e3t)
    w = np.asarray(whole.compute(temp))
    h = npj // 2
    parts = heatc(0, h) + heatc(h, npj)
    assert np.allclose(w, parts, rtol=1e-12, atol=0) and np.all(w[:-1, 1] > 0) and not w[-1].any()   # level npk is land
    assert "moc_zonal" in moc_desc
