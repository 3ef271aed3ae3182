"""Row-block sharded plans behind the C ABI (include/cdftools_hip.h "sharded plans", cdftools_amd/csrc/shard.hip): one
process, one single-device plan per row block, the caller's full-domain arrays scattered and the results landed in
place.  On this 1-GPU box every block is mapped to device 0 (devices=[0,0,..] == CDFT_GPU_MAP=0,0,..): the code path --
threads, streams, strided copies, the halo rows, the carry of cdfpsi handed from block to block -- is the one a
multi-GPU node runs, only the peer copies stay inside one HBM.

Stated results: cdfmoc, cdfmhst, cdfvT, cdfpsi BIT-IDENTICAL to the one-device plan (rows are independent; the psi scan
is chained in order); cdfheatc within 1e-13 * sum|terms| (block sums added in block order)."""
import numpy as np
import pytest

import lift_cases as lc
import oracle_tools as ot
import synth

f32 = np.float32
BLOCKS = [[0, 0], [0, 0, 0], [0] * 5]


def test_shard_from_env_parsing(monkeypatch):
    from cdftools_amd import hip
    monkeypatch.delenv("CDFT_GPUS", raising=False)
    monkeypatch.delenv("CDFT_GPU_MAP", raising=False)
    assert hip.shard_from_env() == [-1]
    monkeypatch.setenv("CDFT_GPUS", "1")
    for bad in ("x", "0", "-2"):
        monkeypatch.setenv("CDFT_GPUS", bad)
        with pytest.raises(hip.CdftError):
            hip.shard_from_env()
    monkeypatch.setenv("CDFT_GPU_MAP", "0,,a")
    with pytest.raises(hip.CdftError):
        hip.shard_from_env()
    # more devices than the box has (none here, one on the GPU box) is refused with a message, not a crash
    monkeypatch.setenv("CDFT_GPU_MAP", "0,63")
    with pytest.raises(hip.CdftError) as e:
        hip.shard_from_env()
    assert "device 63" in str(e.value) or "device 0" in str(e.value)


@pytest.mark.gpu
def test_shard_from_env_on_the_gpu_box(gpu, monkeypatch):
    monkeypatch.setenv("CDFT_GPU_MAP", "0, 0,0")
    assert gpu.shard_from_env() == [0, 0, 0]
    monkeypatch.delenv("CDFT_GPU_MAP")
    monkeypatch.setenv("CDFT_GPUS", "1")
    assert gpu.shard_from_env() == [0]


@pytest.mark.gpu
@pytest.mark.parametrize("devices", BLOCKS)
@pytest.mark.parametrize("npi,npj,npk,nb", [(37, 29, 9, 5), (182, 149, 31, 5), (130, 11, 4, 1)])
def test_sharded_moc_is_bit_identical_to_one_device(gpu, devices, npi, npj, npk, nb):
    import torch
    g = synth.make_grid(npi, npj, npk, seed=npi, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g, nb)
    zv = synth.velocity(g, 3, seed=5)
    one = gpu.MocPlan(npi, npj, npk, nb)
    one.set_mesh(e1v, e3v, ib)
    ref = one.compute(zv)
    sh = gpu.MocPlan(npi, npj, npk, nb, devices=devices)
    sh.set_mesh(e1v, e3v, ib)
    assert np.array_equal(sh.compute(zv), ref)
    # raw metric + byte mask (cdft_moc_set_mesh_vmask), arrays resident in HBM, result in HBM
    sh.set_mesh(torch.from_numpy(e1v).cuda(), torch.from_numpy(g["e3v_0"]).cuda(), torch.from_numpy(ib).cuda(),
                vmask=torch.from_numpy(g["vmask"].astype(np.int8)).cuda())
    got = sh.compute(torch.from_numpy(zv).cuda())
    assert np.array_equal(got.cpu().numpy(), ref)
    # a record stride larger than a record
    pad = np.zeros((3, npk * npj * npi + 7), f32)
    pad[:, : npk * npj * npi] = zv.reshape(3, -1)
    assert np.array_equal(sh.compute(pad, rec_stride=pad.shape[1], npt=3), ref)
    with pytest.raises(gpu.CdftError):
        sh.decomp_set_mesh(e3v, g["e1u"], g["e1u"], g["gdept_1d"], g["tmask"], g["umask"])
    assert "moc_zonal" in sh.describe()


@pytest.mark.gpu
def test_sharded_moc_rejects_bad_masks_and_too_many_blocks(gpu):
    g = synth.make_grid(24, 6, 4, seed=1)
    e1v, e3v, ib = synth.moc_inputs(g)
    with pytest.raises(gpu.CdftError):
        gpu.MocPlan(24, 6, 4, 5, devices=[0] * 7)                 # more blocks than rows
    with pytest.raises(gpu.CdftError):
        gpu.MocPlan(24, 6, 4, 5, devices=[0, 99])
    sh = gpu.MocPlan(24, 6, 4, 5, devices=[0, 0])
    bad = ib.copy()
    bad[5, 3, 1] = 2                                              # in the second block
    with pytest.raises(gpu.CdftError) as e:
        sh.set_mesh(e1v, e3v, bad)
    assert "row block 1" in str(e.value)
    with pytest.raises(gpu.CdftError):
        sh.compute(synth.velocity(g, 1))                         # no mesh


@pytest.mark.gpu
@pytest.mark.parametrize("devices", BLOCKS)
def test_sharded_heatc_matches_one_device(gpu, devices):
    for name in ("default", "full", "mxl_in", "mxl_out"):
        g, temp, full, mxlopt, mxl = lc.heatc_case(name)
        npk, npj, npi = g["tmask"].shape
        mesh = dict(e31d=g["e3t_1d"]) if full else dict(e3t=g["e3t_0"])
        one = gpu.HeatcPlan(npi, npj, npk)
        one.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), **mesh)
        sh = gpu.HeatcPlan(npi, npj, npk, devices=devices)
        sh.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), **mesh)
        for t in range(temp.shape[0]):
            kw = dict(rmxldep=None if mxl is None else mxl[t], gdepw=g["gdepw_1d"], mxloption=mxlopt)
            ref, got = one.compute(temp[t], **kw), sh.compute(temp[t], **kw)
            bound = ot.heatc(g["e1t"], g["e2t"], g["tmask"], np.abs(temp[t]), mxl=kw["rmxldep"], gdepw=g["gdepw_1d"], mxlopt=mxlopt, **mesh)
            assert np.all(np.abs(got - ref) <= 1e-13 * bound + 1e-300), name
    # an r4 mask that is not 0/1 (-M)
    g, temp, *_ = lc.heatc_case("default")
    npk, npj, npi = g["tmask"].shape
    m4 = (g["tmask"] * f32(0.5)).astype(f32)
    one, sh = gpu.HeatcPlan(npi, npj, npk), gpu.HeatcPlan(npi, npj, npk, devices=devices)
    for p in (one, sh):
        p.set_mesh(g["e1t"], g["e2t"], m4, e3t=g["e3t_0"])
    ref, got = one.compute(temp[0]), sh.compute(temp[0])
    assert np.allclose(got, ref, rtol=1e-13, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", BLOCKS)
def test_sharded_mhst_is_bit_identical_to_one_device(gpu, devices):
    for name in lc.MHST_CASES:
        g, nb, kw = lc.mhst_case(name)
        npk, npj, npi = g["tmask"].shape
        m = np.zeros((4 if nb == 5 else 1, npj, npi), f32)
        m[0] = g["vmask"][0]
        if nb == 5:
            m[1], m[2], m[3] = g["tmaskatl"], g["tmaskpac"], g["tmaskind"]
        mesh = dict(e31d=g["e3t_1d"]) if kw["full"] else dict(e3v=g["e3v_0"])
        one, sh = gpu.MhstPlan(npi, npj, npk, m.shape[0]), gpu.MhstPlan(npi, npj, npk, m.shape[0], devices=devices)
        for p in (one, sh):
            p.set_mesh(g["e1v"], m, **mesh)
        fields = {k: kw[k] for k in ("vt", "vs", "zv", "zt", "zs") if k in kw}
        for t in range(next(iter(fields.values())).shape[0]):
            rec = {k: v[t] for k, v in fields.items()}
            rh, rs = one.compute(zdim=kw["zdim"], **rec)
            gh, gs = sh.compute(zdim=kw["zdim"], **rec)
            assert np.array_equal(gh, rh) and np.array_equal(gs, rs), name


@pytest.mark.gpu
@pytest.mark.parametrize("devices", BLOCKS)
def test_sharded_psi_is_bit_identical_to_one_device(gpu, devices):
    import torch
    for name in ("u", "v", "mean_mask_ref", "u_full"):
        g, kw = lc.psi_case(name)
        npk, npj, npi = g["tmask"].shape
        one, sh = gpu.PsiPlan(npi, npj, npk), gpu.PsiPlan(npi, npj, npk, devices=devices)
        fm = g["fmask"][0].astype(f32) if kw.get("mask") else None
        for t in range(2):
            a = {}
            if kw["zu"] is not None:
                a.update(zu=kw["zu"][t], e2u=g["e2u"], **({} if kw.get("full") else dict(e3u=g["e3u_0"])))
            if kw["zv"] is not None:
                a.update(zv=kw["zv"][t], e1v=g["e1v"], **({} if kw.get("full") else dict(e3v=g["e3v_0"])))
            if kw.get("full"):
                a["e31d"] = g["e3t_1d"]
            r4, r8 = one.compute(iiref=kw.get("iiref"), ijref=kw.get("ijref"), fmask=fm, **a)
            g4, g8 = sh.compute(iiref=kw.get("iiref"), ijref=kw.get("ijref"), fmask=fm, **a)
            assert np.array_equal(g4, r4) and np.array_equal(g8, r8), name
            # the reference point in every block in turn
            for ij in (1, npj // 2, npj):
                r4, r8 = one.compute(iiref=3, ijref=ij, **a)
                g4, g8 = sh.compute(iiref=3, ijref=ij, **a)
                assert np.array_equal(g4, r4) and np.array_equal(g8, r8), (name, ij)
            # arrays and results resident in HBM
            d = {k: torch.from_numpy(np.ascontiguousarray(v, f32)).cuda() for k, v in a.items()}
            g4, g8 = sh.compute(host_out=False, **d)
            r4, r8 = one.compute(**a)
            assert np.array_equal(g4.cpu().numpy(), r4) and np.array_equal(g8.cpu().numpy(), r8)
    with pytest.raises(gpu.CdftError):
        sh.scan_open()                                            # -open: one device only


@pytest.mark.gpu
@pytest.mark.parametrize("devices", BLOCKS)
def test_sharded_vt_is_bit_identical_to_one_device(gpu, devices):
    import torch
    for name in ("tags2x3", "vvl"):
        u, v, t, s, e3u, e3v = lc.vt_case(name)
        fl = lambda x: None if x is None else np.ascontiguousarray(x.reshape((-1,) + x.shape[2:]))     # noqa: E731
        a = [fl(x) for x in (u, v, t, s)]
        for out_r4 in (False, True):
            ref = gpu.vt_compute(*a, e3u=fl(e3u), e3v=fl(e3v), out_r4=out_r4)
            got = gpu.vt_compute(*a, e3u=fl(e3u), e3v=fl(e3v), out_r4=out_r4, devices=devices)
            assert set(ref) == set(got)
            for k in ref:
                assert np.array_equal(got[k], ref[k], equal_nan=True), (name, k)
        d = [torch.from_numpy(x).cuda() for x in a]
        got = gpu.vt_compute(*d, devices=devices)
        ref = gpu.vt_compute(*a)
        for k in ("vt", "vs", "ut", "us"):
            assert np.array_equal(got[k].cpu().numpy(), ref[k])


# ------------------------------------------------------------------------------------------ staged records, per-block probes
@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0], [0, 0, 0]])
def test_staged_records_and_per_block_probes(gpu, devices):
    """cdft_moc_stage + cdft_moc_compute_staged (what bench.py times): the records are handed over once into the plan's pool
    layout (every row's weights and records in one run), every later call gives the same bits -- for any number of row
    blocks, since a row's sums depend on nothing but the row in that layout -- within the stated tolerance of the oracle;
    the launch probe of a sharded plan speaks per block."""
    import torch
    npi, npj, npk, nb = 182, 149, 31, 5
    g = synth.make_grid(npi, npj, npk, seed=3, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g, nb)
    zv = synth.velocity(g, 3, seed=5)
    p = gpu.MocPlan(npi, npj, npk, nb, devices=devices)
    p.set_mesh(e1v, e3v, ib)
    import oracle_lib
    dense = p.compute(zv)
    with pytest.raises(gpu.CdftError):                              # nothing staged
        p.compute_staged(np.empty_like(dense))
    one = gpu.MocPlan(npi, npj, npk, nb)
    one.set_mesh(e1v, e3v, ib)
    one.stage(zv)
    ref = one.compute_staged(np.zeros_like(dense))                   # one device, the pool layout
    want = oracle_lib.moc(e1v, e3v, ib, zv)
    lim = 1e-13 * -oracle_lib.moc(e1v, e3v, ib, np.abs(zv)) + 1e-300
    assert np.all(np.abs(ref - want) <= lim) and np.all(np.abs(dense - want) <= lim)
    for src in (zv, torch.from_numpy(zv).cuda()):
        p.stage(src)
        p.profile_enable(True)
        for _ in range(3):
            out = p.compute_staged(np.zeros_like(ref))
            assert np.array_equal(out, ref)
        dout = p.compute_staged(torch.zeros(ref.shape, dtype=torch.float64, device="cuda"))
        assert np.array_equal(dout.cpu().numpy(), ref)
        p.profile_enable(False)
        nblk = 1 if devices is None else len(devices)
        rows = 0
        for b in range(nblk):
            dev, j0, j1 = p.block_info(b)
            assert dev == 0 and j0 == rows and j1 > j0
            rows = j1
        assert rows == npj
        each = p.profile_read_each()                              # the slowest block of every launch (and a reset of every block)
        assert each.size == 4 and np.all(each > 0)
        p.profile_enable(True)
        p.compute_staged(np.zeros_like(ref))
        p.compute_staged(np.zeros_like(ref))
        p.profile_enable(False)
        one = p.profile_read_block(nblk - 1)
        assert one.size == 2 and np.all(one > 0)
        assert p.profile_read_block(nblk - 1).size == 0           # read and reset
        if nblk > 1:
            assert p.profile_read_block(0).size == 2              # (the other blocks keep theirs)
    # lifetime (include/cdftools_hip.h): cdft_moc_compute on other records leaves the staged ones alone, on one device and on
    # row blocks alike; cdft_moc_release_staged gives the pools back
    assert np.array_equal(p.compute(zv), dense) and np.array_equal(p.compute_staged(np.zeros_like(ref)), ref)
    p.release_staged()
    with pytest.raises(gpu.CdftError):
        p.compute_staged(np.zeros_like(ref))
    p.stage(zv)
    assert np.array_equal(p.compute_staged(np.zeros_like(ref)), ref)
    with pytest.raises(gpu.CdftError):
        p.block_info(7)
    with pytest.raises(gpu.CdftError):
        p.profile_read_block(-1)


# ------------------------------------------------------------------------------------------ RCCL inside the library
@pytest.mark.gpu
def test_rccl_collectives_inside_the_library_on_one_device(gpu):
    """CDFT_COLLECTIVE_RCCL on the one GPU there is: librccl opened at run time, ncclCommInitAll over one device, the
    all-gather / broadcast entry points called for real (a communicator of one rank) -- the results are the bits of the
    default transport.  Two blocks on one device are refused with a message (RCCL takes one rank per device); the
    multi-device case is what bench.py's in-process leg reports from the driver's multi-GPU runs ("collectives")."""
    npi, npj, npk = 97, 53, 11
    g = synth.make_grid(npi, npj, npk, seed=9)
    rng = np.random.default_rng(10)
    temp = ((-2 + 30 * rng.random((npk, npj, npi), dtype=f32)) * g["tmask"]).astype(f32)
    u = synth.velocity(g, 1, 11, which="u")[0]
    tm = g["tmask"].astype(np.uint8)

    def run():
        h = gpu.HeatcPlan(npi, npj, npk, devices=[0])
        h.set_sum_mode(False)
        h.set_mesh(g["e1t"], g["e2t"], tm, e3t=g["e3t_0"])
        p = gpu.PsiPlan(npi, npj, npk, devices=[0])
        return h.compute(temp), p.compute(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], iiref=30, ijref=20)
    assert gpu.get_collective() == gpu.COLLECTIVE_PEER
    hs, (p4, p8) = run()
    gpu.set_collective(gpu.COLLECTIVE_RCCL)
    try:
        assert gpu.get_collective() == gpu.COLLECTIVE_RCCL
        hr, (r4, r8) = run()
        with pytest.raises(gpu.CdftError) as e:
            gpu.HeatcPlan(npi, npj, npk, devices=[0, 0])
        assert "one rank per device" in str(e.value)
    finally:
        gpu.set_collective(gpu.COLLECTIVE_PEER)
    assert np.array_equal(hs, hr) and np.array_equal(p4, r4) and np.array_equal(p8, r8)
    one = gpu.PsiPlan(npi, npj, npk).compute(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], iiref=30, ijref=20)
    assert np.array_equal(one[0], r4) and np.array_equal(one[1], r8)
    with pytest.raises(gpu.CdftError):
        gpu.set_collective(7)


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["enqueue", "finish"])
def test_a_failed_collective_aborts_the_communicators_and_says_so(gpu, monkeypatch, where):
    """a block that fails at its RCCL call or behind it (CDFT_SHARD_TEST_FAIL, a test hook): the call returns that block's error
    (it does not hang), the communicators are aborted once every block is past its call, the plan then says they are gone
    instead of handing freed handles to RCCL, and a new plan works"""
    npi, npj, npk = 61, 33, 7
    g = synth.make_grid(npi, npj, npk, seed=19)
    rng = np.random.default_rng(20)
    temp = ((-2 + 30 * rng.random((npk, npj, npi), dtype=f32)) * g["tmask"]).astype(f32)
    tm = g["tmask"].astype(np.uint8)
    gpu.set_collective(gpu.COLLECTIVE_RCCL)
    try:
        h = gpu.HeatcPlan(npi, npj, npk, devices=[0])
        h.set_sum_mode(False)
        h.set_mesh(g["e1t"], g["e2t"], tm, e3t=g["e3t_0"])
        good = h.compute(temp)
        monkeypatch.setenv("CDFT_SHARD_TEST_FAIL", where)
        with pytest.raises(gpu.CdftError) as e:
            h.compute(temp)
        assert "CDFT_SHARD_TEST_FAIL" in str(e.value)
        monkeypatch.delenv("CDFT_SHARD_TEST_FAIL")
        with pytest.raises(gpu.CdftError) as e:
            h.compute(temp)
        assert "aborted by an earlier failure" in str(e.value)
        h2 = gpu.HeatcPlan(npi, npj, npk, devices=[0])
        h2.set_sum_mode(False)
        h2.set_mesh(g["e1t"], g["e2t"], tm, e3t=g["e3t_0"])
        assert np.array_equal(h2.compute(temp), good)
    finally:
        gpu.set_collective(gpu.COLLECTIVE_PEER)


@pytest.mark.gpu
def test_pinned_output_is_asynchronous_and_equal(gpu):
    """CDFT_MEM_PINNED: the result of cdft_moc_compute lands in page-locked memory without the call synchronising; after
    cdft_stream_sync it holds the bits of the synchronous call."""
    import ctypes as C
    from cdftools_amd import _lib
    npi, npj, npk, nb = 182, 149, 31, 5
    g = synth.make_grid(npi, npj, npk, seed=3, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g, nb)
    zv = synth.velocity(g, 2, seed=5)
    p = gpu.MocPlan(npi, npj, npk, nb)
    p.set_mesh(e1v, e3v, ib)
    ref = p.compute(zv)
    zin = gpu.PinnedArray(zv.shape, np.float32)
    zin.a[...] = zv
    out = gpu.PinnedArray(ref.shape, np.float64)
    out.a[...] = -1.0
    lib = _lib.load()
    st = gpu.current_stream_ptr()
    gpu.check(lib.cdft_moc_compute(p._h, C.c_void_p(zin.a.ctypes.data), npi * npj * npk, 2, gpu.MEM_HOST, C.c_void_p(out.a.ctypes.data),
                                   _lib.MEM_PINNED, st))
    gpu.check(lib.cdft_stream_sync(st))
    assert np.array_equal(out.a, ref)
    zin.close()
    out.close()
