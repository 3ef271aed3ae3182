"""The cdfmoc PROGRAM on the path the bench measures (round-4 verdict, first item).

The reference works record by record (src/cdfmoc.f90:338-388: getvar per level, the zonal loop, the cumulated sum, the write).
The layout that reads every input byte once -- all records of a run streamed past register-resident weights, each (jj,jk)
row's weights and records in one run of the plan's POOL -- used to be reached through cdft_moc_stage only, which no program
called.  Now the host feed writes every slab it reads straight into its pool slots (cdft_moc_stage_open / _dest / _close +
cdft_feed_push_rows: the H2D copy IS the staging) and the program makes ONE cdft_moc_compute_staged launch per batch of
records (a batch = what HBM holds; CDFT_MOC_BATCH sets it).  Checked here: the new C-ABI entries against cdft_moc_stage bit
for bit, and the program -- which layout and how many launches it used, read from its CDFT_PLAN line on stderr -- against
the reference program's own moc.nc (tests/golden/prog_cdfmoc_default.npz) and against the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nemo_files
import oracle_lib
import prog_cases as pc
import synth
from test_cli import read_nc, run, tools  # noqa: F401  (tools is a fixture)
from test_prog_diff import BIN, GOLD, diff_case, explained

PLAN_LINE = re.compile(r"CDFT_PLAN cdfmoc: (.*) zonal_launches=(\d+) records=(\d+)")


def _push_rows(gpu, lib, feed, dst_ptr, pitch, rows, nper):
    """`rows` (n, row_len) r4 through the ring, nper rows per slot, to dst_ptr with `pitch` bytes between row starts"""
    n, w = rows.shape
    for r0 in range(0, n, nper):
        host, nb = C.c_void_p(), C.c_size_t(0)
        gpu.check(lib.cdft_feed_slot(feed, C.byref(host), C.byref(nb)))
        k = min(nper, n - r0)
        np.frombuffer((C.c_char * (4 * w * k)).from_address(host.value), dtype=np.float32)[:] = rows[r0:r0 + k].ravel()
        gpu.check(lib.cdft_feed_push_rows(feed, C.c_void_p(dst_ptr), r0 * pitch, pitch, 4 * w, k))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bounce", "2d"])
def test_feed_push_rows_spreads_a_slot_to_a_pitch(gpu, monkeypatch, mode):
    """cdft_feed_push_rows: dense rows of a slot land pitch bytes apart, what lies between them is left alone; both forms"""
    import torch
    from cdftools_amd import _lib
    lib = _lib.load()
    if mode == "2d":
        monkeypatch.setenv("CDFT_FEED_ROWS", "2d")
    else:
        monkeypatch.delenv("CDFT_FEED_ROWS", raising=False)
    feed, st = C.c_void_p(), C.c_void_p()
    gpu.check(lib.cdft_feed_create(C.byref(feed), 7 * 4 * 333, 3))
    gpu.check(lib.cdft_feed_stream(feed, C.byref(st)))
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((40, 333)).astype(np.float32)
    pitch_el = 3 * 336
    dst = torch.full((40 * pitch_el + 8,), -7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    _push_rows(gpu, lib, feed, dst.data_ptr() + 16, 4 * pitch_el, rows, 7)
    gpu.check(lib.cdft_feed_wait(feed))
    got = dst.cpu().numpy()[4:4 + 40 * pitch_el].reshape(40, pitch_el)
    assert np.array_equal(got[:, :333], rows) and np.all(got[:, 333:] == -7.0)
    with pytest.raises(gpu.CdftError):                       # more than a slot holds; a pitch shorter than a row
        gpu.check(lib.cdft_feed_push_rows(feed, C.c_void_p(dst.data_ptr()), 0, 4 * 336, 4 * 333, 8))
    with pytest.raises(gpu.CdftError):
        gpu.check(lib.cdft_feed_push_rows(feed, C.c_void_p(dst.data_ptr()), 0, 4 * 332, 4 * 333, 2))
    gpu.check(lib.cdft_feed_destroy(feed))


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb,npt", [(24, 18, 7, 5, 2), (182, 29, 9, 5, 3), (65, 5, 4, 1, 2), (4322, 3, 4, 5, 12), (1442, 5, 3, 5, 4)])
def test_records_written_into_the_pool_by_the_feed_equal_staged_records(gpu, npi, npj, npk, nb, npt):
    """stage_open / stage_dest / stage_close with the slabs pushed through the ring (levels 1 .. npk-1 only, as the program
    reads them) == cdft_moc_stage of the whole records, bit for bit; one launch for all records (cdft_moc_stats)"""
    from cdftools_amd import _lib
    lib = _lib.load()
    g = synth.make_grid(npi, npj, npk, seed=npi + npj, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g)
    zv = synth.velocity(g, npt, seed=8)
    ref = oracle_lib.moc(e1v, e3v, ib, zv)
    a_plan = gpu.MocPlan(npi, npj, npk, nb)
    a_plan.set_mesh(e1v, e3v, ib)
    a_plan.stage(zv)
    a = a_plan.compute_staged(np.zeros_like(ref))
    a_plan.close()

    plan = gpu.MocPlan(npi, npj, npk, nb)
    plan.set_mesh(e1v, e3v, ib)
    feed, st = C.c_void_p(), C.c_void_p()
    nlev = 2                                                  # levels per slot
    gpu.check(lib.cdft_feed_create(C.byref(feed), nlev * npj * npi * 4, 2))
    gpu.check(lib.cdft_feed_stream(feed, C.byref(st)))
    with pytest.raises(gpu.CdftError):
        plan.stage_dest(0, 0)                                 # not open
    plan.stage_open(npt, stream=st)
    with pytest.raises(gpu.CdftError):
        plan.stage_dest(npt, 0)
    for t in range(npt):
        for k0 in range(0, npk - 1, nlev):
            k1 = min(k0 + nlev, npk - 1)
            dst, pitch = plan.stage_dest(t, k0)
            assert pitch == 4 * (npt + 1) * ((npi + 3) // 4 * 4) and dst % 16 == 0
            _push_rows(gpu, lib, feed, dst, pitch, zv[t, k0:k1].reshape(-1, npi), (k1 - k0) * npj)
    plan.stage_close()
    n0 = plan.stats()
    b = plan.compute_staged(np.full(ref.shape, np.nan), stream=st)
    assert np.array_equal(a, b)
    assert plan.stats() == (n0[0] + 1, n0[1] + npt) and "layout=pool" in plan.describe()
    # the pool given back: nothing staged any more; staging again works
    plan.release_staged()
    with pytest.raises(gpu.CdftError):
        plan.compute_staged(np.zeros_like(ref))
    plan.stage(zv)
    assert np.array_equal(plan.compute_staged(np.zeros_like(ref)), a)
    gpu.check(lib.cdft_feed_destroy(feed))
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk,nb,npt,nlev", [(24, 18, 7, 5, 2, 2), (182, 29, 9, 5, 3, 3), (65, 5, 4, 1, 2, 1), (4322, 3, 5, 5, 12, 2),
                                                     (1442, 5, 3, 5, 4, 5), (12960, 2, 3, 5, 2, 1)])
def test_the_stream_route_equals_the_resident_pool(gpu, npi, npj, npk, nb, npt, nlev):
    """cdft_moc_set_basins + cdft_moc_stream_begin / _dest / _vmask_dest / _compute / _end: the domain a window of nlev levels
    at a time (raw e3v, vmask bytes and every record's levels pushed through the ring, the weights formed in the window),
    no 3-D array resident == cdft_moc_set_mesh_vmask + cdft_moc_stage + cdft_moc_compute_staged, bit for bit (the same rows
    in the same layout); one launch per window, every launch covering all npt records; two runs on one plan alike"""
    from cdftools_amd import _lib
    lib = _lib.load()
    g = synth.make_grid(npi, npj, npk, seed=2 * npi + npj, nbasins=nb)
    e1v, e3v, ib = synth.moc_inputs(g)
    vm = (e3v != 0).astype(np.int8)
    raw = (e3v + (1 - vm) * np.float32(7.5)).astype(np.float32)          # the file's e3v: not masked
    zv = synth.velocity(g, npt, seed=9)
    ref = oracle_lib.moc(e1v, e3v, ib, zv)
    a_plan = gpu.MocPlan(npi, npj, npk, nb)
    a_plan.set_mesh(e1v, raw, ib, vmask=vm)
    a_plan.stage(zv)
    a = a_plan.compute_staged(np.zeros_like(ref))
    a_plan.close()
    assert np.all(np.abs(a - ref) <= 1e-13 * -oracle_lib.moc(e1v, e3v, ib, np.abs(zv)) + 1e-300)

    plan = gpu.MocPlan(npi, npj, npk, nb)
    with pytest.raises(gpu.CdftError):
        plan.stream_begin(npt, nlev)                          # no basins yet
    plan.set_basins(e1v, ib)
    feed, st = C.c_void_p(), C.c_void_p()
    gpu.check(lib.cdft_feed_create(C.byref(feed), nlev * npj * npi * 4, 3))
    gpu.check(lib.cdft_feed_stream(feed, C.byref(st)))
    for run_ in range(2):
        plan.stream_begin(npt, nlev, stream=st)
        n0 = plan.stats()
        nwin = 0
        for k0 in range(0, npk - 1, nlev):
            k1 = min(k0 + nlev, npk - 1)
            host, nbytes = C.c_void_p(), C.c_size_t(0)
            gpu.check(lib.cdft_feed_slot(feed, C.byref(host), C.byref(nbytes)))
            n = (k1 - k0) * npj * npi
            np.frombuffer((C.c_char * n).from_address(host.value), dtype=np.int8)[:] = vm[k0:k1].ravel()
            gpu.check(lib.cdft_feed_push(feed, C.c_void_p(plan.stream_vmask_dest(0)), 0, n))
            for slot in range(npt + 1):
                src = raw[k0:k1] if slot == 0 else zv[slot - 1, k0:k1]
                dst, pitch = plan.stream_dest(slot, 0)
                assert pitch == 4 * (npt + 1) * ((npi + 3) // 4 * 4) and dst % 16 == 0
                _push_rows(gpu, lib, feed, dst, pitch, src.reshape(-1, npi), (k1 - k0) * npj)
            plan.stream_compute(k0, k1 - k0, stream=st)
            nwin += 1
        with pytest.raises(gpu.CdftError):
            plan.stream_compute(npk - 1, 1, stream=st)        # level npk is never read (src/cdfmoc.f90:355)
        b = plan.stream_end(np.full(ref.shape, np.nan), stream=st)
        assert np.array_equal(a, b)
        assert plan.stats() == (n0[0] + nwin, n0[1] + nwin * npt) and "layout=pool" in plan.describe()
    with pytest.raises(gpu.CdftError):
        plan.stream_dest(0, 0)                                # ended: begin again first
    gpu.check(lib.cdft_feed_destroy(feed))
    plan.close()


@pytest.mark.gpu
def test_a_rejected_basin_mask_withdraws_the_mesh(gpu):
    """cdft_moc_set_basins with values other than 0/1 overwrites the plan's bits and classes before it refuses the mask: a mesh set
    before must not let cdft_moc_compute run on them (round-5 advisor) -- the compute is refused until a mesh or mask is accepted"""
    g = synth.make_grid(24, 18, 7, seed=5, nbasins=5)
    e1v, e3v, ib = synth.moc_inputs(g)
    zv = synth.velocity(g, 2, seed=3)
    ref = oracle_lib.moc(e1v, e3v, ib, zv)
    plan = gpu.MocPlan(24, 18, 7, 5)
    plan.set_mesh(e1v, e3v, ib)
    a = plan.compute(zv, np.zeros_like(ref))
    bad = ib.copy()
    bad.flat[7] = 2
    with pytest.raises(gpu.CdftError, match="other than 0/1"):
        plan.set_basins(e1v, bad)
    with pytest.raises(gpu.CdftError, match="has not succeeded"):
        plan.compute(zv, np.zeros_like(ref))
    plan.set_mesh(e1v, e3v, ib)
    assert np.array_equal(plan.compute(zv, np.zeros_like(ref)), a)
    plan.close()


def _plan_line(err):
    m = PLAN_LINE.search(err)
    assert m, err[-600:]
    return m.group(1), int(m.group(2)), int(m.group(3))


@pytest.mark.gpu
def test_the_program_runs_one_launch_on_the_pool_and_writes_the_reference_programs_file(tools, tmp_path):  # noqa: F811
    """cdfmoc with the feed on the files of the program differential: layout=pool, ONE zonal launch for the run's records, and
    moc.nc held to what the reference's own cdfmoc wrote on the same files (tests/golden/prog_cdfmoc_default.npz) by the
    differential's own comparison -- schema, attributes, stdout, data within 1 ulp(r4)"""
    d = str(tmp_path)
    g_, rec_ = nemo_files.make_case(d)
    case = next(c for c in pc.CASES if c[0] == "cdfmoc_default")
    ref, ref_arr = pc.load_golden(os.path.join(GOLD, "prog_cdfmoc_default.npz"))
    other_shape = bool(os.environ.get("CDFT_TEST_SHAPE"))     # (the golden is the differential's grid: a shape sweep checks the launches only)
    got, got_arr = pc.run_case(BIN, case, d, env={"CDFT_FEED": "1"})
    bad = [x for x in diff_case(case, ref, ref_arr, got, got_arr) if not explained(case[0], x[0], x[1])]
    assert other_shape or not bad, bad[:10]
    npt, npk = rec_["v"].shape[0], rec_["v"].shape[1]
    # the default, the stream route: windows of levels, EVERY launch covers all the run's records, in the pool layout
    rc, out, err = run("cdfmoc", case[2], d, env={"CDFT_FEED": "1", "CDFT_TIMING": "1"})
    assert rc == 0, out + err
    desc, launches, records = _plan_line(err)
    assert "layout=pool" in desc and records == launches * npt and 1 <= launches <= npk - 1, err[-400:]
    # CDFT_MOC_ROUTE=pool: all records resident in the plan's pool, ONE launch -- bench.py's very call
    got, got_arr = pc.run_case(BIN, case, d, env={"CDFT_FEED": "1", "CDFT_MOC_ROUTE": "pool"})
    bad = [x for x in diff_case(case, ref, ref_arr, got, got_arr) if not explained(case[0], x[0], x[1])]
    assert other_shape or not bad, bad[:10]
    rc, out, err = run("cdfmoc", case[2], d, env={"CDFT_FEED": "1", "CDFT_TIMING": "1", "CDFT_MOC_ROUTE": "pool"})
    desc, launches, records = _plan_line(err)
    assert rc == 0 and "layout=pool" in desc and (launches, records) == (1, npt), err[-400:]
    # record by record: as many launches as records, on the dense record
    rc, out, err = run("cdfmoc", case[2], d, env={"CDFT_FEED": "1", "CDFT_TIMING": "1", "CDFT_MOC_ROUTE": "record"})
    desc, launches, records = _plan_line(err)
    assert rc == 0 and "layout=pool" not in desc and launches == records == npt


@pytest.mark.gpu
def test_the_program_with_vvl_streams_a_window_per_record(tools, tmp_path):  # noqa: F811
    """cdfmoc -vvl with the feed: the metric changes with every record, so a window holds the e3v and the velocity of ONE record
    (a launch per record and window, layout=pool, no 3-D array resident); moc.nc held to the reference program's own -vvl run
    (tests/golden/prog_cdfmoc_vvl.npz) by the differential's comparison, and equal to the record-by-record route within 1 ulp(r4)"""
    d = str(tmp_path)
    g_, rec_ = nemo_files.make_case(d)
    pc.extra_files(d, g_, rec_)
    case = next(c for c in pc.CASES if c[0] == "cdfmoc_vvl")
    ref, ref_arr = pc.load_golden(os.path.join(GOLD, "prog_cdfmoc_vvl.npz"))
    got, got_arr = pc.run_case(BIN, case, d, env={"CDFT_FEED": "1"})
    bad = [x for x in diff_case(case, ref, ref_arr, got, got_arr) if not explained(case[0], x[0], x[1])]
    assert not bad, bad[:10]
    npt = rec_["v"].shape[0]
    rc, out, err = run("cdfmoc", case[2], d, env={"CDFT_FEED": "1", "CDFT_TIMING": "1", "CDFT_MOC_WINDOW": "4"})
    desc, launches, records = _plan_line(err)
    assert rc == 0 and "layout=pool" in desc and launches == records == npt * 2, err[-400:]      # 6 levels: windows of 4 + 2, per record
    rc, out, err = run("cdfmoc", case[2], d, env={"CDFT_FEED": "1", "CDFT_TIMING": "1", "CDFT_MOC_ROUTE": "pool"})   # (no pool with -vvl: the stream)
    desc, launches, records = _plan_line(err)
    assert rc == 0 and "layout=pool" in desc and launches == records == npt


@pytest.mark.gpu
@pytest.mark.parametrize("ring", [{}, {"CDFT_FEED_MB": "0.0001", "CDFT_FEED_SLOTS": "2"}, {"CDFT_FEED_ROWS": "2d"}],
                         ids=["default_ring", "one_plane_two_slots", "pitched_copies"])
def test_the_program_in_batches_of_records(tools, tmp_path, ring):  # noqa: F811
    """five records, the three routes (CDFT_MOC_ROUTE), in batches of 2 (CDFT_MOC_BATCH: 2 + 2 + 1 -- the pool / the window is
    laid out anew for the last one), of 5, of 1: the same moc.nc within 1 ulp(r4) of the oracle whichever way, and the same bits
    from every way that reads the pool layout (a row's sums depend on nothing but the row there)"""
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=70, npj=23, npk=9, npt=5)
    e1v, e3v, ib = synth.moc_inputs(g, nbasins=5)
    ref = oracle_lib.moc(e1v, e3v, ib, rec["v"])
    files = {}
    one_plane = "CDFT_FEED_MB" in ring
    nwin = 3 if one_plane else 1                            # windows of three levels (3 + 3 + 2, a ring slot per level) or of all eight
    if one_plane:
        ring = dict(ring, CDFT_MOC_WINDOW="3")
    for route, batch, want in (("pool", "2", (3, 5)), ("pool", "5", (1, 5)), ("pool", "", (1, 5)), ("record", "", (5, 5)),
                               ("stream", "", (nwin, 5 * nwin)), ("stream", "2", (3 * nwin, 5 * nwin)), ("stream", "1", (5 * nwin, 5 * nwin))):
        env = dict(ring, CDFT_FEED="1", CDFT_TIMING="1", CDFT_MOC_ROUTE=route)
        if batch:
            env["CDFT_MOC_BATCH"] = batch
        key = route + batch
        rc, out, err = run("cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-o", f"moc{key}.nc"], d, env=env)
        assert rc == 0, out + err
        desc, launches, records = _plan_line(err)
        assert (launches, records) == want and ("layout=pool" in desc) == (route != "record"), (key, err[-300:])
        var, _, _ = read_nc(os.path.join(d, f"moc{key}.nc"))
        for b, nm in enumerate(["zomsfglo", "zomsfatl", "zomsfinp", "zomsfind", "zomsfpac"]):
            w = ref[..., b].astype(np.float32)
            assert np.all(np.abs(var[nm][:, :, :, 0] - w) <= np.spacing(np.abs(w))), (key, nm)
        files[key] = {k: v for k, v in var.items()}
    for key in files:                                       # every route that reads the pool layout: the same bits
        if key != "record":
            for k in files["pool"]:
                assert np.array_equal(files[key][k], files["pool"][k]), (key, k)
