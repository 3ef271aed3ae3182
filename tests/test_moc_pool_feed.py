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
    got, got_arr = pc.run_case(BIN, case, d, env={"CDFT_FEED": "1"})
    bad = [x for x in diff_case(case, ref, ref_arr, got, got_arr) if not explained(case[0], x[0], x[1])]
    assert not bad, bad[:10]
    rc, out, err = run("cdfmoc", case[2], d, env={"CDFT_FEED": "1", "CDFT_TIMING": "1"})
    assert rc == 0, out + err
    desc, launches, records = _plan_line(err)
    assert "layout=pool" in desc and (launches, records) == (1, rec_["v"].shape[0]), err[-400:]
    # record by record (CDFT_MOC_BATCH=1): as many launches as records, on the dense record
    rc, out, err = run("cdfmoc", case[2], d, env={"CDFT_FEED": "1", "CDFT_TIMING": "1", "CDFT_MOC_BATCH": "1"})
    desc, launches, records = _plan_line(err)
    assert rc == 0 and "layout=pool" not in desc and launches == records == rec_["v"].shape[0]


@pytest.mark.gpu
@pytest.mark.parametrize("ring", [{}, {"CDFT_FEED_MB": "0.0001", "CDFT_FEED_SLOTS": "2"}, {"CDFT_FEED_ROWS": "2d"}],
                         ids=["default_ring", "one_plane_two_slots", "pitched_copies"])
def test_the_program_in_batches_of_records(tools, tmp_path, ring):  # noqa: F811
    """five records in batches of 2 (CDFT_MOC_BATCH: 2 + 2 + 1 -- the pool is laid out anew for the last one), of 5 and record
    by record: the same moc.nc within 1 ulp(r4) of the oracle whichever way, the batched ways byte-identical to each other"""
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=70, npj=23, npk=9, npt=5)
    e1v, e3v, ib = synth.moc_inputs(g, nbasins=5)
    ref = oracle_lib.moc(e1v, e3v, ib, rec["v"])
    files = {}
    for batch, want in (("2", (3, 5)), ("5", (1, 5)), ("1", (5, 5)), ("", (1, 5))):
        env = dict(ring, CDFT_FEED="1", CDFT_TIMING="1")
        if batch:
            env["CDFT_MOC_BATCH"] = batch
        rc, out, err = run("cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-o", f"moc{batch}.nc"], d, env=env)
        assert rc == 0, out + err
        desc, launches, records = _plan_line(err)
        assert (launches, records) == want and ("layout=pool" in desc) == (batch != "1"), (batch, err[-300:])
        var, _, _ = read_nc(os.path.join(d, f"moc{batch}.nc"))
        for b, nm in enumerate(["zomsfglo", "zomsfatl", "zomsfinp", "zomsfind", "zomsfpac"]):
            w = ref[..., b].astype(np.float32)
            assert np.all(np.abs(var[nm][:, :, :, 0] - w) <= np.spacing(np.abs(w))), (batch, nm)
        files[batch] = {k: v for k, v in var.items()}
    for k in files["2"]:
        assert np.array_equal(files["2"][k], files["5"][k]) and np.array_equal(files["5"][k], files[""][k]), k
