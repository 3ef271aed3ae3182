"""The host feed (module cdffeed, cdft_feed_* of the C ABI): the programs that have it -- cdfmoc, cdfheatc, cdfpsi, cdfvT,
cdfmhst, cdfmocsig, cdfsig0, cdfsigi, cdfcurl, cdfmean, cdfsum, cdfzonalsum, cdfzonalmean, cdfvhst, cdfvtrp, cdfvint, cdfvertmean -- send their 3-D fields from the file to HBM slab by slab through a page-locked ring when the fields are large
(from 1 GiB on) and as whole records in pageable memory otherwise.  The route must not show: for every option that
touches it, CDFT_FEED=1 and CDFT_FEED=0 give the same stdout and byte-identical files (the pageable route is the one the
other program tests hold against the oracle).  A ring with one plane per slot and two slots (CDFT_FEED_MB / _SLOTS) makes
every slot wrap around many times per field."""
import os

import pytest

import lift_cases as lc
import nemo_files
from test_cli import run, tools  # noqa: F401  (tools is a fixture)


@pytest.fixture(scope="module")
def case(tools, tmp_path_factory):  # noqa: F811
    d = str(tmp_path_factory.mktemp("feed"))
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=3)
    e3 = {p: lc.e3_records(g[f"e3{p}_0"], 3, 8100 + i) for i, p in enumerate("tuv")}
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridV.nc"), g, "depthv", {"vomecrty": rec["v"], "e3v": e3["v"]})
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridU.nc"), g, "depthu", {"vozocrtx": rec["u"], "e3u": e3["u"]})
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridT.nc"), g, "deptht", {"votemper": rec["t"], "vosaline": rec["s"], "e3t": e3["t"]},
                          fill=0.0)
    nemo_files.write_data(os.path.join(d, "EIV_y0001_gridV.nc"), g, "depthv", {"vomecrty": rec["v"], "vomeeivv": (rec["v"][::-1] * 0.3).astype("float32")})
    mxl = (5 + 200 * abs(rec["t"][:, 0]) / 30).astype("float32")
    nemo_files.write_data(os.path.join(d, "mxl.nc"), g, "deptht", {"somxl010": mxl})
    nemo_files.write_data(os.path.join(d, "ssh.nc"), g, "deptht", {"sossheig": (rec["t"][:, 0] / 30 - 0.5).astype("float32")})
    return d


V, U, T = "SYN_y0001_gridV.nc", "SYN_y0001_gridU.nc", "SYN_y0001_gridT.nc"
RUNS = [
    # (record by record: by default the feed takes the domain through HBM in windows of levels laid out like the plan's pool --
    # other r8 additions along i than the dense records of the pageable route, within 1 ulp(r4) in the file:
    # tests/test_moc_pool_feed.py)
    ("cdfmoc", ["-v", V], ["moc.nc"], {"CDFT_MOC_ROUTE": "record"}),
    ("cdfmoc", ["-v", "VVL_y0001_gridV.nc", "-vvl"], ["moc.nc"], {"CDFT_MOC_ROUTE": "record"}),
    ("cdfheatc", ["-f", T], ["heatc.nc"]),
    ("cdfheatc", ["-f", T, "-full"], ["heatc.nc"]),
    ("cdfheatc", ["-f", "VVL_y0001_gridT.nc", "-vvl"], ["heatc.nc"]),
    ("cdfheatc", ["-f", T, "-zoom", "20", "30", "2", "9", "2", "4"], ["heatc.nc"]),          # a window over the eastern edge
    ("cdfheatc", ["-f", T, "-mxloption", "1", "-mxlf", "mxl.nc"], ["heatc.nc"]),
    ("cdfmoc", ["-v", V, "-t", T, "-decomp", "-o", "dec.nc"], ["dec.nc"]),                        # V, T and S of a record through the ring
    ("cdfmoc", ["-v", V, "-t", T, "-decomp", "-teos10", "-o", "dec.nc"], ["dec.nc"]),
    ("cdfmoc", ["-v", "VVL_y0001_gridV.nc", "-t", "VVL_y0001_gridT.nc", "-decomp", "-vvl", "-o", "dec.nc"], ["dec.nc"]),   # the mesh of every record fed too
    ("cdfpsi", ["-u", U, "-v", V], ["psi.nc"]),
    ("cdfpsi", ["-u", U, "-v", V, "-V"], ["psi.nc"]),
    ("cdfpsi", ["-u", U, "-v", V, "-mean", "-mask", "-ref", "5", "4"], ["psi.nc"]),
    ("cdfpsi", ["-u", U, "-v", V, "-full"], ["psi.nc"]),
    ("cdfpsi", ["-u", U, "-v", V, "-open"], ["psi.nc"]),
    ("cdfpsi", ["-u", U, "-v", V, "-mean", "--ssh-file", "ssh.nc"], ["psi.nc"]),
    ("cdfpsi", ["-u", U, "-v", V, "-mean", "--ssh-file", "nothere.nc"], None),                   # both routes stop alike
    ("cdfpsi", ["-u", "VVL_y0001_gridU.nc", "-v", "VVL_y0001_gridV.nc", "-vvl", "-mean"], ["psi.nc"]),
    ("cdfvT", ["-c", "SYN", "-l", "y0001"], ["vt.nc"]),
    ("cdfvT", ["-c", "VVL", "-l", "y0001", "-vvl"], ["vt.nc"]),
    ("cdfmhst", ["-vt", "vt_keep.nc", "-MST"], ["mhst.nc", "zonal_heat_trp.dat", "zonal_salt_trp.dat"]),
    ("cdfmhst", ["-vt", "vt_keep.nc", "-MST", "-Zdim", "-full"], ["mhst.nc"]),
    ("cdfmhst", ["-v", V, "-t", T, "-MST"], ["mhst.nc"]),
    ("cdfmhst", ["-v", "VVL_y0001_gridV.nc", "-t", "VVL_y0001_gridT.nc", "-vvl"], ["mhst.nc"]),
    ("cdfcurl", ["-u", U, "vozocrtx", "-v", V, "vomecrty", "-l", "1-4"], ["curl.nc"]),            # a level at a time, out through page-locked arrays
    ("cdfcurl", ["-u", U, "vozocrtx", "-v", V, "vomecrty", "-l", "2,5", "-T", "-8"], ["curl.nc"]),
    ("cdfcurl", ["-u", U, "vozocrtx", "-v", V, "vomecrty", "-l", "3", "-overf"], ["curl.nc"]),
    ("cdfmean", ["-f", T, "-v", "votemper", "-p", "T", "-var", "-S"], ["cdfmean.nc", "cdfmean.txt", "cdfvar.txt", "cdfsum.txt"]),
    ("cdfmean", ["-f", T, "-v", "vosaline", "-p", "T", "-w", "3", "20", "2", "15", "2", "4", "-B", "new_maskglo.nc", "tmaskatl", "tmaskpac", "-full",
                 "-o", "mw.nc"], ["mw.nc", "cdfmean.txt"]),
    ("cdfmean", ["-f", "VVL_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-vvl"], ["cdfmean.nc", "cdfmean.txt"]),
    ("cdfmean", ["-f", T, "-v", "votemper", "-p", "T", "-zeromean", "-oz", "zm.nc"], ["cdfmean.nc", "zm.nc"]),
    ("cdfsum", ["-f", T, "-v", "votemper", "-p", "T"], ["cdfsum.nc"]),
    ("cdfsum", ["-f", V, "-v", "vomecrty", "-p", "V", "-w", "3", "20", "2", "15", "2", "4", "-M", "mask.nc", "umask", "-o", "w.nc"], ["w.nc"]),
    ("cdfzonalsum", ["-f", T, "-p", "T", "-b", "new_maskglo.nc"], ["zonalsum.nc"]),
    ("cdfzonalsum", ["-f", T, "-p", "T", "-l", "vosaline", "-pdeg", "-pdep"], ["zonalintdeg.nc"]),
    ("cdfzonalmean", ["-f", T, "-p", "T", "-b", "new_maskglo.nc", "-max", "-l", "votemper"], ["zonalmean.nc"]),
    ("cdfvhst", ["-f", "vt_keep.nc", "-o", "vh.nc"], ["vh.nc"]),
    ("cdfvhst", ["-f", "vt_keep.nc", "-full", "-o", "vh.nc"], ["vh.nc"]),
    ("cdfvtrp", ["-u", U, "-v", V, "-bathy", "-o", "tr.nc"], ["tr.nc"]),
    ("cdfvtrp", ["-u", "VVL_y0001_gridU.nc", "-v", "VVL_y0001_gridV.nc", "-vvl", "-o", "tr.nc"], ["tr.nc"]),
    ("cdfvint", ["-f", T, "-o", "vi.nc"], ["vi.nc"]),
    ("cdfvint", ["-f", T, "-v", "vosaline", "-OCCI", "-smean", "-full", "-w", "3", "20", "2", "15", "-o", "sm.nc"], ["sm.nc"]),
    ("cdfvint", ["-f", "VVL_y0001_gridT.nc", "-vvl", "-o", "vi.nc"], ["vi.nc"]),
    ("cdfvertmean", ["-f", T, "-l", "votemper,vosaline", "-p", "T", "-zlim", "15", "120"], ["vertmean.nc"]),
    ("cdfvertmean", ["-f", T, "-l", "votemper", "-p", "T", "-zlim", "15", "120", "-full"], ["vertmean.nc"]),
    ("cdfsig0", ["-t", T], ["sig0.nc"]),                                                          # slabs in through the ring, out through page-locked arrays
    ("cdfsig0", ["-t", T, "-teos10"], ["sig0.nc"]),
    ("cdfsigi", ["-t", T, "-r", "2000"], ["sigi.nc"]),
    ("cdfmocsig", ["-v", V, "-t", T, "-r", "2000"], ["mocsig.nc"]),                               # V, T, S and e3v through the ring
    ("cdfmocsig", ["-v", V, "-t", T, "-r", "0", "-isodep", "-teos10"], ["mocsig.nc"]),
    ("cdfmocsig", ["-v", "EIV_y0001_gridV.nc", "-t", T, "-r", "1000", "-eiv", "-full"], ["mocsig.nc"]),
    ("cdfmocsig", ["-v", "VVL_y0001_gridV.nc", "-t", "VVL_y0001_gridT.nc", "-ntr", "-vvl"], ["mocsig.nc"]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("ring", [{}, {"CDFT_FEED_MB": "0.0001", "CDFT_FEED_SLOTS": "2"}], ids=["default_ring", "one_plane_two_slots"])
def test_both_routes_give_the_same_stdout_and_files(case, ring):
    d = case
    rc, out, err = run("cdfvT", ["-c", "SYN", "-l", "y0001", "-o", "vt_keep.nc"], d, env={"CDFT_FEED": "0"})
    assert rc == 0, out + err
    for tool, args, files, *more in RUNS:
        res = {}
        for feed in ("0", "1"):
            rc, out, err = run(tool, args, d, env=dict(ring, CDFT_FEED=feed, **(more[0] if more else {})))
            data = {f: open(os.path.join(d, f), "rb").read() for f in (files or [])} if rc == 0 else {}
            for f in data:
                os.remove(os.path.join(d, f))
            res[feed] = (rc, out, data)
        assert res["0"][0] == res["1"][0] and (files is None or res["1"][0] == 0), (tool, args, res["1"][1][-400:])
        assert res["0"][1] == res["1"][1], (tool, args)
        assert res["0"][2] == res["1"][2], (tool, args)


@pytest.mark.gpu
def test_the_feed_ring_of_the_c_abi(gpu):
    """cdft_feed_*: slots handed out in turn, a slot reused only after its copy has left it, copies ordered before work queued
    on the feed's stream; resize keeps the stream"""
    import ctypes as C

    import numpy as np
    import torch
    from cdftools_amd import _lib
    lib = _lib.load()
    feed, st, st2 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    gpu.check(lib.cdft_feed_create(C.byref(feed), 4096, 2))
    gpu.check(lib.cdft_feed_stream(feed, C.byref(st)))
    gpu.check(lib.cdft_feed_resize(feed, 1 << 20, 3))
    gpu.check(lib.cdft_feed_stream(feed, C.byref(st2)))
    assert st.value == st2.value
    n = 37
    want = np.arange(n * (1 << 18), dtype=np.float32)
    dst = torch.zeros(want.size, dtype=torch.float32, device="cuda")
    seen = []
    for i in range(n):
        host, nb = C.c_void_p(), C.c_size_t()
        gpu.check(lib.cdft_feed_slot(feed, C.byref(host), C.byref(nb)))
        assert nb.value == 1 << 20
        seen.append(host.value)
        np.frombuffer((C.c_char * (1 << 20)).from_address(host.value), dtype=np.float32)[:] = want[i << 18:(i + 1) << 18]
        gpu.check(lib.cdft_feed_push(feed, C.c_void_p(dst.data_ptr()), i << 20, 1 << 20))
    assert seen[:3] == seen[3:6] and len(set(seen)) == 3
    gpu.check(lib.cdft_feed_wait(feed))
    assert np.array_equal(dst.cpu().numpy(), want)
    with pytest.raises(gpu.CdftError):
        gpu.check(lib.cdft_feed_push(feed, C.c_void_p(dst.data_ptr()), 0, (1 << 20) + 1))
    gpu.check(lib.cdft_feed_destroy(feed))


@pytest.mark.gpu
def test_eos_results_into_page_locked_arrays(gpu):
    """out_space = CDFT_MEM_PINNED of the cdft_eos_* entries: the result is on its way when the call returns and there after
    cdft_stream_sync; two calls in a row share the staging buffer in stream order"""
    import ctypes as C

    import numpy as np
    from cdftools_amd import _lib
    lib = _lib.load()
    n = 300_001
    rng = np.random.default_rng(5)
    t = (-2 + 32 * rng.random(n)).astype(np.float32)
    s = np.where(rng.random(n) < 0.3, 0, 30 + 8 * rng.random(n)).astype(np.float32)
    want = gpu.sig_tool(t, s, pref=2000.0, spval=0.0)
    want0 = gpu.sig_tool(t, s, pref=0.0, spval=0.0)
    bufs = []
    for _ in range(2):
        p = C.c_void_p()
        gpu.check(lib.cdft_host_alloc(C.byref(p), 4 * n))
        bufs.append(p)
    for p, pref in zip(bufs, (2000.0, 0.0)):
        gpu.check(lib.cdft_eos_sig_tool(t.ctypes.data, s.ctypes.data, n, C.c_float(pref), C.c_float(0.0), gpu.EOS_80, 0, p,
                                        _lib.MEM_HOST, _lib.MEM_PINNED, None))
    gpu.check(lib.cdft_stream_sync(None))
    got = [np.frombuffer((C.c_char * (4 * n)).from_address(p.value), dtype=np.float32).copy() for p in bufs]
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want0) and not np.array_equal(want, want0)
    for p in bufs:
        gpu.check(lib.cdft_host_free(p))


@pytest.mark.gpu
def test_bytes_through_the_ring_widened_to_r4_in_hbm(gpu):
    """cdft_feed_push_i1_as_r4: a slot of signed bytes (a NetCDF byte mask) arrives in HBM as the r4 values nf90_get_var would have made of
    them on the host -- any count (tails that are not a multiple of 4), any element offset (stores that are not 16-byte aligned)"""
    import ctypes as C

    import numpy as np
    import torch
    from cdftools_amd import _lib
    lib = _lib.load()
    feed = C.c_void_p()
    gpu.check(lib.cdft_feed_create(C.byref(feed), 1 << 16, 2))
    rng = np.random.default_rng(3)
    dst = torch.full((300000,), -7.0, dtype=torch.float32, device="cuda")
    want = np.full(300000, -7.0, np.float32)
    off = 0
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 1023, 4097, 65536, 7, 40001):
        host, nb = C.c_void_p(), C.c_size_t()
        gpu.check(lib.cdft_feed_slot(feed, C.byref(host), C.byref(nb)))
        b = rng.integers(-128, 128, n, dtype=np.int8)
        np.frombuffer((C.c_char * n).from_address(host.value), dtype=np.int8)[:] = b
        gpu.check(lib.cdft_feed_push_i1_as_r4(feed, C.c_void_p(dst.data_ptr()), off, n))
        want[off:off + n] = b.astype(np.float32)
        off += n + (n % 3)                                         # (gaps: what lies between two pushes is not touched)
    gpu.check(lib.cdft_feed_wait(feed))
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), want)
    assert lib.cdft_feed_push_i1_as_r4(feed, C.c_void_p(dst.data_ptr()), 0, (1 << 16) + 1) != 0       # more than a slot holds
    gpu.check(lib.cdft_feed_destroy(feed))

