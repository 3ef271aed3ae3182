"""The bundled classic-NetCDF backend (cdftools_amd/netcdf/ncclassic.c, the layer under cdfio.F90)
against an independent implementation, scipy.io.netcdf_file, in both directions, for CDF-1 and CDF-2;
CDF-5 (which scipy cannot read) round-trips through the backend itself.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cdftools_amd", "netcdf")
LIB = os.environ.get("CDFT_NCCLASSIC_LIB") or os.path.join(ROOT, "cdftools_amd", "lib", "libncclassic.so")   # (sanitizer build: test_netcdf_asan.py)
BYTE, CHAR, SHORT, INT, FLOAT, DOUBLE = 1, 2, 3, 4, 5, 6
NPT = {BYTE: np.int8, SHORT: np.int16, INT: np.int32, FLOAT: np.float32, DOUBLE: np.float64}


@pytest.fixture(scope="module")
def nc():
    subprocess.check_call(["make", "-C", SRC], stdout=subprocess.DEVNULL)
    lib = C.CDLL(LIB)
    lib.ncc_strerror.restype = C.c_char_p
    return lib


def ok(lib, rc):
    assert rc == 0, lib.ncc_strerror(rc).decode()


def sz(*v):
    return (C.c_size_t * len(v))(*v)


def write_sample(lib, path, fmt):
    ncid = C.c_int()
    ok(lib, lib.ncc_create(path.encode(), fmt, 1, C.byref(ncid)))
    ids = {}
    for name, n in (("time_counter", 0), ("deptht", 3), ("y", 4), ("x", 5)):
        d = C.c_int()
        ok(lib, lib.ncc_def_dim(ncid, name.encode(), C.c_size_t(n), C.byref(d)))
        ids[name] = d.value
    v = {}
    for name, typ, dims in (("nav_lon", FLOAT, ("y", "x")), ("deptht", FLOAT, ("deptht",)),
                            ("time_counter", DOUBLE, ("time_counter",)),
                            ("votemper", FLOAT, ("time_counter", "deptht", "y", "x")),
                            ("tmask", BYTE, ("time_counter", "deptht", "y", "x")),
                            ("sossheig", SHORT, ("time_counter", "y", "x"))):
        vid = C.c_int()
        di = (C.c_int * len(dims))(*[ids[d] for d in dims])
        ok(lib, lib.ncc_def_var(ncid, name.encode(), typ, len(dims), di, C.byref(vid)))
        v[name] = vid.value
    ok(lib, lib.ncc_put_att(ncid, v["votemper"], b"units", CHAR, 4, CHAR, b"degC"))
    fill = np.array([9999.0], np.float32)
    ok(lib, lib.ncc_put_att(ncid, v["votemper"], b"_FillValue", FLOAT, 1, FLOAT, fill.ctypes.data_as(C.c_void_p)))
    sf = np.array([0.01], np.float64)
    ok(lib, lib.ncc_put_att(ncid, v["sossheig"], b"scale_factor", FLOAT, 1, DOUBLE, sf.ctypes.data_as(C.c_void_p)))
    ok(lib, lib.ncc_put_att(ncid, -1, b"history", CHAR, 7, CHAR, b"created"))
    ok(lib, lib.ncc_enddef(ncid))
    rng = np.random.default_rng(3)
    data = {"nav_lon": rng.random((4, 5)).astype(np.float32), "deptht": np.array([5, 15, 30], np.float32),
            "time_counter": np.array([86400.0, 172800.0]), "votemper": rng.random((2, 3, 4, 5)).astype(np.float32),
            "tmask": (rng.random((2, 3, 4, 5)) > 0.5).astype(np.int8),
            "sossheig": rng.integers(-300, 300, (2, 4, 5)).astype(np.int16)}
    mt = {np.dtype(np.float32): FLOAT, np.dtype(np.float64): DOUBLE, np.dtype(np.int8): BYTE, np.dtype(np.int16): SHORT}
    for name, a in data.items():
        if name in ("votemper", "tmask"):        # level by level, record by record, like cdfio's putvar
            for t in range(2):
                for k in range(3):
                    sl = np.ascontiguousarray(a[t, k])
                    ok(lib, lib.ncc_put_vara(ncid, v[name], sz(t, k, 0, 0), sz(1, 1, 4, 5), mt[a.dtype],
                                             sl.ctypes.data_as(C.c_void_p)))
        else:
            ok(lib, lib.ncc_put_vara(ncid, v[name], sz(*([0] * a.ndim)), sz(*a.shape), mt[a.dtype],
                                     a.ctypes.data_as(C.c_void_p)))
    ok(lib, lib.ncc_close(ncid))
    return data


@pytest.mark.parametrize("fmt", [1, 2])
def test_files_written_by_the_backend_are_read_by_scipy(nc, tmp_path, fmt):
    path = str(tmp_path / "w.nc")
    data = write_sample(nc, path, fmt)
    with netcdf_file(path, "r", mmap=False) as f:
        assert f.version_byte == fmt
        assert f.dimensions["time_counter"] is None and f.dimensions["x"] == 5
        for name, a in data.items():
            assert np.array_equal(f.variables[name][:], a), name
        assert f.variables["votemper"].units == b"degC"
        assert f.variables["votemper"]._FillValue == np.float32(9999.0)
        assert abs(f.variables["sossheig"].scale_factor - 0.01) < 1e-9
        assert f.history == b"created"


@pytest.mark.parametrize("fmt", [1, 2])
def test_files_written_by_scipy_are_read_by_the_backend(nc, tmp_path, fmt):
    path = str(tmp_path / "r.nc")
    rng = np.random.default_rng(1)
    want = rng.random((3, 2, 4, 6)).astype(np.float32)
    mask = (rng.random((2, 4, 6)) > 0.4).astype(np.int8)
    with netcdf_file(path, "w", version=fmt) as f:
        f.createDimension("time_counter", None)
        f.createDimension("z", 2); f.createDimension("y", 4); f.createDimension("x", 6)
        v = f.createVariable("vomecrty", "f4", ("time_counter", "z", "y", "x"))
        v[:] = want
        v.missing_value = np.float32(1e20)
        m = f.createVariable("vmask", "i1", ("z", "y", "x"))
        m[:] = mask
        t = f.createVariable("time_counter", "f8", ("time_counter",))
        t[:] = [1.0, 2.0, 3.0]
        t.units = "seconds since 1950-01-01"
    ncid = C.c_int()
    ok(nc, nc.ncc_open(path.encode(), 0, C.byref(ncid)))
    nd, nv, na, un = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ok(nc, nc.ncc_inq(ncid, C.byref(nd), C.byref(nv), C.byref(na), C.byref(un)))
    assert (nd.value, nv.value) == (4, 3)
    ln = C.c_size_t()
    name = C.create_string_buffer(256)
    ok(nc, nc.ncc_inq_dim(ncid, un.value, name, C.byref(ln)))
    assert name.value == b"time_counter" and ln.value == 3
    vid = C.c_int()
    ok(nc, nc.ncc_inq_varid(ncid, b"vomecrty", C.byref(vid)))
    slab = np.empty((4, 6), np.float32)                      # getvar: one level of one record
    ok(nc, nc.ncc_get_vara(ncid, vid, sz(2, 1, 0, 0), sz(1, 1, 4, 6), FLOAT, slab.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(slab, want[2, 1])
    win = np.empty((2, 3), np.float32)                       # zoom window (kimin/kjmin)
    ok(nc, nc.ncc_get_vara(ncid, vid, sz(0, 0, 1, 2), sz(1, 1, 2, 3), FLOAT, win.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(win, want[0, 0, 1:3, 2:5])
    mv = np.empty(1, np.float32)
    ok(nc, nc.ncc_get_att(ncid, vid, b"missing_value", FLOAT, mv.ctypes.data_as(C.c_void_p)))
    assert mv[0] == np.float32(1e20)
    ok(nc, nc.ncc_inq_varid(ncid, b"vmask", C.byref(vid)))
    asf = np.empty((4, 6), np.float32)                       # byte mask converted to r4 on read
    ok(nc, nc.ncc_get_vara(ncid, vid, sz(1, 0, 0), sz(1, 4, 6), FLOAT, asf.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(asf, mask[1].astype(np.float32))
    assert nc.ncc_inq_varid(ncid, b"nothere", C.byref(vid)) == -49
    assert nc.ncc_get_vara(ncid, 0, sz(3, 0, 0, 0), sz(1, 1, 4, 6), FLOAT, slab.ctypes.data_as(C.c_void_p)) != 0
    ok(nc, nc.ncc_close(ncid))
    assert nc.ncc_open(str(tmp_path / "missing.nc").encode(), 0, C.byref(ncid)) == -90


def test_cdf5_round_trip_and_redef_with_data(nc, tmp_path):
    path = str(tmp_path / "c5.nc")
    data = write_sample(nc, path, 5)
    assert open(path, "rb").read(4) == b"CDF\x05"
    ncid = C.c_int()
    ok(nc, nc.ncc_open(path.encode(), 1, C.byref(ncid)))
    # grow the header after data exist (cdfio re-enters define mode to add attributes, e.g. iweight)
    ok(nc, nc.ncc_redef(ncid))
    big = b"x" * 9000
    ok(nc, nc.ncc_put_att(ncid, -1, b"long_history", CHAR, len(big), CHAR, big))
    iw = np.array([12], np.int32)
    ok(nc, nc.ncc_put_att(ncid, 3, b"iweight", INT, 1, INT, iw.ctypes.data_as(C.c_void_p)))
    ok(nc, nc.ncc_enddef(ncid))
    iw[0] = 24                                               # in-place overwrite in data mode
    ok(nc, nc.ncc_put_att(ncid, 3, b"iweight", INT, 1, INT, iw.ctypes.data_as(C.c_void_p)))
    ok(nc, nc.ncc_close(ncid))
    ok(nc, nc.ncc_open(path.encode(), 0, C.byref(ncid)))
    for name, a in data.items():
        vid = C.c_int()
        ok(nc, nc.ncc_inq_varid(ncid, name.encode(), C.byref(vid)))
        got = np.empty_like(a)
        mt = {np.dtype(np.float32): FLOAT, np.dtype(np.float64): DOUBLE, np.dtype(np.int8): BYTE, np.dtype(np.int16): SHORT}
        ok(nc, nc.ncc_get_vara(ncid, vid, sz(*([0] * a.ndim)), sz(*a.shape), mt[a.dtype], got.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(got, a), name
    out = np.zeros(1, np.int32)
    ok(nc, nc.ncc_get_att(ncid, 3, b"iweight", INT, out.ctypes.data_as(C.c_void_p)))
    assert out[0] == 24
    ok(nc, nc.ncc_close(ncid))


def _define(lib, path, fmt, variables, dims):
    """create `path`, define dims {name: len} and variables [(name, dims)] as floats -> (ncid, status of enddef)"""
    ncid = C.c_int()
    ok(lib, lib.ncc_create(path.encode(), fmt, 1, C.byref(ncid)))
    ids = {}
    for name, n in dims.items():
        d = C.c_int()
        ok(lib, lib.ncc_def_dim(ncid, name.encode(), C.c_size_t(n), C.byref(d)))
        ids[name] = d.value
    for name, vd in variables:
        vid = C.c_int()
        di = (C.c_int * len(vd))(*[ids[d] for d in vd])
        ok(lib, lib.ncc_def_var(ncid, name.encode(), FLOAT, len(vd), di, C.byref(vid)))
    return ncid, lib.ncc_enddef(ncid)


def test_classic_formats_refuse_variables_and_offsets_that_do_not_fit(nc, tmp_path):
    """CDF-1 / CDF-2 store vsize (and CDF-1: begin) in 32 bits.  A layout that does not fit is refused with NC_EVARSIZE
    instead of being written with a clamped size or a truncated offset; the one exception of the format -- the last
    fixed-size variable of a file without record variables, or the last record variable -- and CDF-5 are accepted.
    (Headers only: no data is written, the files stay a few hundred bytes.)"""
    big = {"t": 0, "y": 40000, "x": 40000}                       # 40000 x 40000 floats = 6.4 GB
    EVARSIZE = -62
    p = str(tmp_path / "v.nc")
    for fmt in (1, 2):
        ncid, rc = _define(nc, p, fmt, [("a", ("y", "x")), ("b", ("y", "x"))], big)          # a big variable that is not the last
        assert rc == EVARSIZE and b"variable sizes" in nc.ncc_strerror(rc)
        nc.ncc_close(ncid)
        ncid, rc = _define(nc, p, fmt, [("a", ("y", "x")), ("r", ("t", "x"))], big)          # big fixed variable + a record variable
        assert rc == EVARSIZE
        nc.ncc_close(ncid)
        ncid, rc = _define(nc, p, fmt, [("r", ("t", "y", "x")), ("s", ("t", "x"))], big)     # a big record that is not the last
        assert rc == EVARSIZE
        nc.ncc_close(ncid)
    ncid, rc = _define(nc, p, 2, [("small", ("x",)), ("a", ("y", "x"))], big)                # the exception: last fixed variable
    ok(nc, rc)
    nc.ncc_close(ncid)
    ncid, rc = _define(nc, p, 2, [("s", ("t", "x")), ("r", ("t", "y", "x"))], big)           # the exception: last record variable
    ok(nc, rc)
    nc.ncc_close(ncid)
    # CDF-1: three 1.2 GB variables fit one by one, but the third one begins beyond 2^31
    mid = {"y": 17500, "x": 17500}
    ncid, rc = _define(nc, p, 1, [("a", ("y", "x")), ("b", ("y", "x")), ("c", ("y", "x"))], mid)
    assert rc == EVARSIZE
    nc.ncc_close(ncid)
    ncid, rc = _define(nc, p, 2, [("a", ("y", "x")), ("b", ("y", "x")), ("c", ("y", "x"))], mid)   # 64-bit offsets: fine
    ok(nc, rc)
    nc.ncc_close(ncid)
    ncid, rc = _define(nc, p, 5, [("a", ("y", "x")), ("b", ("y", "x"))], big)                # CDF-5: 64-bit sizes
    ok(nc, rc)
    nc.ncc_close(ncid)


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_large_reads_split_over_threads_and_swapped_through_a_buffer(nc, tmp_path, monkeypatch, threads):
    """Reads of tens of MB are cut into parts for CDFT_IO_THREADS threads, each reading through a 256 KiB buffer and
    byte-swapping from it into place: r4, r8, i2 and bytes, sizes that are no multiple of the buffer or of the thread
    count, a window that starts inside the variable -- against scipy's reading of the same file."""
    monkeypatch.setenv("CDFT_IO_THREADS", threads)
    path = str(tmp_path / "big.nc")
    rng = np.random.default_rng(17)
    n4, n8, n2, n1 = 9_000_037, 4_400_013, 17_000_003, 34_000_001          # 36 MB, 35 MB, 34 MB, 34 MB
    arrays = {"a4": rng.random(n4, dtype=np.float32), "a8": rng.random(n8), "a2": rng.integers(-30000, 30000, n2).astype(np.int16),
              "a1": rng.integers(-100, 100, n1).astype(np.int8)}
    with netcdf_file(path, "w", version=2) as f:
        for name, a in arrays.items():
            f.createDimension("d" + name, a.size)
            f.createVariable(name, a.dtype.newbyteorder(">") if a.dtype.itemsize > 1 else a.dtype, ("d" + name,))[:] = a
    ncid = C.c_int()
    ok(nc, nc.ncc_open(path.encode(), 0, C.byref(ncid)))
    mt = {"a4": FLOAT, "a8": DOUBLE, "a2": SHORT, "a1": BYTE}
    for name, a in arrays.items():
        vid = C.c_int()
        ok(nc, nc.ncc_inq_varid(ncid, name.encode(), C.byref(vid)))
        got = np.empty_like(a)
        ok(nc, nc.ncc_get_vara(ncid, vid, sz(0), sz(a.size), mt[name], got.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(got, a), name
        lo = 1_000_003                                                    # a window: neither end on a buffer boundary
        part = np.empty(a.size - lo - 5, a.dtype)
        ok(nc, nc.ncc_get_vara(ncid, vid, sz(lo), sz(part.size), mt[name], part.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(part, a[lo:lo + part.size]), name
    # conversions on the way, done by the same threads from the same buffers: a byte mask and an i2 variable read as r4, r8 as
    # r4 (rounded as the C cast rounds), r4 as r8; whole and from inside the variable
    for name, memtype, dt in (("a1", FLOAT, np.float32), ("a2", FLOAT, np.float32), ("a8", FLOAT, np.float32), ("a4", DOUBLE, np.float64)):
        a = arrays[name]
        vid = C.c_int()
        ok(nc, nc.ncc_inq_varid(ncid, name.encode(), C.byref(vid)))
        got = np.empty(a.size, dt)
        ok(nc, nc.ncc_get_vara(ncid, vid, sz(0), sz(a.size), memtype, got.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(got, a.astype(dt)), name
        lo = 777_777
        part = np.empty(a.size - lo - 3, dt)
        ok(nc, nc.ncc_get_vara(ncid, vid, sz(lo), sz(part.size), memtype, part.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(part, a[lo:lo + part.size].astype(dt)), name
    ok(nc, nc.ncc_close(ncid))


@pytest.mark.parametrize("threads", ["1", "5"])
def test_large_writes_split_over_threads_are_what_scipy_reads(nc, tmp_path, monkeypatch, threads):
    """Writes of the external type are cut into parts for CDFT_IO_THREADS threads, each swapping through a 256 KiB buffer and
    writing at its own offset: a 36 MB r4 variable and a 35 MB r8 one (sizes no multiple of anything), a byte variable, and
    a window written into the middle of a variable -- read back by scipy."""
    monkeypatch.setenv("CDFT_IO_THREADS", threads)
    path = str(tmp_path / "bigw.nc")
    rng = np.random.default_rng(23)
    arrays = {"a4": rng.random(9_000_037, dtype=np.float32), "a8": rng.random(4_400_013), "a1": rng.integers(-100, 100, 20_000_003).astype(np.int8)}
    mt = {"a4": FLOAT, "a8": DOUBLE, "a1": BYTE, "c4": FLOAT, "c2": SHORT}
    # conversions on the way out: r8 results into a float variable (what the programs' putvar of an r8 array does), r4 into i2
    given = {"c4": (rng.random(9_100_019) * 1e3, DOUBLE), "c2": (rng.integers(-30000, 30000, 17_000_005).astype(np.float32), FLOAT)}
    arrays.update({"c4": given["c4"][0].astype(np.float32), "c2": given["c2"][0].astype(np.int16)})
    ncid = C.c_int()
    ok(nc, nc.ncc_create(path.encode(), 2, 1, C.byref(ncid)))
    vid = {}
    for name, a in arrays.items():
        d, v = C.c_int(), C.c_int()
        ok(nc, nc.ncc_def_dim(ncid, ("d" + name).encode(), C.c_size_t(a.size), C.byref(d)))
        ok(nc, nc.ncc_def_var(ncid, name.encode(), mt[name], 1, (C.c_int * 1)(d.value), C.byref(v)))
        vid[name] = v.value
    ok(nc, nc.ncc_enddef(ncid))
    for name, a in arrays.items():
        src, memtype = given.get(name, (a, mt[name]))
        ok(nc, nc.ncc_put_vara(ncid, vid[name], sz(0), sz(a.size), memtype, src.ctypes.data_as(C.c_void_p)))
    patch8 = np.arange(8_400_003, dtype=np.float64) / 7                    # an r8 window into the middle of the float variable
    ok(nc, nc.ncc_put_vara(ncid, vid["c4"], sz(54_321), sz(patch8.size), DOUBLE, patch8.ctypes.data_as(C.c_void_p)))
    arrays["c4"][54_321:54_321 + patch8.size] = patch8.astype(np.float32)
    patch = np.arange(8_500_001, dtype=np.float32)                         # a second write over the middle of a4
    ok(nc, nc.ncc_put_vara(ncid, vid["a4"], sz(123_457), sz(patch.size), FLOAT, patch.ctypes.data_as(C.c_void_p)))
    ok(nc, nc.ncc_close(ncid))
    arrays["a4"][123_457:123_457 + patch.size] = patch
    with netcdf_file(path, "r", mmap=False) as f:
        for name, a in arrays.items():
            assert np.array_equal(f.variables[name][:], a), name


def test_close_reports_a_failed_queued_write(nc, tmp_path):
    """A write queued by ncc_put_vara_behind that fails and that no ncc_wait_writes has collected: ncc_close -- which drains the
    queue -- returns its error instead of 0 (a program that goes from its last putvar_behind straight to closeout must not
    leave a short file behind with exit status 0); a close after a collected error, or with nothing queued, returns 0."""
    nij = 1000
    for collect in (False, True):
        path = str(tmp_path / f"fail{int(collect)}.nc")
        ncid, dt, dz, dx, v4 = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        ok(nc, nc.ncc_create(path.encode(), 2, 1, C.byref(ncid)))
        ok(nc, nc.ncc_def_dim(ncid, b"t", C.c_size_t(0), C.byref(dt)))
        ok(nc, nc.ncc_def_dim(ncid, b"z", C.c_size_t(3), C.byref(dz)))
        ok(nc, nc.ncc_def_dim(ncid, b"x", C.c_size_t(nij), C.byref(dx)))
        ok(nc, nc.ncc_def_var(ncid, b"v4", FLOAT, 3, (C.c_int * 3)(dt.value, dz.value, dx.value), C.byref(v4)))
        ok(nc, nc.ncc_enddef(ncid))
        a = np.arange(nij, dtype=np.float32)
        ok(nc, nc.ncc_put_vara_behind(ncid, v4, sz(0, 0, 0), sz(1, 1, nij), FLOAT, a.ctypes.data_as(C.c_void_p)))
        ok(nc, nc.ncc_put_vara_behind(ncid, v4, sz(0, 3, 0), sz(1, 1, nij), FLOAT, a.ctypes.data_as(C.c_void_p)))   # outside z
        if collect:
            assert nc.ncc_wait_writes(ncid) != 0
            ok(nc, nc.ncc_close(ncid))
        else:
            assert nc.ncc_close(ncid) != 0
        with netcdf_file(path, "r", mmap=False) as f:                          # the good write is in the file either way
            assert np.array_equal(f.variables["v4"][0, 0], a)


@pytest.mark.parametrize("threads", ["1", "5"])
def test_writes_queued_behind_are_in_the_file_in_order(nc, tmp_path, monkeypatch, threads):
    """ncc_put_vara_behind: the levels of a record variable queued one after the other (more requests than the queue holds, r4
    and r8 -> float, a later write over an earlier one), reads of ANOTHER file in between, then ncc_wait_writes; a read of the
    same file waits for the queue by itself; an error of a queued write comes back from ncc_wait_writes; close drains."""
    monkeypatch.setenv("CDFT_IO_THREADS", threads)
    path, other = str(tmp_path / "behind.nc"), str(tmp_path / "other.nc")
    with netcdf_file(other, "w", version=2) as f:
        f.createDimension("n", 1000)
        f.createVariable("a", np.dtype(">f4"), ("n",))[:] = np.arange(1000, dtype=np.float32)
    nlev, nij = 150, 300_007
    rng = np.random.default_rng(41)
    ncid, dt, dz, dx, v4, v8 = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ok(nc, nc.ncc_create(path.encode(), 2, 1, C.byref(ncid)))
    ok(nc, nc.ncc_def_dim(ncid, b"t", C.c_size_t(0), C.byref(dt)))
    ok(nc, nc.ncc_def_dim(ncid, b"z", C.c_size_t(nlev), C.byref(dz)))
    ok(nc, nc.ncc_def_dim(ncid, b"x", C.c_size_t(nij), C.byref(dx)))
    ok(nc, nc.ncc_def_var(ncid, b"v4", FLOAT, 3, (C.c_int * 3)(dt.value, dz.value, dx.value), C.byref(v4)))
    ok(nc, nc.ncc_def_var(ncid, b"v8", FLOAT, 3, (C.c_int * 3)(dt.value, dz.value, dx.value), C.byref(v8)))
    ok(nc, nc.ncc_enddef(ncid))
    a4 = rng.random((nlev, nij), dtype=np.float32)
    a8 = rng.random((nlev, nij)) * 100
    oid, ovid = C.c_int(), C.c_int()
    ok(nc, nc.ncc_open(other.encode(), 0, C.byref(oid)))
    ok(nc, nc.ncc_inq_varid(oid, b"a", C.byref(ovid)))
    for k in range(nlev):
        ok(nc, nc.ncc_put_vara_behind(ncid, v4, sz(0, k, 0), sz(1, 1, nij), FLOAT, a4[k].ctypes.data_as(C.c_void_p)))
        ok(nc, nc.ncc_put_vara_behind(ncid, v8, sz(0, k, 0), sz(1, 1, nij), DOUBLE, a8[k].ctypes.data_as(C.c_void_p)))
        if k % 40 == 0:                                                   # the caller goes on reading another file
            got = np.empty(1000, np.float32)
            ok(nc, nc.ncc_get_vara(oid, ovid, sz(0), sz(1000), FLOAT, got.ctypes.data_as(C.c_void_p)))
            assert got[999] == 999
    patch = np.full(nij, -7.5, np.float32)
    ok(nc, nc.ncc_put_vara_behind(ncid, v4, sz(0, 3, 0), sz(1, 1, nij), FLOAT, patch.ctypes.data_as(C.c_void_p)))   # after level 3's first write
    a4[3] = patch
    back = np.empty(nij, np.float32)                                      # a read of the same file: waits for the queue
    ok(nc, nc.ncc_get_vara(ncid, v4, sz(0, nlev - 1, 0), sz(1, 1, nij), FLOAT, back.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(back, a4[nlev - 1])
    ok(nc, nc.ncc_wait_writes(ncid))
    ok(nc, nc.ncc_put_vara_behind(ncid, v4, sz(0, nlev, 0), sz(1, 1, nij), FLOAT, patch.ctypes.data_as(C.c_void_p)))   # outside the z dimension
    assert nc.ncc_wait_writes(ncid) != 0 and nc.ncc_wait_writes(ncid) == 0
    ok(nc, nc.ncc_put_vara_behind(ncid, v8, sz(1, 0, 0), sz(1, 1, nij), DOUBLE, a8[0].ctypes.data_as(C.c_void_p)))  # a second record, left to close
    ok(nc, nc.ncc_close(ncid))
    ok(nc, nc.ncc_close(oid))
    with netcdf_file(path, "r", mmap=False) as f:
        assert f.variables["v4"].shape[0] == 2
        assert np.array_equal(f.variables["v4"][0], a4)
        assert np.array_equal(f.variables["v8"][0], a8.astype(np.float32))
        assert np.array_equal(f.variables["v8"][1, 0], a8[0].astype(np.float32))
