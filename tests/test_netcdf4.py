"""NetCDF-4 (HDF5) input: cdftools_amd/netcdf/nch5.c, the read plug-in that libncclassic.so loads when it meets an
HDF5 signature (the reference reads such files through libnetcdf, src/cdfio.F90:291-299).

No libnetcdf exists here, so the NetCDF-4 twins of the test files are written by tools/nc4_write.c: the published
layout, with the dimension-scale bookkeeping (CLASS / NAME / DIMENSION_LIST / REFERENCE_LIST) produced by HDF5's
own High-Level library exactly as for libnetcdf.  Checked: the data model and every value seen through the ncc_*
API are those of the classic twin (chunked + shuffled + deflated storage, unlimited record dimension, dimensions
without variable, coordinate variables, with and without creation-order tracking), and the Fortran programs give
byte-identical results on NetCDF-4 inputs.  Skipped where no HDF5 library was found at build time."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import nemo_files
from test_cli import read_nc, run, tools  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cdftools_amd", "netcdf")
LIB = os.environ.get("CDFT_NCCLASSIC_LIB") or os.path.join(ROOT, "cdftools_amd", "lib", "libncclassic.so")   # (sanitizer build: test_netcdf_asan.py)
PLUG = os.environ.get("CDFT_NCH5_LIB") or os.path.join(ROOT, "cdftools_amd", "lib", "libnch5.so")
W4 = os.path.join(ROOT, "tools", "bin", "libnc4write.so")
NPT = {1: np.int8, 3: np.int16, 4: np.int32, 5: np.float32, 6: np.float64}


@pytest.fixture(scope="module")
def libs():
    subprocess.check_call(["make", "-C", SRC], stdout=subprocess.DEVNULL)
    if not (os.path.exists(PLUG) and os.path.exists(W4)):
        pytest.skip("no HDF5 C library at build time: the NetCDF-4 plug-in was not built")
    nc = C.CDLL(LIB)
    nc.ncc_strerror.restype = C.c_char_p
    try:
        w4 = C.CDLL(W4)
    except OSError as e:                                          # the HDF5 library is not on this machine
        pytest.skip("libnc4write.so cannot be loaded: %s" % e)
    return nc, w4


def convert(w4, src, dst, deflate=4, order=1):
    rc = w4.nc4w_convert(src.encode(), dst.encode(), deflate, order)
    assert rc == 0, rc
    with open(dst, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"


def model(nc, path):
    """Everything the ncc_* API shows of a file: dims, unlimited, variables with dims / type / attributes / data."""
    ncid = C.c_int()
    rc = nc.ncc_open(path.encode(), 0, C.byref(ncid))
    assert rc == 0, nc.ncc_strerror(rc).decode()
    nd, nv, na, un = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert nc.ncc_inq(ncid, C.byref(nd), C.byref(nv), C.byref(na), C.byref(un)) == 0
    name, ln = C.create_string_buffer(256), C.c_size_t()
    dims = []
    for d in range(nd.value):
        assert nc.ncc_inq_dim(ncid, d, name, C.byref(ln)) == 0
        dims.append((name.value.decode(), ln.value))

    def atts(varid, n):
        out = {}
        for a in range(n):
            assert nc.ncc_inq_attname(ncid, varid, a, name) == 0
            xt, al = C.c_int(), C.c_size_t()
            assert nc.ncc_inq_att(ncid, varid, name.value, C.byref(xt), C.byref(al)) == 0
            if xt.value == 2:
                buf = C.create_string_buffer(al.value + 1)
                assert nc.ncc_get_att(ncid, varid, name.value, 2, buf) == 0
                out[name.value.decode()] = buf.raw[:al.value]
            else:
                arr = np.empty(al.value, NPT[xt.value])
                assert nc.ncc_get_att(ncid, varid, name.value, xt.value, arr.ctypes.data_as(C.c_void_p)) == 0
                out[name.value.decode()] = arr
        return out
    variables = {}
    for v in range(nv.value):
        xt, vnd, vna = C.c_int(), C.c_int(), C.c_int()
        dimids = (C.c_int * 16)()
        assert nc.ncc_inq_var(ncid, v, name, C.byref(xt), C.byref(vnd), dimids, C.byref(vna)) == 0
        vname = name.value.decode()
        vdims = [dims[dimids[i]][0] for i in range(vnd.value)]
        shape = [dims[dimids[i]][1] for i in range(vnd.value)]
        data = np.empty(shape, NPT[xt.value])
        assert nc.ncc_get_vara(ncid, v, None, None, xt.value, data.ctypes.data_as(C.c_void_p)) == 0
        variables[vname] = (xt.value, vdims, atts(v, vna.value), data)
    out = dict(dims=dims, unlim=dims[un.value][0] if un.value >= 0 else None, gatts=atts(-1, na.value), vars=variables, ncid=ncid)
    return out


def same_model(a, b):
    assert sorted(a["dims"]) == sorted(b["dims"]) and a["unlim"] == b["unlim"]
    assert set(a["vars"]) == set(b["vars"])
    for nm, (xt, vdims, at, data) in a["vars"].items():
        xt2, vdims2, at2, data2 = b["vars"][nm]
        assert (xt, vdims) == (xt2, vdims2), nm
        assert np.array_equal(data, data2), nm
        assert set(at) == set(at2), (nm, set(at) ^ set(at2))
        for k in at:
            assert (at[k] == at2[k]) if isinstance(at[k], bytes) else np.array_equal(at[k], at2[k]), (nm, k)
    assert set(a["gatts"]) == set(b["gatts"])


@pytest.mark.parametrize("deflate,order", [(4, 1), (0, 1), (4, 0)])
def test_netcdf4_twin_reads_like_the_classic_file(libs, tmp_path, deflate, order):
    nc, w4 = libs
    d = str(tmp_path)
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=3)
    for fname in ("SYN_y0001_gridT.nc", "mesh_zgr.nc", "mask.nc", "new_maskglo.nc"):
        src, dst = os.path.join(d, fname), os.path.join(d, "n4_" + fname)
        convert(w4, src, dst, deflate, order)
        a, b = model(nc, src), model(nc, dst)
        same_model(a, b)
        if order:                                               # creation order tracked: the file's own order
            assert list(a["vars"]) == list(b["vars"]) and a["dims"] == b["dims"]
        fmt = C.c_int()
        assert nc.ncc_inq_format(b["ncid"], C.byref(fmt)) == 0 and fmt.value == 3
        if fname.endswith("gridT.nc"):                          # getvar-style windows: one level of one record, a zoom
            vid = C.c_int()
            assert nc.ncc_inq_varid(b["ncid"], b"votemper", C.byref(vid)) == 0
            slab = np.empty((19, 26), np.float32)
            st, ct = (C.c_size_t * 4)(2, 3, 0, 0), (C.c_size_t * 4)(1, 1, 19, 26)
            assert nc.ncc_get_vara(b["ncid"], vid, st, ct, 5, slab.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(slab, rec["t"][2, 3])
            win = np.empty((4, 5), np.float64)                  # converted to r8 on the way
            st, ct = (C.c_size_t * 4)(0, 1, 7, 11), (C.c_size_t * 4)(1, 1, 4, 5)
            assert nc.ncc_get_vara(b["ncid"], vid, st, ct, 6, win.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(win, rec["t"][0, 1, 7:11, 11:16].astype(np.float64))
            st = (C.c_size_t * 4)(3, 0, 0, 0)                   # past the last record
            assert nc.ncc_get_vara(b["ncid"], vid, st, ct, 6, win.ctypes.data_as(C.c_void_p)) != 0
            assert nc.ncc_put_vara(b["ncid"], vid, st, ct, 6, win.ctypes.data_as(C.c_void_p)) != 0   # read only
        assert nc.ncc_close(a["ncid"]) == 0 and nc.ncc_close(b["ncid"]) == 0
    ncid = C.c_int()
    assert nc.ncc_open(os.path.join(d, "n4_mask.nc").encode(), 1, C.byref(ncid)) == -37    # NetCDF-4 files are never written


def test_missing_plugin_is_reported_not_fatal(tmp_path):
    """Without the plug-in a NetCDF-4 file gives an error code and a message; classic files are unaffected."""
    code = r'''
import ctypes as C, sys
nc = C.CDLL(sys.argv[1]); nc.ncc_strerror.restype = C.c_char_p
ncid = C.c_int()
rc = nc.ncc_open(sys.argv[2].encode(), 0, C.byref(ncid))
print(rc, nc.ncc_strerror(rc).decode())
'''
    subprocess.check_call(["make", "-C", SRC], stdout=subprocess.DEVNULL)
    p = tmp_path / "fake.nc"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + bytes(512))
    env = dict(os.environ, CDFT_NCH5_LIB=str(tmp_path / "nothere.so"))
    out = subprocess.run(["python3", "-c", code, LIB, str(p)], env=env, capture_output=True, text=True).stdout
    assert out.startswith("-101 ") and "plug-in" in out


@pytest.mark.gpu
def test_programs_give_identical_results_on_netcdf4_inputs(libs, tools, tmp_path):  # noqa: F811
    nc, w4 = libs
    d = str(tmp_path)
    nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=2)
    os.makedirs(os.path.join(d, "n4"))
    for fname in sorted(os.listdir(d)):
        if fname.endswith(".nc"):
            convert(w4, os.path.join(d, fname), os.path.join(d, "n4", fname), 5, 1)
    d4 = os.path.join(d, "n4")
    for tool, args, outfile in (("cdfmoc", ["-v", "SYN_y0001_gridV.nc"], "moc.nc"),
                                ("cdfheatc", ["-f", "SYN_y0001_gridT.nc"], "heatc.nc"),
                                ("cdfsig0", ["-t", "SYN_y0001_gridT.nc"], "sig0.nc"),
                                ("cdfzonalmean", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-b", "new_maskglo.nc"], "zonalmean.nc")):
        outs = []
        for where in (d, d4):
            rc, out, err = run(tool, args, where)
            assert rc == 0, (tool, out + err)
            outs.append(read_nc(os.path.join(where, outfile)))
        (v1, a1, d1), (v2, a2, d2) = outs
        assert d1 == d2 and list(v1) == list(v2), tool
        for nm in v1:
            assert np.array_equal(v1[nm], v2[nm], equal_nan=True), (tool, nm)


@pytest.mark.gpu
def test_nc4_option_writes_netcdf4_with_the_same_content(libs, tools, tmp_path):  # noqa: F811
    """-nc4 (src/cdfio.F90:291-299, :438-441): a NetCDF-4 file, chunked and deflated, whose content seen through the
    reader equals the classic output of the same command; the HDF5 side is checked with h5dump where it exists."""
    nc, w4 = libs
    d = str(tmp_path)
    nemo_files.make_case(d, npi=40, npj=62, npk=6, npt=3)
    for tool, args, outfile in (("cdfsig0", ["-t", "SYN_y0001_gridT.nc"], "sig0.nc"),
                                ("cdfvT", ["-c", "SYN", "-l", "y0001"], "vt.nc"),
                                ("cdfsum", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "T"], "cdfsum.nc")):
        rc, out, err = run(tool, args + ["-o", "classic_" + outfile], d)
        assert rc == 0, (tool, out + err)
        # (with the host feed forced on as well: cdfsig0 and cdfvT then queue their slabs behind, which a NetCDF-4 output
        # takes at once -- the HDF5 library is not entered from a second thread)
        for feed in ("1", "0"):
            rc, out, err = run(tool, args + ["-nc4", "-o", "nc4_" + outfile], d, env={"CDFT_FEED": feed})
            assert rc == 0, (tool, feed, out + err)
            if feed == "1":
                os.replace(os.path.join(d, "nc4_" + outfile), os.path.join(d, "nc4feed_" + outfile))
        a, b = model(nc, os.path.join(d, "nc4feed_" + outfile)), model(nc, os.path.join(d, "nc4_" + outfile))
        same_model(a, b)
        nc.ncc_close(a["ncid"]); nc.ncc_close(b["ncid"])
        p3, p4 = os.path.join(d, "classic_" + outfile), os.path.join(d, "nc4_" + outfile)
        with open(p4, "rb") as f:
            assert f.read(8) == b"\x89HDF\r\n\x1a\n", tool
        a, b = model(nc, p3), model(nc, p4)
        same_model(a, b)
        assert a["dims"] == b["dims"] and list(a["vars"]) == list(b["vars"])
        for k in a["gatts"]:
            assert a["gatts"][k] == b["gatts"][k] if isinstance(a["gatts"][k], bytes) else np.array_equal(a["gatts"][k], b["gatts"][k])
        fmt = C.c_int()
        assert nc.ncc_inq_format(b["ncid"], C.byref(fmt)) == 0 and fmt.value == 3
        nc.ncc_close(a["ncid"]); nc.ncc_close(b["ncid"])
    h5dump = "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        txt = subprocess.run([h5dump, "-H", "-p", os.path.join(d, "nc4_sig0.nc")], capture_output=True, text=True).stdout
        blk = txt[txt.index('DATASET "vosigma0"'):]
        assert "CHUNKED ( 1, 1, 2, 40 )" in blk and "COMPRESSION DEFLATE { LEVEL 1 }" in blk       # (t,z,y/30,x)
        assert "DIMENSION_LIST" in blk and "H5S_UNLIMITED" in blk


@pytest.mark.parametrize("dtype,xtype,shuffle", [(np.float32, 5, 1), (np.float64, 6, 1), (np.int16, 3, 0)])
def test_whole_chunk_writes_are_deflated_in_parallel_and_read_back(libs, tmp_path, monkeypatch, dtype, xtype, shuffle):
    """A put_vara that covers whole chunks of a deflated variable is shuffled and deflated by several threads and handed
    to HDF5 chunk by chunk (H5Dwrite_chunk); the library's own filters inflate it on the way back.  23 rows in chunks of
    5: the last chunk of every level is an edge chunk.  CDFT_NC4_SERIAL=1 (the library's filter pipeline) must give a
    file with the same content, and a write that does not start on a chunk boundary takes that route by itself."""
    nc, _ = libs
    npt, npk, npj, npi = 2, 3, 23, 17
    rng = np.random.default_rng(5)
    ref = (rng.standard_normal((npt, npk, npj, npi)) * 100).astype(dtype)

    def write(path, serial, rows_at_a_time=None):
        monkeypatch.setenv("CDFT_NC4_SERIAL", "1" if serial else "0")
        ncid, v = C.c_int(), C.c_int()
        assert nc.ncc_create(path.encode(), 3, 1, C.byref(ncid)) == 0
        ids = []
        for name, n in (("time_counter", 0), ("z", npk), ("y", npj), ("x", npi)):
            d = C.c_int()
            assert nc.ncc_def_dim(ncid, name.encode(), C.c_size_t(n), C.byref(d)) == 0
            ids.append(d.value)
        assert nc.ncc_def_var(ncid, b"field", xtype, 4, (C.c_int * 4)(*ids), C.byref(v)) == 0
        assert nc.ncc_def_var_chunking(ncid, v, (C.c_size_t * 4)(1, 1, 5, npi)) == 0
        assert nc.ncc_def_var_deflate(ncid, v, shuffle, 1) == 0
        assert nc.ncc_enddef(ncid) == 0
        mt = xtype
        for t in range(npt):
            for k in range(npk):
                if rows_at_a_time is None:
                    a = np.ascontiguousarray(ref[t, k])
                    rc = nc.ncc_put_vara(ncid, v, (C.c_size_t * 4)(t, k, 0, 0), (C.c_size_t * 4)(1, 1, npj, npi), mt,
                                         a.ctypes.data_as(C.c_void_p))
                    assert rc == 0, nc.ncc_strerror(rc).decode()
                else:                                   # 3 rows a call: not whole chunks
                    for j0 in range(0, npj, rows_at_a_time):
                        a = np.ascontiguousarray(ref[t, k, j0:j0 + rows_at_a_time])
                        rc = nc.ncc_put_vara(ncid, v, (C.c_size_t * 4)(t, k, j0, 0), (C.c_size_t * 4)(1, 1, a.shape[0], npi),
                                             mt, a.ctypes.data_as(C.c_void_p))
                        assert rc == 0, nc.ncc_strerror(rc).decode()
        assert nc.ncc_close(ncid) == 0

    def read(path):
        ncid, v = C.c_int(), C.c_int()
        assert nc.ncc_open(path.encode(), 0, C.byref(ncid)) == 0
        assert nc.ncc_inq_varid(ncid, b"field", C.byref(v)) == 0
        out = np.empty((npt, npk, npj, npi), dtype)
        assert nc.ncc_get_vara(ncid, v, (C.c_size_t * 4)(0, 0, 0, 0), (C.c_size_t * 4)(npt, npk, npj, npi), xtype,
                               out.ctypes.data_as(C.c_void_p)) == 0
        nc.ncc_close(ncid)
        return out

    def read_block(path, st, ct, serial):
        monkeypatch.setenv("CDFT_NC4_SERIAL", "1" if serial else "0")
        ncid, v = C.c_int(), C.c_int()
        assert nc.ncc_open(path.encode(), 0, C.byref(ncid)) == 0
        assert nc.ncc_inq_varid(ncid, b"field", C.byref(v)) == 0
        out = np.empty(ct, dtype)
        assert nc.ncc_get_vara(ncid, v, (C.c_size_t * 4)(*st), (C.c_size_t * 4)(*ct), xtype, out.ctypes.data_as(C.c_void_p)) == 0
        nc.ncc_close(ncid)
        return out

    plug = C.CDLL(PLUG)                                # the instance libncclassic has loaded: its counters
    plug.nch5_chunks_inflated.restype = plug.nch5_chunks_deflated.restype = C.c_ulong
    par, ser, rag = (str(tmp_path / n) for n in ("par.nc", "ser.nc", "rag.nc"))
    w0 = plug.nch5_chunks_deflated()
    write(par, serial=False)
    assert plug.nch5_chunks_deflated() - w0 == npt * npk * 5              # 5 chunks a level, all by the threaded route
    w0 = plug.nch5_chunks_deflated()
    write(ser, serial=True)
    assert plug.nch5_chunks_deflated() == w0
    write(rag, serial=False, rows_at_a_time=3)
    for p in (par, ser, rag):
        for serial in (False, True):                  # chunks inflated by threads here / by the library
            monkeypatch.setenv("CDFT_NC4_SERIAL", "1" if serial else "0")
            r0 = plug.nch5_chunks_inflated()
            assert np.array_equal(read(p), ref), (p, serial)
            assert plug.nch5_chunks_inflated() - r0 == (0 if serial else npt * npk * 5)
            # requests that cut through chunks, start inside one, or lie in a single chunk
            for st, ct in (((1, 0, 3, 2), (1, 3, 17, 13)), ((0, 1, 5, 0), (2, 1, 10, 17)), ((1, 2, 21, 4), (1, 1, 2, 6)),
                           ((0, 0, 0, 0), (1, 1, 23, 17)), ((0, 2, 7, 16), (2, 1, 9, 1))):
                want = ref[st[0]:st[0] + ct[0], st[1]:st[1] + ct[1], st[2]:st[2] + ct[2], st[3]:st[3] + ct[3]]
                assert np.array_equal(read_block(p, st, ct, serial), want), (p, serial, st, ct)
    monkeypatch.setenv("CDFT_NC4_SERIAL", "0")
    assert abs(os.path.getsize(par) - os.path.getsize(ser)) <= 0.02 * os.path.getsize(ser)       # the same deflate, chunk for chunk
    h5dump = "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):                                           # the HDF5 tool inflates every chunk too
        r = subprocess.run([h5dump, "-d", "/field", "-y", "-w", "0", par], capture_output=True, text=True)
        assert r.returncode == 0 and "error" not in r.stderr.lower() and "unable" not in r.stderr.lower(), r.stderr[-400:]
        body = r.stdout[r.stdout.index("DATA {") + 6:]
        body = body[:body.index("}")]
        nums = np.array([float(x) for x in body.replace("}", " ").replace(",", " ").split()
                         if x[0].isdigit() or (x[0] == "-" and len(x) > 1 and x[1].isdigit())])
        assert nums.size == ref.size and np.allclose(nums, ref.ravel().astype(np.float64), rtol=1e-5, atol=1e-30)
