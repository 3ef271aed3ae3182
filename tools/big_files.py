"""CDF-5 files of grids whose 3-D fields pass 4 GiB (eORCA12: 4.68 GB), written through the bundled backend (ncc_* over
ctypes; scipy cannot write the format) one level at a time, so that the generator's memory stays a few planes.
Writer: the small wrapper; make_big_case: the file set tests/nemo_files.make_case writes (mesh_hgr, mesh_zgr, mask,
new_maskglo, SYN_y0001_grid{T,U,V}.nc), with r8 sums of what it wrote for the programs' printed integrals to be held against."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTE, CHAR, FLOAT, DOUBLE = 1, 2, 5, 6
nc = C.CDLL(os.path.join(ROOT, "cdftools_amd", "lib", "libncclassic.so"))
nc.ncc_strerror.restype = C.c_char_p


def ok(rc):
    if rc != 0:
        raise RuntimeError(nc.ncc_strerror(rc).decode())


class Writer:
    def __init__(self, path, dims):
        self.id = C.c_int()
        ok(nc.ncc_create(path.encode(), 5, 1, C.byref(self.id)))
        self.dim = {}
        for name, n in dims:
            d = C.c_int()
            ok(nc.ncc_def_dim(self.id, name.encode(), C.c_size_t(n or 0), C.byref(d)))
            self.dim[name] = d.value
        self.var = {}

    def define(self, name, xtype, dims, **atts):
        ids = (C.c_int * len(dims))(*[self.dim[d] for d in dims])
        v = C.c_int()
        ok(nc.ncc_def_var(self.id, name.encode(), xtype, len(dims), ids, C.byref(v)))
        for k, val in atts.items():
            if isinstance(val, str):
                b = val.encode()
                ok(nc.ncc_put_att(self.id, v, k.encode(), CHAR, C.c_size_t(len(b)), CHAR, b))
            else:                                                     # one r4 number (_FillValue ...)
                x = C.c_float(float(val))
                ok(nc.ncc_put_att(self.id, v, k.encode(), FLOAT, C.c_size_t(1), FLOAT, C.byref(x)))
        self.var[name] = (v.value, xtype, len(dims))

    def enddef(self):
        ok(nc.ncc_enddef(self.id))

    def put(self, name, start, a):
        v, xtype, nd = self.var[name]
        a = np.ascontiguousarray(a)
        mt = {np.dtype(np.int8): BYTE, np.dtype(np.float32): FLOAT, np.dtype(np.float64): DOUBLE}[a.dtype]
        st = (C.c_size_t * nd)(*start)
        ct = (C.c_size_t * nd)(*([1] * (nd - a.ndim) + list(a.shape)))
        ok(nc.ncc_put_vara(self.id, v, st, ct, mt, a.ctypes.data_as(C.c_void_p)))

    def close(self):
        ok(nc.ncc_close(self.id))


def make_big_case(d, npi, npj, npk, npt=1, seed=20240601):
    """The file set of tests/nemo_files.make_case (same names, dimensions, variables and attributes; its own random
    fields), level by level.  -> {"heat": sum of e1t*e2t*e3t*tmask*T of record 1 in r8, "vol": sum of e1t*e2t*e3t*tmask,
    "tvol": sum of T*e1t*e2t*e3t*tmask, "wet": wet T cells}."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    rng = np.random.default_rng(seed)
    f32 = np.float32
    metric = {nm: (f32(2000.0) + f32(26000.0) * rng.random((npj, npi), dtype=f32)).astype(f32)
              for nm in ("e1t", "e2t", "e1u", "e2u", "e1v", "e2v", "e1f", "e2f")}
    e31d = synth.e3_profile(npk)
    gdepw = np.concatenate([[0.0], np.cumsum(e31d[:-1], dtype=np.float64)]).astype(f32)
    gdept = (gdepw + 0.5 * e31d).astype(f32)
    mbathy = np.where(rng.random((npj, npi)) < 0.35, 0, 1 + np.floor((npk - 1) * rng.random((npj, npi))).astype(np.int32))
    mbathy = np.minimum(mbathy, npk - 1).astype(np.int32)
    mbathy[0, :] = 0
    frac = (f32(0.2) + f32(0.8) * rng.random((npj, npi), dtype=f32)).astype(f32)
    gphi = np.broadcast_to(np.linspace(-78.0, 89.9, npj, dtype=f32)[:, None], (npj, npi)).copy()
    lam = np.linspace(-180.0, 180.0, npi - 2, endpoint=False, dtype=f32)
    glam = np.broadcast_to(np.concatenate([lam, lam[:2]])[None, :], (npj, npi)).copy()
    band = (np.arange(npi) * 3) // npi

    dims4 = [("t", None), ("z", npk), ("y", npj), ("x", npi)]
    w = Writer(os.path.join(d, "mesh_hgr.nc"), dims4)
    names2 = list(metric) + ["glamf", "gphif", "gphiv", "gphit", "gphiu"]
    for nm in names2:
        w.define(nm, FLOAT, ("t", "y", "x"))
    w.enddef()
    for nm in names2:
        w.put(nm, (0, 0, 0), metric[nm] if nm in metric else glam if nm == "glamf" else gphi)
    w.close()

    t1 = (0 < mbathy).astype(np.int8)
    w = Writer(os.path.join(d, "new_maskglo.nc"), [("y", npj), ("x", npi)])
    for nm in ("tmaskatl", "tmaskpac", "tmaskind"):
        w.define(nm, BYTE, ("y", "x"))
    w.enddef()
    for b, nm in enumerate(("tmaskatl", "tmaskpac", "tmaskind")):
        w.put(nm, (0, 0), (t1 * (band == b)[None]).astype(np.int8))
    w.close()

    wz = Writer(os.path.join(d, "mesh_zgr.nc"), dims4)
    for nm in ("e3t_0", "e3u_0", "e3v_0"):
        wz.define(nm, FLOAT, ("t", "z", "y", "x"))
    for nm in ("gdepw_1d", "gdept_1d", "e3t_1d"):
        wz.define(nm, FLOAT, ("t", "z"))
    wz.define("hdepw", FLOAT, ("t", "y", "x"))
    wz.enddef()
    wz.put("gdepw_1d", (0, 0), gdepw); wz.put("gdept_1d", (0, 0), gdept); wz.put("e3t_1d", (0, 0), e31d)
    wm = Writer(os.path.join(d, "mask.nc"), dims4)
    for nm in ("tmask", "umask", "vmask", "fmask"):
        wm.define(nm, BYTE, ("t", "z", "y", "x"))
    wm.enddef()

    def data_file(name, depname, fields, fill=None):
        f = Writer(os.path.join(d, name), [("time_counter", None), (depname, npk), ("y", npj), ("x", npi)])
        f.define("nav_lon", FLOAT, ("y", "x")); f.define("nav_lat", FLOAT, ("y", "x"))
        f.define(depname, FLOAT, (depname,), units="m")
        f.define("time_counter", DOUBLE, ("time_counter",), units="seconds since 1950-01-01 00:00:00", calendar="noleap",
                 time_origin="1950-JAN-01 00:00:00")
        for nm in fields:
            atts = {} if fill is None else {"_FillValue": fill}
            f.define(nm, FLOAT, ("time_counter", depname, "y", "x"), **atts)
        f.enddef()
        f.put("nav_lon", (0, 0), glam); f.put("nav_lat", (0, 0), gphi)
        f.put(depname, (0,), gdept)
        f.put("time_counter", (0,), np.arange(1, npt + 1, dtype=np.float64) * 86400.0 * 30)
        return f

    ft = data_file("SYN_y0001_gridT.nc", "deptht", ("votemper", "vosaline"), fill=0.0)
    fu = data_file("SYN_y0001_gridU.nc", "depthu", ("vozocrtx",))
    fv = data_file("SYN_y0001_gridV.nc", "depthv", ("vomecrty",))

    stats = {"heat": 0.0, "vol": 0.0, "tvol": 0.0, "wet": 0}
    hdepw = np.zeros((npj, npi), np.float64)
    area = metric["e1t"].astype(np.float64) * metric["e2t"].astype(np.float64)
    for k in range(npk):
        tm = (k < mbathy).astype(np.int8)
        um = np.zeros_like(tm); vm = np.zeros_like(tm); fm = np.zeros_like(tm)
        um[:, :-1] = tm[:, :-1] * tm[:, 1:]
        vm[:-1] = tm[:-1] * tm[1:]
        fm[:-1, :-1] = tm[:-1, :-1] * tm[:-1, 1:] * tm[1:, :-1] * tm[1:, 1:]
        for nm, a in (("tmask", tm), ("umask", um), ("vmask", vm), ("fmask", fm)):
            wm.put(nm, (0, k, 0, 0), a)
        e3t = np.full((npj, npi), e31d[k], f32)
        bottom = mbathy == k + 1                                  # partial step: the last wet cell is thinner
        e3t[bottom] *= frac[bottom]
        e3u = e3t.copy(); e3v = e3t.copy()
        e3u[:, :-1] = np.minimum(e3t[:, :-1], e3t[:, 1:])
        e3v[:-1] = np.minimum(e3t[:-1], e3t[1:])
        wz.put("e3t_0", (0, k, 0, 0), e3t); wz.put("e3u_0", (0, k, 0, 0), e3u); wz.put("e3v_0", (0, k, 0, 0), e3v)
        hdepw += e3t.astype(np.float64) * tm
        vol = area * e3t.astype(np.float64) * tm
        for t in range(npt):
            tt = ((f32(-2) + f32(32) * rng.random((npj, npi), dtype=f32)) * tm).astype(f32)
            ss = ((f32(30) + f32(8) * rng.random((npj, npi), dtype=f32)) * tm).astype(f32)
            uu = ((rng.random((npj, npi), dtype=f32) - f32(0.5)) * um).astype(f32)
            vv = ((rng.random((npj, npi), dtype=f32) - f32(0.5)) * vm).astype(f32)
            ft.put("votemper", (t, k, 0, 0), tt); ft.put("vosaline", (t, k, 0, 0), ss)
            fu.put("vozocrtx", (t, k, 0, 0), uu); fv.put("vomecrty", (t, k, 0, 0), vv)
            if t == 0:
                stats["tvol"] += float((tt.astype(np.float64) * vol).sum())
        stats["vol"] += float(vol.sum())
        stats["wet"] += int(tm.sum())
    wz.put("hdepw", (0, 0, 0), hdepw.astype(f32))
    for f in (wz, wm, ft, fu, fv):
        f.close()
    stats["heat"] = stats["tvol"]
    return stats
