"""ctypes access to oracle/_ref/liblift_ref.so: harnesses around LITERAL statements of the reference's
cdfmoc / cdfheatc / cdfmhst / cdfpsi / cdfvT / cdfcurl programs (oracle/ref_lift.py says what they are and
are not).  Test infrastructure only: used to pin the C oracle live (where the library was built) and to make
tests/golden/lift_*.npz (tests/make_golden_lift.py).  Arrays use the reversed-dimension convention of
tests/synth.py: (npt,npk,npj,npi) C-contiguous == Fortran (npi,npj,npk,npt)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "liblift_ref.so")
_lib = None


def available():
    return os.path.exists(LIB)


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
    return _lib


def _f(a, shape=None):
    a = np.ascontiguousarray(a, np.float32)
    return a if shape is None else a.reshape(shape)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i(*v):
    return [C.c_int(int(x)) for x in v]


def _dummy(*shape):
    return np.zeros(shape, np.float32)


def cdfmoc(g, zv, nbasins=5, full=False, decomp=None, teos10=False):
    """g: synth grid; zv (npt,npk,npj,npi).  decomp: dict(t=, s=) records for -decomp.
    -> dict var (npt,nvar,npk,npj) r4 [what putvar is handed], dmoc (npt,npk,npj,nb) r8, dec (npt,3,npk,npj,nb) r8."""
    npk, npj, npi = g["tmask"].shape
    zv = _f(zv, (-1, npk, npj, npi))
    npt = zv.shape[0]
    ldec = decomp is not None
    nbo = 6 if nbasins == 5 else 1
    nvar = nbo * (4 if ldec else 1)
    f2 = lambda k: _f(g[k])                         # noqa: E731
    m3 = lambda k: _f(g[k].astype(np.float32))      # noqa: E731
    t = _f(decomp["t"], zv.shape) if ldec else _dummy(*zv.shape)
    s = _f(decomp["s"], zv.shape) if ldec else _dummy(*zv.shape)
    o_var = np.zeros((npt, nvar, npk, npj), np.float32)
    o_dmoc = np.zeros((npt, npk, npj, nbasins), np.float64)
    o_dec = np.zeros((npt, 3, npk, npj, nbasins), np.float64)
    args = [f2("e1v"), f2("e1u"), f2("gphiv"), _f(g["gdepw_1d"]), _f(g["gdept_1d"]), _f(g["e3t_1d"]), m3("vmask"),
            m3("umask"), m3("tmask"), _f(g["e3v_0"]), m3("tmaskatl"), m3("tmaskind"), m3("tmaskpac"), zv, t, s]
    load().lift_cdfmoc(*_i(npi, npj, npk, npt, nbasins, ldec, full, teos10, nvar), *[_p(a) for a in args],
                       _p(o_var), _p(o_dmoc), _p(o_dec))
    return {"var": o_var, "dmoc": o_dmoc, "dec": o_dec}


def cdfheatc(g, temp, full=False, mxlopt=0, mxl=None):
    """temp (npt,npk,npj,npi) -> var (npt,3,npk) r8 [heatc3d, heatc2d(k), heatc3dpervol], sums (npt,npk,3) r8."""
    npk, npj, npi = g["tmask"].shape
    temp = _f(temp, (-1, npk, npj, npi))
    npt = temp.shape[0]
    mxl = _f(mxl, (npt, npj, npi)) if mxl is not None else _dummy(npt, npj, npi)
    o_var = np.zeros((npt, 3, npk), np.float64)
    o_sums = np.zeros((npt, npk, 3), np.float64)
    args = [_f(g["e1t"]), _f(g["e2t"]), _f(g["gdepw_1d"]), _f(g["e3t_1d"]), _f(g["tmask"].astype(np.float32)), _f(g["e3t_0"]),
            temp, mxl]
    load().lift_cdfheatc(*_i(npi, npj, npk, npt, full, mxlopt), *[_p(a) for a in args], _p(o_var), _p(o_sums))
    return {"var": o_var, "sums": o_sums}


def cdfmhst(g, vt=None, vs=None, zv=None, zt=None, zs=None, nbasins=5, full=False, zdim=False, mst=True):
    """-> var (npt,12,npk,npj) r4 [REAL(dmtrp) per output variable, level 1 only without zdim], zon (npt,8,npj) r8."""
    npk, npj, npi = g["tmask"].shape
    sep = vt is None
    shape = (-1, npk, npj, npi)
    a4 = lambda a, like: _f(a, shape) if a is not None else _dummy(*like)     # noqa: E731
    first = _f(zv if sep else vt, shape)
    npt = first.shape[0]
    like = first.shape
    o_var = np.zeros((npt, 12, npk, npj), np.float32)
    o_zon = np.zeros((npt, 8, npj), np.float64)
    m2 = lambda k: _f(g[k].astype(np.float32))    # noqa: E731
    args = [_f(g["e1v"]), _f(g["gphiv"]), _f(g["gdept_1d"]), _f(g["e3t_1d"]), _f(g["e3v_0"]), _f(g["vmask"][0].astype(np.float32)),
            m2("tmaskatl"), m2("tmaskpac"), m2("tmaskind"), a4(vt, like), a4(vs, like), a4(zv, like), a4(zt, like), a4(zs, like)]
    load().lift_cdfmhst(*_i(npi, npj, npk, npt, nbasins, sep, full, zdim, mst), *[_p(a) for a in args], _p(o_var), _p(o_zon))
    return {"var": o_var, "zon": o_zon}


def cdfpsi(g, zu=None, zv=None, full=False, mask=False, mean=False, opn=False, ssh=None, lev=False, iiref=None, ijref=None,
           fmask=None):
    """-> var (npt,3,nlo,npj,npi) r4, dpsi (npt,npj,npi) r8.  zu / zv (npt,npk,npj,npi); which of them is given selects
    the default (U), -V, or both (-mean / -open)."""
    npk, npj, npi = g["tmask"].shape
    shape = (-1, npk, npj, npi)
    first = _f(zu if zu is not None else zv, shape)
    npt = first.shape[0]
    a4 = lambda a: _f(a, shape) if a is not None else _dummy(*first.shape)     # noqa: E731
    nlo = npk if lev else 1
    o_var = np.zeros((npt, 3, nlo, npj, npi), np.float32)
    o_dpsi = np.zeros((npt, npj, npi), np.float64)
    fm = _f(fmask if fmask is not None else g["fmask"][0].astype(np.float32))
    sshv = _f(ssh, (npt, npj, npi)) if ssh is not None else _dummy(npt, npj, npi)
    ii = npi if iiref is None else iiref
    ij = npj if ijref is None else ijref
    args = [_f(g["e1v"]), _f(g["e2u"]), fm, _f(g["e3t_1d"]), _f(g["e3u_0"]), _f(g["e3v_0"]), a4(zu), a4(zv), sshv]
    load().lift_cdfpsi(*_i(npi, npj, npk, npt, zu is not None, zv is not None, full, mask, mean, opn, ssh is not None, lev, ii, ij),
                       *[_p(a) for a in args], C.c_int(nlo), _p(o_var), _p(o_dpsi))
    return {"var": o_var, "dpsi": o_dpsi}


def cdfvt(u, v, t, s, e3u=None, e3v=None):
    """u, v, t, s (ntags,nfr,npk,npj,npi) -> (6,npk,npj,npi) r8: vt, vs, ut, us [, e3u, e3v means with vvl]."""
    u = _f(u)
    ntags, nfr, npk, npj, npi = u.shape
    vvl = e3u is not None
    a = [u, _f(v), _f(t), _f(s), _f(e3u) if vvl else _dummy(*u.shape), _f(e3v) if vvl else _dummy(*u.shape)]
    o = np.zeros((6, npk, npj, npi), np.float64)
    load().lift_cdfvt(*_i(npi, npj, npk, ntags, nfr, vvl), *[_p(x) for x in a], _p(o))
    return o


def cdfcurl(g, u, v, forcing=False, tpoint=False, overf=False):
    """u, v (npt,nlev,npj,npi) -> drotn (npt,nlev,npj,npi) r8 as handed to putvar."""
    u = _f(u)
    npt, nlev, npj, npi = u.shape
    a = [_f(g["e1u"]), _f(g["e1f"]), _f(g["e2v"]), _f(g["e2f"]), _f(g["glamf"]), _f(g["gphif"]), u, _f(v)]
    o = np.zeros((npt, nlev, npj, npi), np.float64)
    load().lift_cdfcurl(*_i(npi, npj, nlev, npt, nlev, forcing, tpoint, overf), *[_p(x) for x in a], _p(o))
    return o
