"""ctypes access to oracle/liboracle.so -- the CPU restatement (checker only; tests, smoke, bench
cpu_baseline).  Same reversed-dimension array convention as cdftools_amd/hip.py."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def load():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)  # no-op when fresh
        _lib = C.CDLL(LIB)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def moc_decomp(e1v, e3v, ibmask, zv, ztemp, zsal, tmask, umask, e1u, zcoef, gdept, teos10=False, dmoc_sh=None):
    """One record of cdfmoc -decomp (oracle_moc_decomp) -> dict total / sh / bt / ag, each (npk,npj,nb) r8."""
    lib = load()
    f = lambda a: np.ascontiguousarray(a, np.float32)   # noqa: E731
    e1v, e3v, zv, ztemp, zsal, e1u, zcoef, gdept = map(f, (e1v, e3v, zv, ztemp, zsal, e1u, zcoef, gdept))
    ibmask = np.ascontiguousarray(ibmask, np.int16)
    tmask, umask = np.ascontiguousarray(tmask, np.uint8), np.ascontiguousarray(umask, np.uint8)
    npk, npj, npi = e3v.shape
    nb = ibmask.shape[-1]
    tot = moc(e1v, e3v, ibmask, zv)[0]
    sh = np.zeros_like(tot) if dmoc_sh is None else np.array(dmoc_sh, np.float64, order="C")
    bt, ag = np.empty_like(tot), np.empty_like(tot)
    lib.oracle_moc_decomp(C.c_int(npi), C.c_int(npj), C.c_int(npk), C.c_int(nb), _p(e1v), _p(e3v), _p(ibmask), _p(zv),
                          _p(ztemp), _p(zsal), _p(tmask), _p(umask), _p(e1u), _p(zcoef), _p(gdept),
                          C.c_int(1 if teos10 else 0), _p(tot), _p(sh), _p(bt), _p(ag))
    return {"total": tot, "sh": sh, "bt": bt, "ag": ag}


def mocsig(zv, zt, zs, e1v, ibmask, nbins, sigmin, sigstp, pref=0.0, teos10=False, e3v=None, e31d=None, zveiv=None, gdep=None,
           spv=(0.0, 0.0, 0.0), isodep=False, ntr=False):
    """oracle_mocsig -> dmoc (npj,nbins,nb) r8 [, depi]; ntr: neutral density instead of sigmai (-ntr)."""
    lib = load()
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)   # noqa: E731
    zv, zt, zs, e1v, e3v, e31d, zveiv, gdep = map(f, (zv, zt, zs, e1v, e3v, e31d, zveiv, gdep))
    ib = np.ascontiguousarray(ibmask, np.int16)
    npk, npj, npi = zv.shape
    nb = ib.shape[-1]
    dmoc = np.empty((npj, nbins, nb), np.float64)
    depi = np.empty((npj, nbins, nb), np.float64) if isodep else None
    P = lambda a: None if a is None else _p(a)   # noqa: E731
    lib.oracle_mocsig(C.c_int(npi), C.c_int(npj), C.c_int(npk), C.c_int(nb), P(zv), P(zveiv), P(zt), P(zs), P(e1v), P(e3v), P(e31d),
                      P(ib), P(gdep), C.c_float(spv[0]), C.c_float(spv[1]), C.c_float(spv[2]), C.c_float(pref),
                      C.c_int(2 if ntr else (1 if teos10 else 0)), C.c_int(nbins), C.c_float(sigmin), C.c_float(sigstp), P(dmoc), P(depi))
    return (dmoc, depi) if isodep else dmoc


def moc(e1v, e3v, ibmask, zv):
    """e1v (npj,npi), e3v (npk,npj,npi), ibmask (npj,npi,nb) i2, zv (npt,npk,npj,npi)
    -> dmoc (npt,npk,npj,nb) r8, one oracle_moc call per record."""
    lib = load()
    e1v = np.ascontiguousarray(e1v, np.float32)
    e3v = np.ascontiguousarray(e3v, np.float32)
    ibmask = np.ascontiguousarray(ibmask, np.int16)
    zv = np.ascontiguousarray(zv, np.float32)
    npk, npj, npi = e3v.shape
    nb = ibmask.shape[-1]
    zv = zv.reshape(-1, npk, npj, npi)
    out = np.empty((zv.shape[0], npk, npj, nb), np.float64)
    for t in range(zv.shape[0]):
        lib.oracle_moc(C.c_int(npi), C.c_int(npj), C.c_int(npk), C.c_int(nb), _p(e1v), _p(e3v),
                       _p(ibmask), _p(zv[t]), _p(out[t]))
    return out


def num_threads():
    return load().oracle_num_threads()


def set_num_threads(n):
    load().oracle_set_num_threads(C.c_int(int(n)))


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def sigmai(ptem, psal, pref=0.0, teos10=False, pref2d=None):
    """oracle_sigmai: r8 result with the shape of ptem."""
    lib = load()
    t, s = _f32(ptem), _f32(psal)
    out = np.empty(t.shape, np.float64)
    p2 = _f32(pref2d) if pref2d is not None else None
    lib.oracle_sigmai(_p(t), _p(s), C.c_long(t.size), C.c_float(pref), _p(p2) if p2 is not None else None,
                      C.c_int(1 if teos10 else 0), _p(out))
    return out


def sigmai_jmd94(ptem, psal, pref=0.0, pref2d=None):
    lib = load()
    t, s = _f32(ptem), _f32(psal)
    out = np.empty(t.shape, np.float64)
    p2 = _f32(pref2d) if pref2d is not None else None
    lib.oracle_sigmai_jmd94(_p(t), _p(s), C.c_long(t.size), C.c_float(pref),
                            _p(p2) if p2 is not None else None, _p(out))
    return out


def sigmantr(ptem, psal):
    """oracle_sigmantr: neutral density - 1000 (src/eos.f90:634-687), r8, not masked."""
    lib = load()
    t, s = _f32(ptem), _f32(psal)
    out = np.empty(t.shape, np.float64)
    lib.oracle_sigmantr(_p(t), _p(s), C.c_long(t.size), _p(out))
    return out


def eos_coef(which, ptem, psal, pdep=0.0, teos10=False, pdep2d=None):
    """oracle_eos_coef: which 0 alpha, 1 beta (polynomial), 2 alpha/beta, 3 beta (McDougall 1987); r8."""
    lib = load()
    t, s = _f32(ptem), _f32(psal)
    out = np.empty(t.shape, np.float64)
    p2 = _f32(pdep2d) if pdep2d is not None else None
    lib.oracle_eos_coef(C.c_int(which), _p(t), _p(s), C.c_long(t.size), C.c_float(pdep), _p(p2) if p2 is not None else None,
                        C.c_int(1 if teos10 else 0), _p(out))
    return out


def spice(ptem, psal, flament=False):
    lib = load()
    t, s = _f32(ptem), _f32(psal)
    out = np.empty(t.shape, np.float64)
    lib.oracle_spice(C.c_int(1 if flament else 0), _p(t), _p(s), C.c_long(t.size), _p(out))
    return out


def bn2(t_up, t_dn, s_up, s_dn, e3w, pdep, teos10=False, mcd87=False):
    """oracle_bn2: N^2 between two T levels, r4."""
    lib = load()
    a = [_f32(x) for x in (t_up, t_dn, s_up, s_dn, e3w)]
    out = np.empty(a[0].shape, np.float32)
    lib.oracle_bn2(C.c_int(1 if mcd87 else 0), *[_p(x) for x in a], C.c_long(a[0].size), C.c_float(pdep),
                   C.c_int(1 if teos10 else 0), _p(out))
    return out


def sig_tool(ptem, psal, pref=0.0, spval=0.0, teos10=False):
    """cdfsig0 / cdfsigi level loop: r4 result = sigma * (S /= spval)."""
    lib = load()
    t, s = _f32(ptem), _f32(psal)
    out = np.empty(t.shape, np.float32)
    lib.oracle_sig_tool(_p(t), _p(s), C.c_long(t.size), C.c_float(pref), C.c_float(spval),
                        C.c_int(1 if teos10 else 0), _p(out))
    return out


# ------------------------------------------------------------------ oracle/_ref (the real reference)
REF_EOS = os.path.join(ORACLE_DIR, "_ref", "libeos_ref.so")
_ref = None


def ref_eos_available():
    return os.path.exists(REF_EOS)


class RefEos:
    """The unmodified src/eos.f90 compiled by oracle/ref_build.sh (build container only)."""

    def __init__(self):
        global _ref
        if _ref is None:
            _ref = C.CDLL(REF_EOS)
        self.lib = _ref
        self.mode = None

    def init(self, teos10):
        if self.mode != bool(teos10):
            self.lib.ref_eos_init(C.c_int(1 if teos10 else 0))
            self.mode = bool(teos10)

    def _call(self, fn, t, s, *extra):
        t, s = _f32(t), _f32(s)
        t2, s2 = t.reshape(-1, t.shape[-1]) if t.ndim > 1 else t.reshape(1, -1), None
        s2 = s.reshape(t2.shape)
        out = np.empty(t2.shape, np.float64)
        kpj, kpi = t2.shape
        fn(_p(t2), _p(s2), *extra, C.c_int(kpi), C.c_int(kpj), _p(out))
        return out.reshape(t.shape)

    def sigma0(self, t, s, teos10=False):
        self.init(teos10)
        return self._call(self.lib.ref_sigma0, t, s)

    def sigmai(self, t, s, pref, teos10=False):
        self.init(teos10)
        return self._call(self.lib.ref_sigmai, t, s, C.c_float(pref))

    def sigmantr(self, t, s):
        return self._call(self.lib.ref_sigmantr, t, s)

    def spice(self, t, s, flament=False):
        t, s = _f32(t), _f32(s)
        out = np.empty(t.shape, np.float64)
        self.lib.ref_spice(_p(t), _p(s), C.c_int(t.shape[-1]), C.c_int(t.size // t.shape[-1]), C.c_int(1 if flament else 0), _p(out))
        return out

    def coef(self, which, t, s, pdep, teos10=False):
        """pdep a number: the 2-D members; an array like t: the profile members (one call over the raveled points)."""
        self.init(teos10)
        t, s = _f32(t), _f32(s)
        out = np.empty(t.shape, np.float64)
        if np.ndim(pdep) == 0:
            self.lib.ref_coef2d(C.c_int(which), _p(t), _p(s), C.c_float(pdep), C.c_int(t.shape[-1]), C.c_int(t.size // t.shape[-1]), _p(out))
        else:
            p = _f32(pdep)
            self.lib.ref_coefz(C.c_int(which), _p(t), _p(s), _p(p), C.c_int(t.size), _p(out))
        return out

    def coef0d(self, which, t, s, dep, teos10=False):
        self.init(teos10)
        out = C.c_double()
        self.lib.ref_coef0d(C.c_int(which), C.c_double(t), C.c_double(s), C.c_double(dep), C.byref(out))
        return out.value

    def bn2(self, t_up, t_dn, s_up, s_dn, e3w, pdep, teos10=False, mcd87=False, up_plane=1):
        """t_up .. e3w (npj,npi); the two levels are stacked as the reference's (kpi,kpj,2) with the upper one in plane
        up_plane (1 or 2), as cdfbn2's swapping iup / idown do."""
        self.init(teos10)
        tt = np.stack([_f32(t_up), _f32(t_dn)] if up_plane == 1 else [_f32(t_dn), _f32(t_up)])
        ss = np.stack([_f32(s_up), _f32(s_dn)] if up_plane == 1 else [_f32(s_dn), _f32(s_up)])
        e = _f32(e3w)
        out = np.empty(e.shape, np.float32)
        self.lib.ref_bn2(C.c_int(1 if mcd87 else 0), _p(tt), _p(ss), C.c_float(pdep), _p(e), C.c_int(e.shape[-1]), C.c_int(e.size // e.shape[-1]),
                         C.c_int(up_plane), C.c_int(3 - up_plane), _p(out))
        return out

    def sigmai2d(self, t, s, pref2d, teos10=False):
        self.init(teos10)
        p = _f32(pref2d)
        return self._call(self.lib.ref_sigmai2d, t, s, _p(p))

    def sigmai_jmd94(self, t, s, pref):
        self.init(False)
        return self._call(self.lib.ref_sigmai_jmd94, t, s, C.c_float(pref))

    def sigmai2d_jmd94(self, t, s, pref2d):
        self.init(False)
        p = _f32(pref2d)
        return self._call(self.lib.ref_sigmai2d_jmd94, t, s, _p(p))
