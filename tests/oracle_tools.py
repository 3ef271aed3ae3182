"""ctypes wrappers for oracle/cdf_oracle_tools.c (cdfheatc / cdfmhst / cdfpsi / cdfvT / cdfcurl
restatements).  Checker only, like oracle_lib.  Arrays use the reversed-dimension convention."""
import ctypes as C

import numpy as np

from oracle_lib import _p, load


def _f(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def _pp(a):
    return None if a is None else _p(a)


def heatc(e1t, e2t, tmask, temp, e3t=None, e31d=None, mxl=None, gdepw=None, mxlopt=0, r4sum=False):
    """-> sums (npk,3) r8: dsurf, dvol2d, dsum2d per level (src/cdfheatc.f90:225-229)."""
    lib = load()
    e1t, e2t, temp, e3t, e31d, mxl, gdepw = map(_f, (e1t, e2t, temp, e3t, e31d, mxl, gdepw))
    tm = np.ascontiguousarray(tmask, np.float32)
    npk, npj, npi = temp.shape
    out = np.empty((npk, 3), np.float64)
    lib.oracle_heatc(C.c_int(npi), C.c_int(npj), C.c_int(npk), _p(e1t), _p(e2t), _pp(e3t), _pp(e31d), _p(tm),
                     _p(temp), _pp(mxl), _pp(gdepw), C.c_int(mxlopt), C.c_int(1 if r4sum else 0), _p(out))
    return out


def mhst(e1v, masks, e3v=None, e31d=None, vt=None, vs=None, zv=None, zt=None, zs=None, zdim=False):
    """masks (nb,npj,npi) r4, basin 0 = global.  -> heat, salt: (npj,nb) or (npk,npj,nb) with zdim."""
    lib = load()
    e1v, e3v, e31d, vt, vs, zv, zt, zs = map(_f, (e1v, e3v, e31d, vt, vs, zv, zt, zs))
    masks = _f(masks)
    nb, npj, npi = masks.shape
    npk = (vt if vt is not None else zv).shape[0]
    shape = (npk, npj, nb) if zdim else (npj, nb)
    heat, salt = np.zeros(shape, np.float64), np.zeros(shape, np.float64)
    lib.oracle_mhst(C.c_int(npi), C.c_int(npj), C.c_int(npk), C.c_int(nb), _p(e1v), _pp(e3v), _pp(e31d),
                    _pp(vt), _pp(vs), _pp(zv), _pp(zt), _pp(zs), _p(masks), C.c_int(1 if zdim else 0),
                    _p(heat), _p(salt))
    return heat, salt


def psi(npk, zu=None, e2u=None, e3u=None, zv=None, e1v=None, e3v=None, e31d=None, iiref=None, ijref=None,
        fmask=None):
    """-> (psi4 (npj,npi) r4, dpsi8 (npj,npi) r8) (src/cdfpsi.f90:271-438, default/-V/-mean/-mask/-ref)."""
    lib = load()
    zu, e2u, e3u, zv, e1v, e3v, e31d, fmask = map(_f, (zu, e2u, e3u, zv, e1v, e3v, e31d, fmask))
    npj, npi = (e2u if e2u is not None else e1v).shape
    iiref = npi if iiref is None else iiref
    ijref = npj if ijref is None else ijref
    p4, d8 = np.empty((npj, npi), np.float32), np.empty((npj, npi), np.float64)
    lib.oracle_psi(C.c_int(npi), C.c_int(npj), C.c_int(npk), _pp(zu), _pp(e2u), _pp(e3u), _pp(zv), _pp(e1v),
                   _pp(e3v), _pp(e31d), C.c_int(iiref), C.c_int(ijref), _pp(fmask), _p(p4), _p(d8))
    return p4, d8


def psi_open(npk, zu, e2u, e3u, zv, e1v, e3v, lmean=False, e31d=None, iiref=None, ijref=None, noref=False, fmask=None):
    """-open path (src/cdfpsi.f90:316-366) -> (psi4, dpsi8)."""
    lib = load()
    zu, e2u, e3u, zv, e1v, e3v, e31d, fmask = map(_f, (zu, e2u, e3u, zv, e1v, e3v, e31d, fmask))
    npj, npi = e2u.shape
    iiref = npi if iiref is None else iiref
    ijref = npj if ijref is None else ijref
    p4, d8 = np.empty((npj, npi), np.float32), np.empty((npj, npi), np.float64)
    lib.oracle_psi_open(C.c_int(npi), C.c_int(npj), C.c_int(npk), _p(zu), _p(e2u), _pp(e3u), _p(zv), _p(e1v), _pp(e3v),
                        _pp(e31d), C.c_int(1 if lmean else 0), C.c_int(iiref), C.c_int(ijref), C.c_int(1 if noref else 0),
                        _pp(fmask), _p(p4), _p(d8))
    return p4, d8


def vt_level(u, v, t, s, e3u=None, e3v=None):
    """u,v,t,s (nfr,npj,npi) r4 for ONE level -> dict of (npj,npi) r8 means (src/cdfvT.f90:160-259)."""
    lib = load()
    u, v, t, s, e3u, e3v = map(_f, (u, v, t, s, e3u, e3v))
    nfr, npj, npi = u.shape
    o = {k: np.empty((npj, npi), np.float64) for k in ("vt", "vs", "ut", "us", "e3u", "e3v")}
    lib.oracle_vt(C.c_int(npi), C.c_int(npj), C.c_int(nfr), C.c_long(npi * npj), _p(u), _p(v), _p(t), _p(s),
                  _pp(e3u), _pp(e3v), _p(o["vt"]), _p(o["vs"]), _p(o["ut"]), _p(o["us"]), _p(o["e3u"]), _p(o["e3v"]))
    if e3u is None:
        del o["e3u"], o["e3v"]
    return o


def curl(un, vn, e1u, e2v, e1f, e2f, lperio=False, ltpoint=False, ff=None):
    """one level (npj,npi) -> rotn (npj,npi) r8 (src/cdfcurl.f90:262-301)."""
    lib = load()
    un, vn, e1u, e2v, e1f, e2f = map(_f, (un, vn, e1u, e2v, e1f, e2f))
    ffc = None if ff is None else np.ascontiguousarray(ff, np.float64)
    npj, npi = un.shape
    out = np.empty((npj, npi), np.float64)
    lib.oracle_curl(C.c_int(npi), C.c_int(npj), _p(un), _p(vn), _p(e1u), _p(e2v), _p(e1f), _p(e2f),
                    C.c_int(1 if lperio else 0), C.c_int(1 if ltpoint else 0), _pp(ffc), _p(out))
    return out


def vint_transport(vel, m2d, e3=None, e31d=None):
    """sum_k vel*m2d*e3*1.d0 (src/cdfvtrp.f90:211-218) -> (npj,npi) r8."""
    lib = load()
    vel, m2d, e3, e31d = map(_f, (vel, m2d, e3, e31d))
    npk, npj, npi = vel.shape
    out = np.empty((npj, npi), np.float64)
    lib.oracle_vint_transport(C.c_int(npi), C.c_int(npj), C.c_int(npk), _p(vel), _p(m2d), _pp(e3), _pp(e31d), _p(out))
    return out


def vtrp_slope(dtrpu, dtrpv, hdepw, e1u, e2v, tmask):
    """src/cdfvtrp.f90:210-252 (-bathy) -> (along, cross), each (npj,npi) r8."""
    lib = load()
    du, dv = np.ascontiguousarray(dtrpu, np.float64), np.ascontiguousarray(dtrpv, np.float64)
    hdepw, e1u, e2v, tmask = map(_f, (hdepw, e1u, e2v, tmask))
    npj, npi = du.shape
    a, c = np.empty((npj, npi), np.float64), np.empty((npj, npi), np.float64)
    lib.oracle_vtrp_slope(C.c_int(npi), C.c_int(npj), _p(du), _p(dv), _p(hdepw), _p(e1u), _p(e2v), _p(tmask), _p(a), _p(c))
    return a, c


def vhst_transport(xt, xs, m2d, e3=None, e31d=None):
    """src/cdfvhst.f90:167-178 for one pair -> (heat, salt), each (npj,npi) r8."""
    lib = load()
    xt, xs, m2d, e3, e31d = map(_f, (xt, xs, m2d, e3, e31d))
    npk, npj, npi = xt.shape
    h, s = np.empty((npj, npi), np.float64), np.empty((npj, npi), np.float64)
    lib.oracle_vhst_transport(C.c_int(npi), C.c_int(npj), C.c_int(npk), _p(xt), _p(xs), _p(m2d), _pp(e3), _pp(e31d), _p(h), _p(s))
    return h, s


def mean(e1, e2, zmask, ibmask, zv, e3=None, e31d=None, form=0):
    """src/cdfmean.f90:405-431 (form 0) or src/cdfsum.f90:303-327 (form 1; 2 for a file without depth)
    -> sums (npk,nb,4) r8: dsurf, dvol2d, dsum2d, dvar2d."""
    lib = load()
    e1, e2, zmask, zv, e3, e31d = map(_f, (e1, e2, zmask, zv, e3, e31d))
    ib = np.ascontiguousarray(ibmask, np.int16)
    npk, npj, npi = zv.shape
    nb = ib.shape[-1]
    out = np.empty((npk, nb, 4), np.float64)
    lib.oracle_mean_form(C.c_int(npi), C.c_int(npj), C.c_int(npk), C.c_int(nb), _p(e1), _p(e2), _pp(e3), _pp(e31d), _p(zmask),
                         _p(ib), _p(zv), C.c_int(form), _p(out))
    return out


def zonal(e1, e2, zmaskvar, zmask, zv, mean_form=False):
    """cdfzonalsum / cdfzonalmean loops -> dict of (nb,nlev,npj): sum [, area, max, min]."""
    lib = load()
    e1, e2, zmaskvar, zmask, zv = map(_f, (e1, e2, zmaskvar, zmask, zv))
    nlev, npj, npi = zv.shape
    nb = zmask.shape[-1]
    out = {"sum": np.empty((nb, nlev, npj), np.float64)}
    pa = px = pn = None
    if mean_form:
        out["area"] = np.empty((nb, nlev, npj), np.float64)
        out["max"], out["min"] = np.empty((nb, nlev, npj), np.float32), np.empty((nb, nlev, npj), np.float32)
        pa, px, pn = _p(out["area"]), _p(out["max"]), _p(out["min"])
    lib.oracle_zonal(C.c_int(npi), C.c_int(npj), C.c_int(nlev), C.c_int(nb), _p(e1), _p(e2), _p(zmaskvar), _p(zmask), _p(zv),
                     C.c_int(1 if mean_form else 0), _p(out["sum"]), pa, px, pn)
    return out


def vint_cumul(zt, tmask, e31d, gdepo, lstd=False, sclf=1.0, mean=False, e3t=None, full=False):
    """src/cdfvint.f90:266-311 -> (npko,npj,npi) r4.  e31d (npk) always drives the depth counters; with
    full=True it is also the thickness."""
    lib = load()
    zt, tmask, e31d, gdepo, e3t = map(_f, (zt, tmask, e31d, gdepo, e3t))
    npk, npj, npi = zt.shape
    out = np.empty((len(gdepo), npj, npi), np.float32)
    lib.oracle_vint_cumul(C.c_int(npi), C.c_int(npj), C.c_int(npk), C.c_int(len(gdepo)), _p(zt), _pp(e3t),
                          _p(e31d) if full else None, _p(tmask), _p(e31d), _p(gdepo), C.c_int(1 if lstd else 0),
                          C.c_float(sclf), C.c_int(1 if mean else 0), _p(out))
    return out


def vertmean(zv, zmask, gdep_first, gdep_last, rdep_up, rdep_down, e3=None, e31d=None):
    lib = load()
    zv, zmask, e3, e31d = map(_f, (zv, zmask, e3, e31d))
    nlev, npj, npi = zv.shape
    out = np.empty((npj, npi), np.float64)
    lib.oracle_vertmean(C.c_int(npi), C.c_int(npj), C.c_int(nlev), _p(zv), _pp(e3), _pp(e31d), _p(zmask), C.c_float(gdep_first),
                        C.c_float(gdep_last), C.c_float(rdep_up), C.c_float(rdep_down), _p(out))
    return out
