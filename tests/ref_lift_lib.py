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


def _e3(static, records, npt):
    """-vvl: `records` (npt,npk,npj,npi) is the vertical metric of every record (the harness then reads h_e3(:,:,jk,jt), as the
    reference reads the e3 variable of the data file with ktime=jt); without it the one static field and kvvl = 0."""
    if records is None:
        return _f(static), 0
    r = _f(records)
    assert r.shape[0] == npt and r.shape[1:] == np.shape(static), (r.shape, np.shape(static), npt)
    return r, 1


def cdfmoc(g, zv, nbasins=5, full=False, decomp=None, teos10=False, e3v_t=None):
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
    e3v, vvl = _e3(g["e3v_0"], e3v_t, npt)
    args = [f2("e1v"), f2("e1u"), f2("gphiv"), _f(g["gdepw_1d"]), _f(g["gdept_1d"]), _f(g["e3t_1d"]), m3("vmask"),
            m3("umask"), m3("tmask"), e3v, m3("tmaskatl"), m3("tmaskind"), m3("tmaskpac"), zv, t, s]
    load().lift_cdfmoc(*_i(npi, npj, npk, npt, nbasins, ldec, full, teos10, nvar, vvl), *[_p(a) for a in args],
                       _p(o_var), _p(o_dmoc), _p(o_dec))
    return {"var": o_var, "dmoc": o_dmoc, "dec": o_dec}


def cdfheatc(g, temp, full=False, mxlopt=0, mxl=None, e3t_t=None):
    """temp (npt,npk,npj,npi) -> var (npt,3,npk) r8 [heatc3d, heatc2d(k), heatc3dpervol], sums (npt,npk,3) r8."""
    npk, npj, npi = g["tmask"].shape
    temp = _f(temp, (-1, npk, npj, npi))
    npt = temp.shape[0]
    mxl = _f(mxl, (npt, npj, npi)) if mxl is not None else _dummy(npt, npj, npi)
    o_var = np.zeros((npt, 3, npk), np.float64)
    o_sums = np.zeros((npt, npk, 3), np.float64)
    e3t, vvl = _e3(g["e3t_0"], e3t_t, npt)
    args = [_f(g["e1t"]), _f(g["e2t"]), _f(g["gdepw_1d"]), _f(g["e3t_1d"]), _f(g["tmask"].astype(np.float32)), e3t,
            temp, mxl]
    load().lift_cdfheatc(*_i(npi, npj, npk, npt, full, mxlopt, vvl), *[_p(a) for a in args], _p(o_var), _p(o_sums))
    return {"var": o_var, "sums": o_sums}


def cdfmhst(g, vt=None, vs=None, zv=None, zt=None, zs=None, nbasins=5, full=False, zdim=False, mst=True, e3v_t=None):
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
    e3v, vvl = _e3(g["e3v_0"], e3v_t, npt)
    args = [_f(g["e1v"]), _f(g["gphiv"]), _f(g["gdept_1d"]), _f(g["e3t_1d"]), e3v, _f(g["vmask"][0].astype(np.float32)),
            m2("tmaskatl"), m2("tmaskpac"), m2("tmaskind"), a4(vt, like), a4(vs, like), a4(zv, like), a4(zt, like), a4(zs, like)]
    load().lift_cdfmhst(*_i(npi, npj, npk, npt, nbasins, sep, full, zdim, mst, vvl), *[_p(a) for a in args], _p(o_var), _p(o_zon))
    return {"var": o_var, "zon": o_zon}


def cdfpsi(g, zu=None, zv=None, full=False, mask=False, mean=False, opn=False, ssh=None, lev=False, iiref=None, ijref=None,
           fmask=None, e3u_t=None, e3v_t=None):
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
    e3u, vvl = _e3(g["e3u_0"], e3u_t, npt)
    e3v, vvl2 = _e3(g["e3v_0"], e3v_t, npt)
    assert vvl == vvl2
    args = [_f(g["e1v"]), _f(g["e2u"]), fm, _f(g["e3t_1d"]), e3u, e3v, a4(zu), a4(zv), sshv]
    load().lift_cdfpsi(*_i(npi, npj, npk, npt, zu is not None, zv is not None, full, mask, mean, opn, ssh is not None, lev, ii, ij),
                       *[_p(a) for a in args], C.c_int(nlo), C.c_int(vvl), _p(o_var), _p(o_dpsi))
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


# ------------------------------------------------------------------------------------------ SURVEY 8f tools
def _m(a):
    return _f(np.asarray(a).astype(np.float32))


def cdfvhst(g, vt, vs, ut, us, full=False, e3v_t=None, e3u_t=None):
    """vt, vs, ut, us (npt,npk,npj,npi) -> (npt,4,npj,npi) r4: SNGL of the four vertical integrals (vt, vs, ut, us)."""
    vt = _f(vt)
    npt, npk, npj, npi = vt.shape
    o = np.zeros((npt, 4, npj, npi), np.float32)
    e3v, vvl = _e3(g["e3v_0"], e3v_t, npt)
    e3u, _ = _e3(g["e3u_0"], e3u_t, npt)
    a = [_f(g["e1v"]), _f(g["e2u"]), _f(g["e3t_1d"]), e3v, e3u, vt, _f(vs), _f(ut), _f(us)]
    load().lift_cdfvhst(*_i(npi, npj, npk, npt, full, vvl), *[_p(x) for x in a], _p(o))
    return o


def cdfvtrp(g, u, v, full=False, bathy=None, e3v_t=None, e3u_t=None):
    """u, v (npt,npk,npj,npi) -> (npt,4,npj,npi) r4: REAL(dtrpu), REAL(dtrpv) [, along- and cross-slope with bathy=hdepw]."""
    u = _f(u)
    npt, npk, npj, npi = u.shape
    o = np.zeros((npt, 4, npj, npi), np.float32)
    hd = _f(bathy) if bathy is not None else _dummy(npj, npi)
    e3v, vvl = _e3(g["e3v_0"], e3v_t, npt)
    e3u, _ = _e3(g["e3u_0"], e3u_t, npt)
    a = [_f(g["e1v"]), _f(g["e2u"]), _f(g["e1u"]), _f(g["e2v"]), _m(g["tmask"][0]), hd, _f(g["e3t_1d"]), e3v, e3u,
         u, _f(v)]
    load().lift_cdfvtrp(*_i(npi, npj, npk, npt, full, bathy is not None, vvl), *[_p(x) for x in a], _p(o))
    return o


def cdfsum(e1, e2, e3, mask, field, dep, forcing=False, nowght=False, nomsk=False, e3_t=None):
    """field (npt,npk,npj,npi) -> var (npt,3,npk) r4, sums (npt,npk,3) r8 (dsurf, dvol2d, dsum2d)."""
    field = _f(field)
    npt, npk, npj, npi = field.shape
    o_var, o_sums = np.zeros((npt, 3, npk), np.float32), np.zeros((npt, npk, 3), np.float64)
    e3, vvl = _e3(e3, e3_t, npt)
    a = [_f(e1), _f(e2), _f(dep), e3, _m(mask), field]
    load().lift_cdfsum(*_i(npi, npj, npk, npt, forcing, nowght, nomsk, vvl), *[_p(x) for x in a], _p(o_var), _p(o_sums))
    return {"var": o_var, "sums": o_sums}


def cdfmean(e1, e2, e3, e31d, mask, basins, field, dep, full=False, var=False, lsum=False, e3_t=None):
    """basins (nb,npj,npi) r4 0/1; field (npt,npk,npj,npi) -> var (npt,nv,npk) r4, var0 (npt,nv) r4, sums (npt,npk,nb,4) r8."""
    field = _f(field)
    npt, npk, npj, npi = field.shape
    basins = _m(basins)
    nb = basins.shape[0]
    nv = (2 + (2 if var else 0) + (2 if lsum else 0)) * nb
    o_var, o_var0 = np.zeros((npt, nv, npk), np.float32), np.zeros((npt, nv), np.float32)
    o_sums = np.zeros((npt, npk, nb, 4), np.float64)
    e3, vvl = _e3(e3, e3_t, npt)
    a = [_f(e1), _f(e2), _f(e31d), _f(dep), e3, _m(mask), basins, field]
    load().lift_cdfmean(*_i(npi, npj, npk, npt, nb, full, var, lsum, vvl), *[_p(x) for x in a], C.c_int(nv), _p(o_var), _p(o_var0), _p(o_sums))
    return {"var": o_var, "var0": o_var0, "sums": o_sums}


def cdfvertmean(dep, e31d, e3, mask, field, up, down, full=False, e3_t=None):
    """field (npt,npk,npj,npi) -> (npt,npj,npi) r4 REAL(dvertmean)."""
    field = _f(field)
    npt, npk, npj, npi = field.shape
    o = np.zeros((npt, 1, npj, npi), np.float32)
    e3, vvl = _e3(e3, e3_t, npt)
    a = [_f(dep), _f(e31d), e3, _m(mask), field]
    load().lift_cdfvertmean(*_i(npi, npj, npk, npt, full, vvl), C.c_float(up), C.c_float(down), *[_p(x) for x in a], _p(o))
    return o[:, 0]


def cdfvint(gdepw, e31d, tmask, e3t, field, full=False, gsop=False, occi=False, mean=False, sclf=1.0, e3t_t=None):
    """field (npt,npk,npj,npi) -> (npt,npko,npj,npi) r4 as handed to putvar per output depth."""
    field = _f(field)
    npt, npk, npj, npi = field.shape
    npko = 7 if gsop else 3 if occi else npk
    o = np.zeros((npt, npko + 1, npj, npi), np.float32)
    e3t, vvl = _e3(e3t, e3t_t, npt)
    a = [_f(gdepw), _f(e31d), _m(tmask), e3t, field]
    load().lift_cdfvint(*_i(npi, npj, npk, npt, full, gsop, occi, mean, vvl), C.c_float(sclf), *[_p(x) for x in a], C.c_int(npko), _p(o))
    return o[:, :npko]


def _zonal_args(g, e1, e2, phi, dep, mask, field):
    return [_f(e1), _f(e2), _f(phi), _f(dep), _m(mask), _m(g["tmaskatl"]), _m(g["tmaskind"]), _m(g["tmaskpac"]), field]


def cdfzonalsum(g, e1, e2, phi, dep, mask, field, nb=5, pdeg=False):
    """field (npt,npk,npj,npi) -> var (npt,nb,npk,npj) r4, r8 (npt,nb,npk,npj)."""
    field = _f(field)
    npt, npk, npj, npi = field.shape
    o_var, o_r8 = np.zeros((npt, nb, npk, npj), np.float32), np.zeros((npt, nb, npk, npj), np.float64)
    load().lift_cdfzonalsum(*_i(npi, npj, npk, npt, nb, pdeg), *[_p(x) for x in _zonal_args(g, e1, e2, phi, dep, mask, field)],
                            _p(o_var), _p(o_r8))
    return {"var": o_var, "r8": o_r8}


def cdfzonalmean(g, e1, e2, phi, dep, mask, field, nb=5, lmax=False):
    """-> var (npt,3*nb,npk,npj) r4 [mean, max, min blocks of nb], r8 mean and area (npt,nb,npk,npj)."""
    field = _f(field)
    npt, npk, npj, npi = field.shape
    o_var = np.zeros((npt, 3 * nb, npk, npj), np.float32)
    o_r8, o_area = np.zeros((npt, nb, npk, npj), np.float64), np.zeros((npt, nb, npk, npj), np.float64)
    load().lift_cdfzonalmean(*_i(npi, npj, npk, npt, nb, lmax), *[_p(x) for x in _zonal_args(g, e1, e2, phi, dep, mask, field)],
                             _p(o_var), _p(o_r8), _p(o_area))
    return {"var": o_var, "r8": o_r8, "area": o_area}


def cdfmocsig(g, zv, zt, zs, nbins, sigmin, sigstp, pref=0.0, nb=5, full=False, isodep=False, eiv=None, ntr=False, teos10=False,
              spv=(0.0, 0.0, 0.0), e3v_t=None):
    """zv, zt, zs (npt,npk,npj,npi) -> var (npt,2*nb,nbins,npj) r4, dmoc and depi (npt,npj,nbins,nb) r8."""
    zv = _f(zv)
    npt, npk, npj, npi = zv.shape
    o_var = np.zeros((npt, 2 * nb, nbins, npj), np.float32)
    o_dmoc, o_depi = np.zeros((npt, npj, nbins, nb), np.float64), np.zeros((npt, npj, nbins, nb), np.float64)
    ve = _f(eiv) if eiv is not None else _dummy(*zv.shape)
    e3v, vvl = _e3(g["e3v_0"], e3v_t, npt)
    a = [_f(g["e1v"]), _f(g["e3t_1d"]), _f(g["gdept_1d"]), _m(g["vmask"][0]), _m(g["tmaskatl"]), _m(g["tmaskind"]), _m(g["tmaskpac"]),
         e3v, zv, ve, _f(zt), _f(zs)]
    load().lift_cdfmocsig(*_i(npi, npj, npk, npt, nb, full, isodep, eiv is not None, ntr, teos10, nbins, vvl), C.c_float(sigmin), C.c_float(sigstp),
                          C.c_float(pref), C.c_float(spv[0]), C.c_float(spv[1]), C.c_float(spv[2]), *[_p(x) for x in a], _p(o_var), _p(o_dmoc),
                          _p(o_depi))
    return {"var": o_var, "dmoc": o_dmoc, "depi": o_depi}


def cdfsig0(zt, zs, sps=0.0, teos10=False):
    """zt, zs (npt,npk,npj,npi) r4 -> zsig0 (npt,npk,npj,npi) r4 as handed to putvar (src/cdfsig0.f90:139-155)."""
    zt, zs = _f(zt), _f(zs)
    npt, npk, npj, npi = zt.shape
    o = np.zeros(zt.shape, np.float32)
    load().lift_cdfsig0(*_i(npi, npj, npk, npt, teos10), C.c_float(sps), _p(zt), _p(zs), _p(o))
    return o


def cdfsigi(zt, zs, pref, spval=0.0, teos10=False):
    """-> zsigi (npt,npk,npj,npi) r4 as handed to putvar (src/cdfsigi.f90:130-145)."""
    zt, zs = _f(zt), _f(zs)
    npt, npk, npj, npi = zt.shape
    o = np.zeros(zt.shape, np.float32)
    load().lift_cdfsigi(*_i(npi, npj, npk, npt, teos10), C.c_float(pref), C.c_float(spval), _p(zt), _p(zs), _p(o))
    return o


def cdfmoc_rapid(gphiv, e1v, gdepw, gdept, e31d, tmask, e3v, v, taux, t, s, npigs, full=False, e3v_t=None):
    """The Rapid section alone (src/cdfmoc.f90:598-1004): gphiv, e1v (ni); tmask, e3v (npk,ni); v, t, s (npt,npk,ni); taux (npt,ni)
    -> what putvar is handed, (npt,npk,33) r4 (variable 1 has every level, the others level 1)."""
    v = _f(v)
    npt, npk, ni = v.shape
    o = np.zeros((npt, npk, 33), np.float32)
    e3v, vvl = _e3(e3v, e3v_t, npt)
    a = [_f(gphiv), _f(e31d), _f(e1v), _f(gdepw), _f(np.asarray(tmask, np.float32)), _f(gdept), e3v, v, _f(taux), _f(t), _f(s)]
    load().lift_cdfmoc_rapid(*_i(ni, npigs, npk, npt, full, vvl), *[_p(x) for x in a], _p(o))
    return o


def cdfmaxmoc(depw8, lat, moc, latmin, latmax, depmin, depmax):
    """depw8 (npk) r8 as stored in moc.nc (negative), lat (npj), moc (npk,npj) r4 -> the six numbers handed to putvar0d."""
    moc = _f(moc)
    npk, npj = moc.shape
    d8 = np.ascontiguousarray(depw8, np.float64)
    o = np.zeros(6, np.float32)
    load().lift_cdfmaxmoc(*_i(npj, npk), C.c_float(latmin), C.c_float(latmax), C.c_float(depmin), C.c_float(depmax), _p(d8), _p(_f(lat)),
                          _p(moc), _p(o))
    return o
