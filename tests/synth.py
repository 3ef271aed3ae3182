"""Synthetic NEMO C-grid configurations for the parity tests (SURVEY.md section 8d recipe).

All arrays are C-contiguous with reversed (Fortran) dimensions: 2-D (npj,npi), 3-D (npk,npj,npi),
records (npt,npk,npj,npi).  Small grids are made with NumPy on the host; make_grid_torch builds the
same kind of configuration directly in HBM for the full-size property tests.
"""
import numpy as np


def e3_profile(npk):
    k = np.arange(npk, dtype=np.float64)
    return (1.0 + 199.0 * 0.5 * (1.0 + np.tanh((k - 0.6 * npk) / (0.15 * npk + 1.0)))).astype(np.float32)


def make_grid(npi, npj, npk, seed=20240601, land=0.35, nbasins=5):
    rng = np.random.default_rng(seed)
    f32 = np.float32
    g = {"npi": npi, "npj": npj, "npk": npk}

    def metric():
        return (f32(2000.0) + f32(26000.0) * rng.random((npj, npi), dtype=f32)).astype(f32)

    for nm in ("e1t", "e2t", "e1u", "e2u", "e1v", "e2v", "e1f", "e2f"):
        g[nm] = metric()
    e31d = e3_profile(npk)
    g["e3t_1d"] = e31d
    g["gdepw_1d"] = np.concatenate([[0.0], np.cumsum(e31d[:-1], dtype=np.float64)]).astype(f32)
    g["gdept_1d"] = (g["gdepw_1d"] + 0.5 * e31d).astype(f32)

    u = rng.random((npj, npi))
    mbathy = np.where(u < land, 0, 1 + np.floor((npk - 1) * rng.random((npj, npi))).astype(np.int64))
    mbathy = np.minimum(mbathy, npk - 1)          # level npk is always below the bottom
    mbathy[0, :] = 0                              # jj=1 is land on global grids (src/cdfpsi.f90:394)
    k = np.arange(npk)[:, None, None]
    tmask = (k < mbathy[None]).astype(np.int8)
    umask = np.zeros_like(tmask)
    vmask = np.zeros_like(tmask)
    fmask = np.zeros_like(tmask)
    umask[:, :, :-1] = tmask[:, :, :-1] * tmask[:, :, 1:]
    vmask[:, :-1, :] = tmask[:, :-1, :] * tmask[:, 1:, :]
    fmask[:, :-1, :-1] = tmask[:, :-1, :-1] * tmask[:, :-1, 1:] * tmask[:, 1:, :-1] * tmask[:, 1:, 1:]
    g.update(tmask=tmask, umask=umask, vmask=vmask, fmask=fmask, mbathy=mbathy)

    # partial steps: the bottom wet cell is thinner
    e3t = np.broadcast_to(e31d[:, None, None], (npk, npj, npi)).astype(f32).copy()
    frac = (f32(0.2) + f32(0.8) * rng.random((npj, npi), dtype=f32)).astype(f32)
    jj, ii = np.nonzero(mbathy > 0)
    e3t[mbathy[jj, ii] - 1, jj, ii] *= frac[jj, ii]
    e3u = e3t.copy()
    e3v = e3t.copy()
    e3u[:, :, :-1] = np.minimum(e3t[:, :, :-1], e3t[:, :, 1:])
    e3v[:, :-1, :] = np.minimum(e3t[:, :-1, :], e3t[:, 1:, :])
    g.update(e3t_0=e3t, e3u_0=e3u, e3v_0=e3v)

    # three disjoint i-bands x tmask level 1 (new_maskglo.nc: tmaskatl / tmaskpac / tmaskind)
    band = (np.arange(npi) * 3) // max(npi, 1)
    t1 = tmask[0]
    g["tmaskatl"] = (t1 * (band == 0)[None, :]).astype(np.int8)
    g["tmaskpac"] = (t1 * (band == 1)[None, :]).astype(np.int8)
    g["tmaskind"] = (t1 * (band == 2)[None, :]).astype(np.int8)

    g["gphiv"] = np.broadcast_to(np.linspace(-78.0, 89.9, npj, dtype=f32)[:, None], (npj, npi)).copy()
    g["gphif"] = g["gphiv"].copy()
    lam = np.linspace(-180.0, 180.0, max(npi - 2, 1), endpoint=False, dtype=f32)
    glamf = np.empty(npi, f32)
    glamf[: lam.size] = lam
    if npi >= 3:
        glamf[npi - 2:] = lam[:2] if lam.size >= 2 else lam[0]   # 2-column E-W wrap: glamf(1)==glamf(npi-1)
    g["glamf"] = np.broadcast_to(glamf[None, :], (npj, npi)).copy()
    g["nbasins"] = nbasins
    return g


def moc_inputs(g, nbasins=None):
    """What src/cdfmoc.f90:306-336,:343-348 hands to its hot loop: e1v, masked e3v, ibmask."""
    nb = g["nbasins"] if nbasins is None else nbasins
    npj, npi = g["e1v"].shape
    e3v = (g["e3v_0"] * g["vmask"].astype(np.float32)).astype(np.float32)   # get_e3v, :576-596
    ib = np.zeros((npj, npi, nb), np.int16)
    ib[:, :, 0] = g["vmask"][0]                                             # npglo = vmask level 1
    if nb >= 5:                                                             # 1 glo 2 atl 3 inp 4 ind 5 pac
        ib[:, :, 1] = g["tmaskatl"]
        ib[:, :, 3] = g["tmaskind"]
        ib[:, :, 4] = g["tmaskpac"]
        ib[:, :, 2] = np.minimum(ib[:, :, 4] + ib[:, :, 3], 1)
    for b in range(5, nb):                                                  # extra synthetic basins
        ib[:, :, b] = g["tmask"][0] * ((np.arange(npi) + b) % 2)[None, :]
    ib[:, 0, 0] = 0                                                         # :335-336
    ib[:, npi - 1, 0] = 0
    return g["e1v"], e3v, ib


def velocity(g, npt, seed=1, which="v"):
    rng = np.random.default_rng(seed)
    npk, npj, npi = g["tmask"].shape
    mask = g["vmask" if which == "v" else "umask"].astype(np.float32)
    z = (rng.random((npt, npk, npj, npi), dtype=np.float32) - np.float32(0.5)) * mask[None]
    return z.astype(np.float32)


def numpy_moc(e1v, e3v, ibmask, zv):
    """Independent NumPy restatement of src/cdfmoc.f90:352-388 (sequential r8 sums via cumsum),
    used to cross-check the C oracle bit for bit on small grids."""
    zv = zv.reshape((-1,) + e3v.shape)
    npt = zv.shape[0]
    npk, npj, npi = e3v.shape
    nb = ibmask.shape[-1]
    out = np.zeros((npt, npk, npj, nb), np.float64)
    w = (e1v[None] * e3v).astype(np.float32)                       # e1v*e3v in r4
    for b in range(nb):
        wm = (w * ibmask[None, :, :, b].astype(np.float32)).astype(np.float32)
        for t in range(npt):
            p = (wm * zv[t]).astype(np.float32).astype(np.float64)  # r4 product, then *1.d0
            # dmoc = dmoc - p, sequentially along i, starting from 0
            s = np.cumsum(-p, axis=-1)[..., -1] if npi > 0 else np.zeros((npk, npj))
            s[npk - 1] = 0.0                                         # jk = 1, npk-1 only
            out[t, :, :, b] = s
    for jk in range(npk - 2, -1, -1):
        out[:, jk] = out[:, jk + 1] + out[:, jk] / 1.0e6
    return out
