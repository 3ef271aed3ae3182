"""Host-side mirror of the C ABI for Python callers (tests, bench.py, smoke).

The product is libcdftools_hip.so behind include/cdftools_hip.h; the drop-in host programs are
Fortran (cdftools_amd/fortran/).  This module only gives Python the same calls with array
arguments, so that the parity tests read like the reference loops they check.

Array convention: the reference is Fortran, column-major.  A NumPy / torch array handed to these
wrappers is C-contiguous with the dimensions REVERSED, e.g. the reference's zv(npiglo,npjglo,npk)
per record is shape (npt, npk, npjglo, npiglo) here, ibmask(nbasins,npiglo,npjglo) is
(npjglo, npiglo, nbasins) and dmoc(nbasins,npjglo,npk) per record is (npt, npk, npjglo, nbasins).
NumPy arrays are host memory (the library stages them through HBM); torch CUDA tensors are used in
place (no copy).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MEM_DEVICE, MEM_HOST, MEM_PINNED, CdftError, check

try:  # torch is plumbing for device memory / streams only
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _arg(x, dtype, name):
    """-> (pointer, memspace, keepalive) for a NumPy array (host) or torch CUDA tensor (device)."""
    if _is_tensor(x):
        want = {np.float32: torch.float32, np.float64: torch.float64, np.int16: torch.int16,
                np.uint8: torch.uint8, np.int8: torch.int8}[dtype]
        if x.dtype != want:
            raise TypeError(f"{name}: expected {want}, got {x.dtype}")
        if not x.is_contiguous():
            raise ValueError(f"{name}: tensor must be contiguous")
        if x.is_cuda:
            return C.c_void_p(x.data_ptr()), MEM_DEVICE, x
        a = x.numpy()
        return C.c_void_p(a.ctypes.data), MEM_HOST, a
    a = np.ascontiguousarray(x, dtype=dtype)
    return C.c_void_p(a.ctypes.data), MEM_HOST, a


def current_stream_ptr():
    """hipStream_t of torch's current stream (so torch.cuda.Event brackets see the kernels)."""
    if torch is not None and torch.cuda.is_available():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(0)


def device_count() -> int:
    n = C.c_int(0)
    check(_lib.load().cdft_device_count(C.byref(n)))
    return n.value


class Timer:
    """HIP events on a given stream (cdft_timer_*)."""

    def __init__(self):
        self._h = C.c_void_p()
        check(_lib.load().cdft_timer_create(C.byref(self._h)))

    def start(self, stream=None):
        check(_lib.load().cdft_timer_start(self._h, stream or current_stream_ptr()))

    def stop(self, stream=None):
        check(_lib.load().cdft_timer_stop(self._h, stream or current_stream_ptr()))

    def elapsed_ms(self) -> float:
        ms = C.c_float(0)
        check(_lib.load().cdft_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            _lib.load().cdft_timer_destroy(self._h)
        except Exception:
            pass


COLLECTIVE_PEER, COLLECTIVE_RCCL = 0, 1


def set_collective(mode):
    """How the blocks of the sharded plans made from now on exchange data: COLLECTIVE_PEER (default) or COLLECTIVE_RCCL."""
    check(_lib.load().cdft_shard_set_collective(int(mode)))


def get_collective():
    m = C.c_int(0)
    check(_lib.load().cdft_shard_get_collective(C.byref(m)))
    return m.value


class PinnedArray:
    """Page-locked host memory from cdft_host_alloc seen as a NumPy array (`.a`); copies from / to it are asynchronous."""

    def __init__(self, shape, dtype):
        self._p = C.c_void_p()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        check(_lib.load().cdft_host_alloc(C.byref(self._p), n))
        self.a = np.frombuffer((C.c_char * n).from_address(self._p.value), dtype=dtype).reshape(shape)

    def close(self):
        if self._p:
            self.a = None
            _lib.load().cdft_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_from_env():
    """CDFT_GPU_MAP / CDFT_GPUS as the drop-in programs read them (cdft_shard_from_env) -> list of device numbers,
    one per row block ([-1] = one block on the current device: no sharding)."""
    n = C.c_int(0)
    dev = (C.c_int * 64)()
    check(_lib.load().cdft_shard_from_env(C.byref(n), dev, 64))
    return [dev[i] for i in range(n.value)]


def _devs(devices):
    arr = (C.c_int * len(devices))(*[int(d) for d in devices])
    return len(devices), arr


class MocPlan:
    """cdfmoc hot path (src/cdfmoc.f90:325-388): cdft_moc_* of the C ABI.  devices=[d0, d1, ..]: a sharded plan, one
    row block per entry (block b on HIP device d_b); the arrays handed to it are the full domain's."""

    def __init__(self, npiglo: int, npjglo: int, npk: int, nbasins: int, devices=None):
        self.npi, self.npj, self.npk, self.nb = npiglo, npjglo, npk, nbasins
        self._h = C.c_void_p()
        if devices is None:
            check(_lib.load().cdft_moc_create(C.byref(self._h), npiglo, npjglo, npk, nbasins))
        else:
            check(_lib.load().cdft_moc_create_sharded(C.byref(self._h), npiglo, npjglo, npk, nbasins, *_devs(devices)))

    def set_mesh(self, e1v, e3v, ibmask, vmask=None, stream=None):
        """e1v (npj,npi) r4; e3v (npk,npj,npi) r4 already * vmask -- or the raw metric with vmask (npk,npj,npi) i1 given,
        the product then being formed on the device; ibmask (npj,npi,nb) i2 in {0,1}."""
        if vmask is not None:
            p1, s1, k1 = _arg(e1v, np.float32, "e1v")
            p2, s2, k2 = _arg(e3v, np.float32, "e3v")
            p3, s3, k3 = _arg(ibmask, np.int16, "ibmask")
            p4, s4, k4 = _arg(vmask, np.int8, "vmask")
            if not (s1 == s2 == s3 == s4):
                raise ValueError("e1v, e3v, vmask and ibmask must live in the same memory space")
            check(_lib.load().cdft_moc_set_mesh_vmask(self._h, p1, p2, p4, p3, s1, stream or current_stream_ptr()))
            return
        p1, s1, k1 = _arg(e1v, np.float32, "e1v")
        p2, s2, k2 = _arg(e3v, np.float32, "e3v")
        p3, s3, k3 = _arg(ibmask, np.int16, "ibmask")
        if not (s1 == s2 == s3):
            raise ValueError("e1v, e3v and ibmask must live in the same memory space")
        for k, n, nm in ((k1, self.npi * self.npj, "e1v"), (k2, self.npi * self.npj * self.npk, "e3v"),
                         (k3, self.npi * self.npj * self.nb, "ibmask")):
            if int(np.prod(k.shape)) != n:
                raise ValueError(f"{nm}: expected {n} elements, got shape {tuple(k.shape)}")
        check(_lib.load().cdft_moc_set_mesh(self._h, p1, p2, p3, s1, stream or current_stream_ptr()))

    def set_tuning(self, waves_per_row=0, vec_per_lane=0, mode=0):
        check(_lib.load().cdft_moc_set_tuning(self._h, waves_per_row, vec_per_lane, mode))

    def decomp_set_mesh(self, e3v, e1u, zcoef, gdept, tmask, umask, stream=None):
        """-decomp mesh (host arrays): e3v (npk,npj,npi) r4 masked, e1u / zcoef (npj,npi) r4, gdept (npk) r4,
        tmask / umask (npk,npj,npi) bytes 0/1."""
        keep = [np.ascontiguousarray(e3v, np.float32), np.ascontiguousarray(e1u, np.float32),
                np.ascontiguousarray(zcoef, np.float32), np.ascontiguousarray(gdept, np.float32),
                np.ascontiguousarray(tmask, np.uint8), np.ascontiguousarray(umask, np.uint8)]
        n2, n3 = self.npi * self.npj, self.npi * self.npj * self.npk
        for a, n, nm in zip(keep, (n3, n2, n2, self.npk, n3, n3), ("e3v", "e1u", "zcoef", "gdept", "tmask", "umask")):
            if a.size != n:
                raise ValueError(f"{nm}: expected {n} elements, got shape {a.shape}")
        check(_lib.load().cdft_moc_decomp_set_mesh(self._h, *[C.c_void_p(a.ctypes.data) for a in keep], MEM_HOST,
                                                   stream or current_stream_ptr()))

    def decomp(self, zv, ztemp, zsal, dmoc_sh=None, eos=0, stream=None):
        """One record (host arrays (npk,npj,npi) r4) -> dict of (npk,npj,nb) r8: total, sh, bt, ag.
        dmoc_sh: the values the shear part accumulates onto (None = zeros, i.e. the first record)."""
        ins = [np.ascontiguousarray(x, np.float32) for x in (zv, ztemp, zsal)]
        shape = (self.npk, self.npj, self.nb)
        out = {k: np.zeros(shape, np.float64) for k in ("total", "sh", "bt", "ag")}
        if dmoc_sh is not None:
            out["sh"][...] = dmoc_sh
        check(_lib.load().cdft_moc_decomp(self._h, *[C.c_void_p(a.ctypes.data) for a in ins], eos, MEM_HOST,
                                          *[C.c_void_p(out[k].ctypes.data) for k in ("total", "sh", "bt", "ag")], MEM_HOST,
                                          stream or current_stream_ptr()))
        return out

    def describe(self) -> str:
        buf = C.create_string_buffer(256)
        check(_lib.load().cdft_moc_describe(self._h, buf, 256))
        return buf.value.decode()

    def profile_enable(self, on=True):
        check(_lib.load().cdft_moc_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """-> (summed zonal-kernel duration in ms, number of launches) since the last read."""
        ms, n = C.c_double(0), C.c_int(0)
        check(_lib.load().cdft_moc_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_read_each(self, capacity=65536):
        """-> array of the zonal-kernel durations (ms) of every launch since the last read."""
        buf = (C.c_float * capacity)()
        n = C.c_int(0)
        check(_lib.load().cdft_moc_profile_read_each(self._h, buf, capacity, C.byref(n)))
        return np.frombuffer(buf, dtype=np.float32, count=min(n.value, capacity)).astype(np.float64)

    def profile_read_block(self, block, capacity=65536):
        """-> the zonal-kernel durations (ms) of the launches of ONE row block since the last read (a sharded plan's
        profile_read / profile_read_each give the slowest block of every launch)."""
        buf = (C.c_float * capacity)()
        n = C.c_int(0)
        check(_lib.load().cdft_moc_profile_read_block(self._h, block, buf, capacity, C.byref(n)))
        return np.frombuffer(buf, dtype=np.float32, count=min(n.value, capacity)).astype(np.float64)

    def block_info(self, block):
        """-> (device, j0, j1): the device of a row block and the rows [j0,j1) of the domain it owns."""
        d, j0, j1 = C.c_int(0), C.c_int(0), C.c_int(0)
        check(_lib.load().cdft_moc_block_info(self._h, block, C.byref(d), C.byref(j0), C.byref(j1)))
        return d.value, j0.value, j1.value

    def reserve(self, npt, stream=None):
        """cdft_moc_reserve: the pool for npt staged records, made now (before anything else is allocated in HBM)."""
        check(_lib.load().cdft_moc_reserve(self._h, int(npt), stream or current_stream_ptr()))

    def stage(self, zv, rec_stride=None, npt=None, stream=None):
        """Hand the records over once (cdft_moc_stage): a sharded plan moves every block's rows into that block's HBM;
        compute_staged then runs on them any number of times without touching the inputs again."""
        pz, sz, kz = _arg(zv, np.float32, "zv")
        n3 = self.npi * self.npj * self.npk
        if npt is None:
            npt = int(np.prod(kz.shape)) // n3
        self._staged = (kz, npt)                          # a single-device plan borrows a device array: keep it alive
        check(_lib.load().cdft_moc_stage(self._h, pz, rec_stride or n3, npt, sz, stream or current_stream_ptr()))

    def stage_rows(self, zv, j0, npj_full, stream=None):
        """cdft_moc_stage_rows: zv (npt,npk,npj_full,npi) r4 of a LARGER domain, this plan's rows being [j0, j0+npj) of it."""
        pz, sz, kz = _arg(zv, np.float32, "zv")
        lev = self.npi * npj_full
        npt = int(np.prod(kz.shape)) // (lev * self.npk)
        self._staged = (kz, npt)
        base = C.c_void_p(pz.value + 4 * self.npi * j0)
        check(_lib.load().cdft_moc_stage_rows(self._h, base, lev * self.npk, lev, npt, sz, stream or current_stream_ptr()))

    # ---- the stream route (cdft_moc_stream_*): a window of levels at a time, nothing 3-D resident
    def set_basins(self, e1v, ibmask, stream=None):
        p1, s1, k1 = _arg(e1v, np.float32, "e1v")
        pm, sm, km = _arg(ibmask, np.int16, "ibmask")
        if s1 != sm:
            raise ValueError("e1v and ibmask must live in the same memory space")
        self._mesh_keep = (k1, km)
        check(_lib.load().cdft_moc_set_basins(self._h, p1, pm, s1, stream or current_stream_ptr()))

    def stream_begin(self, npt, nlev, stream=None):
        check(_lib.load().cdft_moc_stream_begin(self._h, int(npt), int(nlev), stream or current_stream_ptr()))

    def stream_dest(self, slot, level):
        """-> (device address of row 1 of `level` of the window in slot `slot` (0: metric, 1..npt: records), bytes between rows)"""
        d, pitch = C.c_void_p(), C.c_size_t(0)
        check(_lib.load().cdft_moc_stream_dest(self._h, int(slot), int(level), C.byref(d), C.byref(pitch)))
        return d.value, pitch.value

    def stream_vmask_dest(self, level):
        d = C.c_void_p()
        check(_lib.load().cdft_moc_stream_vmask_dest(self._h, int(level), C.byref(d)))
        return d.value

    def stream_compute(self, k0, nlev, with_vmask=True, stream=None):
        check(_lib.load().cdft_moc_stream_compute(self._h, int(k0), int(nlev), 1 if with_vmask else 0, stream or current_stream_ptr()))

    def stream_end(self, out, stream=None):
        po, so, ko = _arg(out, np.float64, "dmoc")
        check(_lib.load().cdft_moc_stream_end(self._h, po, so, stream or current_stream_ptr()))
        return out

    def stage_open(self, npt, stream=None):
        """cdft_moc_stage_open: the pool laid out for npt records the caller will write itself (weights brought in)."""
        check(_lib.load().cdft_moc_stage_open(self._h, int(npt), stream or current_stream_ptr()))

    def stage_dest(self, record, level):
        """cdft_moc_stage_dest -> (device address of row 1 of `level` of `record` (both 0-based), bytes between rows)."""
        d, pitch = C.c_void_p(), C.c_size_t(0)
        check(_lib.load().cdft_moc_stage_dest(self._h, int(record), int(level), C.byref(d), C.byref(pitch)))
        return d.value, pitch.value

    def stage_close(self):
        check(_lib.load().cdft_moc_stage_close(self._h))

    def release_staged(self):
        """cdft_moc_release_staged: forget the staged records and give the pool's memory back."""
        self._staged = None
        check(_lib.load().cdft_moc_release_staged(self._h))

    def stats(self):
        """cdft_moc_stats -> (zonal-kernel launches so far, records they covered)."""
        nl, nr = C.c_int64(0), C.c_int64(0)
        check(_lib.load().cdft_moc_stats(self._h, C.byref(nl), C.byref(nr)))
        return nl.value, nr.value

    def compute_pitched(self, zv_ptr, rec_stride, row_stride, npt, out, stream=None):
        """Records in a pitched layout in HBM (cdft_moc_compute_pitched): zv_ptr a device address (int), strides in elements."""
        po, so, ko = _arg(out, np.float64, "dmoc")
        check(_lib.load().cdft_moc_compute_pitched(self._h, C.c_void_p(int(zv_ptr)), int(rec_stride), int(row_stride), int(npt), po, so,
                                                   stream or current_stream_ptr()))
        return out

    def compute_staged(self, out, stream=None):
        """dmoc (npt,npk,npj,nb) r8 of the staged records into `out` (NumPy array or CUDA tensor)."""
        po, so, ko = _arg(out, np.float64, "dmoc")
        check(_lib.load().cdft_moc_compute_staged(self._h, po, so, stream or current_stream_ptr()))
        return out

    def compute_staged_windows(self, nlev, out, stream=None):
        """cdft_moc_compute_staged_windows: the staged pool, one launch per window of nlev levels (the stream route's launches)."""
        po, so, ko = _arg(out, np.float64, "dmoc")
        check(_lib.load().cdft_moc_compute_staged_windows(self._h, int(nlev), po, so, stream or current_stream_ptr()))
        return out

    def compute(self, zv, out=None, rec_stride=None, npt=None, stream=None):
        """zv (npt,npk,npj,npi) r4 -> dmoc (npt,npk,npj,nb) r8 in Sv (same memory space as zv
        unless `out` is given)."""
        pz, sz, kz = _arg(zv, np.float32, "zv")
        n3 = self.npi * self.npj * self.npk
        if npt is None:
            total = int(np.prod(kz.shape))
            if total % n3:
                raise ValueError(f"zv: {total} elements is not a whole number of records of {n3}")
            npt = total // n3
        if rec_stride is None:
            rec_stride = n3
        shape = (npt, self.npk, self.npj, self.nb)
        if out is None:
            if sz == MEM_DEVICE:
                out = torch.empty(shape, dtype=torch.float64, device=zv.device)
            else:
                out = np.empty(shape, dtype=np.float64)
        po, so, ko = _arg(out, np.float64, "dmoc")
        if int(np.prod(ko.shape)) != int(np.prod(shape)):
            raise ValueError(f"dmoc: expected shape {shape}, got {tuple(ko.shape)}")
        check(_lib.load().cdft_moc_compute(self._h, pz, rec_stride, npt, sz, po, so,
                                           stream or current_stream_ptr()))
        return out

    def close(self):
        if self._h:
            _lib.load().cdft_moc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


EOS_80, EOS_TEOS10, EOS_JMD94, EOS_NTR = 0, 1, 2, 3
EOS_FAST_FMA = 1


def moc_mask_e3v(e3v, vmask, stream=None):
    """get_e3v's product (src/cdfmoc.f90:576-596) in place on device tensors: e3v (r4) *= vmask (int8), cdft_moc_mask_e3v."""
    import torch
    if not (_is_tensor(e3v) and _is_tensor(vmask) and e3v.is_cuda and vmask.is_cuda):
        raise TypeError("moc_mask_e3v: device tensors only (a host caller multiplies its own arrays)")
    if e3v.dtype != torch.float32 or vmask.dtype not in (torch.int8, torch.uint8) or not e3v.is_contiguous() or not vmask.is_contiguous():
        raise TypeError("moc_mask_e3v: e3v contiguous float32, vmask contiguous int8 / uint8")
    if e3v.numel() != vmask.numel():
        raise ValueError("moc_mask_e3v: %d cells of e3v, %d of vmask" % (e3v.numel(), vmask.numel()))
    check(_lib.load().cdft_moc_mask_e3v(C.c_void_p(e3v.data_ptr()), C.c_void_p(vmask.data_ptr()), e3v.numel(), stream or current_stream_ptr()))
    return e3v


def _out_like(x, dtype_np, dtype_t):
    if _is_tensor(x) and x.is_cuda:
        return torch.empty(x.shape, dtype=dtype_t, device=x.device)
    return np.empty(tuple(x.shape), dtype=dtype_np)


def sigmai(ptem, psal, pref=0.0, eos=EOS_80, flags=0, pref2d=None, out=None, stream=None):
    """sigmai(ptem, psal, pref) -> r8 (src/eos.f90:802-887; pref2d: :1002-1087; sigma0 = pref 0)."""
    pt, st_, kt = _arg(ptem, np.float32, "ptem")
    ps, ss, ks = _arg(psal, np.float32, "psal")
    n = int(np.prod(kt.shape))
    if ss != st_ or int(np.prod(ks.shape)) != n:
        raise ValueError("ptem and psal must have the same size and memory space")
    if out is None:
        out = _out_like(kt, np.float64, torch.float64 if torch is not None else None)
    po, so, ko = _arg(out, np.float64, "sig")
    if int(np.prod(ko.shape)) != n:
        raise ValueError("sig: wrong size")
    s = stream or current_stream_ptr()
    if pref2d is None:
        check(_lib.load().cdft_eos_sigmai(pt, ps, n, float(pref), eos, flags, po, st_, so, s))
    else:
        pp, sp, kp = _arg(pref2d, np.float32, "pref2d")
        if sp != st_ or int(np.prod(kp.shape)) != n:
            raise ValueError("pref2d must match ptem in size and memory space")
        check(_lib.load().cdft_eos_sigmai2d(pt, ps, pp, n, eos, flags, po, st_, so, s))
    return out


def sigma0(ptem, psal, eos=EOS_80, flags=0, out=None, stream=None):
    """sigma0 = sigmai at 0 m (src/eos.f90:601-632)."""
    return sigmai(ptem, psal, 0.0, eos, flags, None, out, stream)


EOS_MCD87 = 4
EOSQ_ALPHA, EOSQ_BETA, EOSQ_ALBET_MCD87, EOSQ_BETA_MCD87, EOSQ_SPICE_POL, EOSQ_SPICE_F02 = range(6)


def eos_field(quantity, ptem, psal, pdep=0.0, eos=EOS_80, out=None, stream=None):
    """cdft_eos_field: alpha, beta, albet, beta (McDougall 1987) or spiciness at every point -> r8 (src/eos.f90:1315-1775,
    :689-800).  pdep: one depth, or an array like ptem with a depth per point (the profile members of the generics)."""
    pt, st_, kt = _arg(ptem, np.float32, "ptem")
    ps, ss, ks = _arg(psal, np.float32, "psal")
    n = int(np.prod(kt.shape))
    if ss != st_ or int(np.prod(ks.shape)) != n:
        raise ValueError("ptem and psal must have the same size and memory space")
    pp, dep = None, 0.0
    if np.ndim(pdep) == 0:
        dep = float(pdep)
    else:
        pp, sp, kp = _arg(pdep, np.float32, "pdep")
        if sp != st_ or int(np.prod(kp.shape)) != n:
            raise ValueError("pdep must match ptem in size and memory space")
    if out is None:
        out = _out_like(kt, np.float64, torch.float64 if torch is not None else None)
    po, so, ko = _arg(out, np.float64, "out")
    if int(np.prod(ko.shape)) != n:
        raise ValueError("out: wrong size")
    check(_lib.load().cdft_eos_field(int(quantity), pt, ps, pp, dep, n, eos, po, st_, so, stream or current_stream_ptr()))
    return out


def eos_point(quantity, ptem, psal, pdep, eos=EOS_80):
    """alpha_0d / beta_0d: one point, r8 arguments (src/eos.f90:1505-1543, :1596-1641)."""
    import ctypes
    out = ctypes.c_double()
    check(_lib.load().cdft_eos_point(int(quantity), float(ptem), float(psal), float(pdep), eos, ctypes.byref(out)))
    return out.value


def alpha(ptem, psal, pdep=0.0, eos=EOS_80, **kw):
    return eos_field(EOSQ_ALPHA, ptem, psal, pdep, eos, **kw)


def beta(ptem, psal, pdep=0.0, eos=EOS_80, **kw):
    """eos = EOS_MCD87: beta(..., ld_eos80=.TRUE.) (src/eos.f90:1693-1775)."""
    return eos_field(EOSQ_BETA_MCD87 if eos == EOS_MCD87 else EOSQ_BETA, ptem, psal, pdep, EOS_80 if eos == EOS_MCD87 else eos, **kw)


def albet(ptem, psal, pdep=0.0, **kw):
    return eos_field(EOSQ_ALBET_MCD87, ptem, psal, pdep, **kw)


def spice(ptem, psal, flament=False, **kw):
    """spice(ptem, psal, kpi, kpj, ld_teos10) -- the GSW fit -- or, flament=True, spice(ptem, psal, kpi, kpj)."""
    return eos_field(EOSQ_SPICE_F02 if flament else EOSQ_SPICE_POL, ptem, psal, 0.0, **kw)


def eosbn2(ptem_up, ptem_down, psal_up, psal_down, pe3w, pdep, eos=EOS_80, out=None, stream=None):
    """cdft_eos_bn2: N^2 between two T levels -> r4 (src/eos.f90:1203-1312); eos = EOS_MCD87 for the ld_eos80 member."""
    a = [_arg(x, np.float32, nm) for x, nm in zip((ptem_up, ptem_down, psal_up, psal_down, pe3w), ("ptem_up", "ptem_down", "psal_up", "psal_down", "pe3w"))]
    n = int(np.prod(a[0][2].shape))
    if any(sp != a[0][1] or int(np.prod(k.shape)) != n for _, sp, k in a):
        raise ValueError("the five arrays must have the same size and memory space")
    if out is None:
        out = _out_like(a[0][2], np.float32, torch.float32 if torch is not None else None)
    po, so, ko = _arg(out, np.float32, "bn2")
    if int(np.prod(ko.shape)) != n:
        raise ValueError("bn2: wrong size")
    check(_lib.load().cdft_eos_bn2(*[p for p, _, _ in a], float(pdep), n, eos, po, a[0][1], so, stream or current_stream_ptr()))
    return out


def sig_tool(ptem, psal, pref=0.0, spval=0.0, eos=EOS_80, flags=0, out=None, stream=None):
    """cdfsig0 / cdfsigi level loop: r4 = REAL(sigmai * (S /= spval)) (src/cdfsig0.f90:146-152)."""
    pt, st_, kt = _arg(ptem, np.float32, "ptem")
    ps, ss, ks = _arg(psal, np.float32, "psal")
    n = int(np.prod(kt.shape))
    if ss != st_ or int(np.prod(ks.shape)) != n:
        raise ValueError("ptem and psal must have the same size and memory space")
    if out is None:
        out = _out_like(kt, np.float32, torch.float32 if torch is not None else None)
    po, so, ko = _arg(out, np.float32, "sig4")
    check(_lib.load().cdft_eos_sig_tool(pt, ps, n, float(pref), float(spval), eos, flags, po, st_, so,
                                        stream or current_stream_ptr()))
    return out


class _Plan:
    """Common life-cycle of the opaque C plans."""

    _destroy = None

    def close(self):
        if getattr(self, "_h", None):
            getattr(_lib.load(), self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _null():
    return C.c_void_p(0)


class HeatcPlan(_Plan):
    """cdfheatc level sums (src/cdfheatc.f90:205-240): cdft_heatc_* of the C ABI."""

    _destroy = "cdft_heatc_destroy"

    def __init__(self, npiglo, npjglo, npk, devices=None):
        self.npi, self.npj, self.npk = npiglo, npjglo, npk
        self._h = C.c_void_p()
        if devices is None:
            check(_lib.load().cdft_heatc_create(C.byref(self._h), npiglo, npjglo, npk))
        else:
            check(_lib.load().cdft_heatc_create_sharded(C.byref(self._h), npiglo, npjglo, npk, *_devs(devices)))

    def set_mesh(self, e1t, e2t, tmask, e3t=None, e31d=None, stream=None):
        """e1t,e2t (npj,npi) r4; tmask (npk,npj,npi) uint8/int8 0-1 bytes or r4; e3t (npk,npj,npi) r4
        or e31d (npk) r4 (-full)."""
        p1, s1, k1 = _arg(e1t, np.float32, "e1t")
        p2, s2, k2 = _arg(e2t, np.float32, "e2t")
        is_r4 = (tmask.dtype == np.float32) if not _is_tensor(tmask) else (tmask.dtype == torch.float32)
        pm, sm, km = _arg(tmask, np.float32 if is_r4 else np.uint8, "tmask")
        p3, s3, k3 = _arg(e3t, np.float32, "e3t") if e3t is not None else (_null(), s1, None)
        p4, s4, k4 = _arg(e31d, np.float32, "e31d") if e31d is not None else (_null(), s1, None)
        if not (s1 == s2 == sm == s3 == s4):
            raise ValueError("all mesh arrays must live in the same memory space")
        check(_lib.load().cdft_heatc_set_mesh(self._h, p1, p2, p3, p4, pm, 1 if is_r4 else 0, s1,
                                              stream or current_stream_ptr()))

    def set_sum_mode(self, r4seq):
        """r4seq=True: the reference's sequential r4 SUMs, bit for bit (CDFT_HEATC_SUM_R4SEQ); False: r8 sums (default)."""
        check(_lib.load().cdft_heatc_set_sum_mode(self._h, 1 if r4seq else 0))

    def chain(self, carry_in=None, want_last=False):
        """row blocks in separate plans (r4seq): carry_in (npk,3) r4 starts the next compute's chains; want_last returns
        the (npk,3) r4 sums the last compute ended with."""
        last = np.empty((self.npk, 3), np.float32) if want_last else None
        keep = None if carry_in is None else np.ascontiguousarray(carry_in, np.float32)
        check(_lib.load().cdft_heatc_chain(self._h, None if keep is None else C.c_void_p(keep.ctypes.data),
                                           None if last is None else C.c_void_p(last.ctypes.data)))
        return last

    def compute(self, temp, rmxldep=None, gdepw=None, mxloption=0, stream=None, device_out=None):
        """temp (npk,npj,npi) r4 -> sums (npk,3) r8 = dsurf, dvol2d, dsum2d per level: a host array (the call returns
        when it is there), or -- device_out = a CUDA r8 tensor of 3*npk elements -- left in HBM, asynchronously."""
        pt, st_, kt = _arg(temp, np.float32, "temp")
        pm, sm, km = _arg(rmxldep, np.float32, "rmxldep") if rmxldep is not None else (_null(), st_, None)
        pg, sg, kg = _arg(np.asarray(gdepw, np.float32), np.float32, "gdepw") if gdepw is not None else (_null(), 0, None)
        if sm != st_:
            raise ValueError("rmxldep must live in the same memory space as temp")
        if device_out is not None:
            po, so, ko = _arg(device_out, np.float64, "sums")
            check(_lib.load().cdft_heatc_compute(self._h, pt, st_, pm, pg, mxloption, po, so, stream or current_stream_ptr()))
            return device_out
        out = np.empty((self.npk, 3), np.float64)
        check(_lib.load().cdft_heatc_compute(self._h, pt, st_, pm, pg, mxloption, C.c_void_p(out.ctypes.data),
                                             MEM_HOST, stream or current_stream_ptr()))
        return out


def _opt(x, name):
    """optional r4 array -> (pointer, space or None, keepalive)"""
    if x is None:
        return _null(), None, None
    return _arg(x, np.float32, name)


def _same_space(*spaces):
    s = {x for x in spaces if x is not None}
    if len(s) > 1:
        raise ValueError("all input arrays of one call must live in the same memory space")
    return s.pop() if s else MEM_HOST


class MhstPlan(_Plan):
    """cdfmhst (src/cdfmhst.f90:308-367): cdft_mhst_* of the C ABI."""

    _destroy = "cdft_mhst_destroy"

    def __init__(self, npiglo, npjglo, npk, nbasins, devices=None):
        self.npi, self.npj, self.npk, self.nb = npiglo, npjglo, npk, nbasins
        self._h = C.c_void_p()
        if devices is None:
            check(_lib.load().cdft_mhst_create(C.byref(self._h), npiglo, npjglo, npk, nbasins))
        else:
            check(_lib.load().cdft_mhst_create_sharded(C.byref(self._h), npiglo, npjglo, npk, nbasins, *_devs(devices)))

    def set_mesh(self, e1v, masks, e3v=None, e31d=None, stream=None):
        """e1v (npj,npi); masks (nb,npj,npi) r4, slab 0 global; e3v (npk,npj,npi) or e31d (npk)."""
        p1, s1, k1 = _arg(e1v, np.float32, "e1v")
        pm, sm, km = _arg(masks, np.float32, "masks")
        p3, s3, k3 = _opt(e3v, "e3v")
        p4, s4, k4 = _opt(e31d, "e31d")
        sp = _same_space(s1, sm, s3, s4)
        check(_lib.load().cdft_mhst_set_mesh(self._h, p1, p3, p4, pm, sp, stream or current_stream_ptr()))

    def compute(self, vt=None, vs=None, zv=None, zt=None, zs=None, zdim=False, stream=None, device_out=None):
        """-> (heat, salt) host arrays (npj,nb) or (npk,npj,nb) with zdim: raw zonal sums.  device_out = (heat, salt) CUDA
        tensors of that shape: the results stay in HBM and the call does not wait for them (the kernels alone)."""
        a = [_opt(x, n) for x, n in ((vt, "vt"), (vs, "vs"), (zv, "zv"), (zt, "zt"), (zs, "zs"))]
        sp = _same_space(*[x[1] for x in a])
        shape = (self.npk, self.npj, self.nb) if zdim else (self.npj, self.nb)
        if device_out is not None:
            dh, dsl = device_out
            check(_lib.load().cdft_mhst_compute(self._h, a[0][0], a[1][0], a[2][0], a[3][0], a[4][0], sp, 1 if zdim else 0,
                                                C.c_void_p(dh.data_ptr()), C.c_void_p(dsl.data_ptr()), MEM_DEVICE,
                                                stream or current_stream_ptr()))
            return dh, dsl
        heat, salt = np.empty(shape, np.float64), np.empty(shape, np.float64)
        check(_lib.load().cdft_mhst_compute(self._h, a[0][0], a[1][0], a[2][0], a[3][0], a[4][0], sp,
                                            1 if zdim else 0, C.c_void_p(heat.ctypes.data),
                                            C.c_void_p(salt.ctypes.data), MEM_HOST, stream or current_stream_ptr()))
        return heat, salt


class PsiPlan(_Plan):
    """cdfpsi (src/cdfpsi.f90:271-438, default / -V / -mean / -mask / -ref): cdft_psi_*."""

    _destroy = "cdft_psi_destroy"

    def __init__(self, npiglo, npjglo, npk, devices=None):
        self.npi, self.npj, self.npk = npiglo, npjglo, npk
        self._h = C.c_void_p()
        if devices is None:
            check(_lib.load().cdft_psi_create(C.byref(self._h), npiglo, npjglo, npk))
        else:
            check(_lib.load().cdft_psi_create_sharded(C.byref(self._h), npiglo, npjglo, npk, *_devs(devices)))

    # --- the stages used when the domain is row-block partitioned (cdftools_amd/shard.py) ---
    def integrate(self, zu=None, e2u=None, e3u=None, zv=None, e1v=None, e3v=None, e31d=None, stream=None):
        names = ("zu", "e2u", "e3u", "zv", "e1v", "e3v", "e31d")
        a = [_opt(x, n) for x, n in zip((zu, e2u, e3u, zv, e1v, e3v, e31d), names)]
        sp = _same_space(*[x[1] for x in a])
        check(_lib.load().cdft_psi_integrate(self._h, a[0][0], a[1][0], a[2][0], a[3][0], a[4][0], a[5][0], a[6][0],
                                             sp, stream or current_stream_ptr()))

    def scan_u(self, carry=None, first_block=True, stream=None, device_out=False):
        """j-scan of the U path; carry (npi,) r8 (host array or CUDA tensor) from the block to the south -> this
        block's last row (npi,) r8: a host array, or a CUDA tensor with device_out (the multi-GPU chain keeps both
        in HBM)."""
        if carry is None:
            pc, sc, keep = _null(), MEM_HOST, None
        else:
            pc, sc, keep = _arg(carry, np.float64, "carry")
            if int(np.prod(keep.shape)) != self.npi:
                raise ValueError(f"carry has shape {tuple(keep.shape)}, expected {(self.npi,)}")
        if device_out:
            last = torch.empty(self.npi, dtype=torch.float64, device="cuda")
            pl, sl = C.c_void_p(last.data_ptr()), MEM_DEVICE
        else:
            last = np.empty(self.npi, np.float64)
            pl, sl = C.c_void_p(last.ctypes.data), MEM_HOST
        check(_lib.load().cdft_psi_scan_u(self._h, pc, sc, 1 if first_block else 0, pl, sl, stream or current_stream_ptr()))
        return last

    def scan_open(self, lmean=False, stream=None):
        """-open: psi from the row npjglo-2 / column npiglo-2 (needs integrate() with both velocity sets)."""
        check(_lib.load().cdft_psi_scan_open(self._h, 1 if lmean else 0, stream or current_stream_ptr()))

    def value(self, ii, ij, stream=None):
        v = C.c_double(0)
        check(_lib.load().cdft_psi_value(self._h, ii, ij, C.byref(v), stream or current_stream_ptr()))
        return v.value

    def finish(self, ref_value=None, iiref=None, ijref=None, fmask=None, stream=None):
        pm, sm, km = _opt(fmask, "fmask")
        p4, d8 = np.empty((self.npj, self.npi), np.float32), np.empty((self.npj, self.npi), np.float64)
        use = ref_value is not None
        check(_lib.load().cdft_psi_finish(self._h, 1 if use else 0, float(ref_value) if use else 0.0,
                                          iiref or 1, ijref or 1, pm, MEM_HOST if sm is None else sm,
                                          C.c_void_p(p4.ctypes.data), C.c_void_p(d8.ctypes.data), MEM_HOST,
                                          stream or current_stream_ptr()))
        return p4, d8

    def compute(self, zu=None, e2u=None, e3u=None, zv=None, e1v=None, e3v=None, e31d=None, iiref=None,
                ijref=None, fmask=None, stream=None, host_out=True, want_r8=True):
        """-> (psi4 (npj,npi) r4, dpsi8 (npj,npi) r8 or None): host arrays, or tensors in HBM when the
        inputs are resident and host_out is False."""
        names = ("zu", "e2u", "e3u", "zv", "e1v", "e3v", "e31d", "fmask")
        a = [_opt(x, n) for x, n in zip((zu, e2u, e3u, zv, e1v, e3v, e31d, fmask), names)]
        sp = _same_space(*[x[1] for x in a])
        shape = (self.npj, self.npi)
        if sp == MEM_DEVICE and not host_out:
            dev = next(x for x in (zu, zv) if x is not None).device
            p4 = torch.empty(shape, dtype=torch.float32, device=dev)
            d8 = torch.empty(shape, dtype=torch.float64, device=dev) if want_r8 else None
            pp4, pd8, so = C.c_void_p(p4.data_ptr()), C.c_void_p(d8.data_ptr()) if want_r8 else _null(), MEM_DEVICE
        else:
            p4 = np.empty(shape, np.float32)
            d8 = np.empty(shape, np.float64) if want_r8 else None
            pp4, pd8, so = C.c_void_p(p4.ctypes.data), C.c_void_p(d8.ctypes.data) if want_r8 else _null(), MEM_HOST
        check(_lib.load().cdft_psi_compute(self._h, a[0][0], a[1][0], a[2][0], a[3][0], a[4][0], a[5][0], a[6][0],
                                           self.npi if iiref is None else iiref, self.npj if ijref is None else ijref,
                                           a[7][0], sp, pp4, pd8, so, stream or current_stream_ptr()))
        return p4, d8


def vt_compute(u, v, t, s, e3u=None, e3v=None, out_r4=False, stream=None, devices=None):
    """cdfvT (src/cdfvT.f90:160-259).  u, v, t, s: (nframes,npk,npj,npi) r4 (NumPy: host; torch CUDA:
    resident).  -> dict of (npk,npj,npi) arrays vt, vs, ut, us (+ e3u, e3v with -vvl), r8 or r4,
    in the memory space of the inputs.  devices=[..]: row blocks on those devices (cdft_vt_compute_sharded)."""
    names = ("u", "v", "t", "s", "e3u", "e3v")
    a = [_opt(x, n) for x, n in zip((u, v, t, s, e3u, e3v), names)]
    sp = _same_space(*[x[1] for x in a])
    nfr, npk, npj, npi = a[0][2].shape
    vvl = e3u is not None
    keys = ("vt", "vs", "ut", "us") + (("e3u", "e3v") if vvl else ())
    if sp == MEM_DEVICE:
        o = {k: torch.empty((npk, npj, npi), dtype=torch.float32 if out_r4 else torch.float64, device=u.device)
             for k in keys}
        ptr = {k: C.c_void_p(x.data_ptr()) for k, x in o.items()}
    else:
        o = {k: np.empty((npk, npj, npi), np.float32 if out_r4 else np.float64) for k in keys}
        ptr = {k: C.c_void_p(x.ctypes.data) for k, x in o.items()}
    if devices is not None:
        if sp == MEM_DEVICE:
            torch.cuda.synchronize()              # the blocks run on their own streams: the inputs must be complete
        check(_lib.load().cdft_vt_compute_sharded(a[0][0], a[1][0], a[2][0], a[3][0], a[4][0], a[5][0], npk * npj * npi, nfr,
                                                  npi, npj, npk, ptr["vt"], ptr["vs"], ptr["ut"], ptr["us"],
                                                  ptr.get("e3u", _null()), ptr.get("e3v", _null()), 1 if out_r4 else 0,
                                                  *_devs(devices)))
        return o
    check(_lib.load().cdft_vt_compute(a[0][0], a[1][0], a[2][0], a[3][0], a[4][0], a[5][0], npk * npj * npi, nfr,
                                      npi, npj, npk, sp, ptr["vt"], ptr["vs"], ptr["ut"], ptr["us"],
                                      ptr.get("e3u", _null()), ptr.get("e3v", _null()), 1 if out_r4 else 0, sp,
                                      stream or current_stream_ptr()))
    return o


class MeanPlan(_Plan):
    """cdfmean (src/cdfmean.f90:405-431): cdft_mean_*; host arrays."""

    _destroy = "cdft_mean_destroy"

    def __init__(self, npiglo, npjglo, npk, nbasins=1):
        self.npi, self.npj, self.npk, self.nb = npiglo, npjglo, npk, nbasins
        self._h = C.c_void_p()
        check(_lib.load().cdft_mean_create(C.byref(self._h), npiglo, npjglo, npk, nbasins))

    def set_mesh(self, e1, e2, zmask, ibmask, e3=None, e31d=None, stream=None):
        """e1, e2 (npj,npi); zmask (npk,npj,npi) r4; ibmask (npj,npi,nb) i2; e3 (npk,npj,npi) or e31d (npk)."""
        k = [np.ascontiguousarray(e1, np.float32), np.ascontiguousarray(e2, np.float32),
             None if e3 is None else np.ascontiguousarray(e3, np.float32),
             None if e31d is None else np.ascontiguousarray(e31d, np.float32),
             np.ascontiguousarray(zmask, np.float32), np.ascontiguousarray(ibmask, np.int16)]
        ptr = [_null() if a is None else C.c_void_p(a.ctypes.data) for a in k]
        check(_lib.load().cdft_mean_set_mesh(self._h, *ptr, MEM_HOST, stream or current_stream_ptr()))

    def set_form(self, form):
        """0 = cdfmean, 1 = cdfsum, 2 = cdfsum on a file without depth, 3 = cdfmean without the variance sum (cdft_mean_set_form)."""
        check(_lib.load().cdft_mean_set_form(self._h, int(form)))

    def compute(self, zv, stream=None, device_out=None):
        """zv (npk,npj,npi) r4 -> sums (npk,nb,4) r8: dsurf, dvol2d, dsum2d, dvar2d.  device_out: a CUDA tensor of that shape --
        the sums stay in HBM and the call does not wait for them (the kernels alone)."""
        pz, sz, kz = _arg(zv, np.float32, "zv")
        if device_out is not None:
            check(_lib.load().cdft_mean_compute(self._h, pz, sz, C.c_void_p(device_out.data_ptr()), MEM_DEVICE, stream or current_stream_ptr()))
            return device_out
        out = np.empty((self.npk, self.nb, 4), np.float64)
        check(_lib.load().cdft_mean_compute(self._h, pz, sz, C.c_void_p(out.ctypes.data), MEM_HOST, stream or current_stream_ptr()))
        return out


class ZonalPlan(_Plan):
    """cdfzonalsum / cdfzonalmean (src/cdfzonalsum.f90:296-313, src/cdfzonalmean.f90:313-330): cdft_zonal_*."""

    _destroy = "cdft_zonal_destroy"

    def __init__(self, npiglo, npjglo, npk, nbasins=1):
        self.npi, self.npj, self.npk, self.nb = npiglo, npjglo, npk, nbasins
        self._h = C.c_void_p()
        check(_lib.load().cdft_zonal_create(C.byref(self._h), npiglo, npjglo, npk, nbasins))

    def set_mesh(self, e1, e2, zmaskvar, zmask, stream=None):
        """e1, e2 (npj,npi); zmaskvar (npk,npj,npi) r4; zmask (npj,npi,nb) r4 (host arrays)."""
        k = [np.ascontiguousarray(x, np.float32) for x in (e1, e2, zmaskvar, zmask)]
        check(_lib.load().cdft_zonal_set_mesh(self._h, *[C.c_void_p(a.ctypes.data) for a in k], MEM_HOST,
                                              stream or current_stream_ptr()))

    def area(self, nlev=None):
        """darea (nb,nlev,npj) r8 of the mesh: it does not depend on the field (cdft_zonal_area)."""
        nlev = self.npk if nlev is None else nlev
        out = np.empty((self.nb, nlev, self.npj), np.float64)
        check(_lib.load().cdft_zonal_area(self._h, nlev, C.c_void_p(out.ctypes.data), MEM_HOST, current_stream_ptr()))
        return out

    def compute(self, zv, mean_form=False, minmax=False, stream=None, device_out=False, pinned=False, want_area=True):
        """zv (nlev,npj,npi) r4 -> dict of (nb,nlev,npj) arrays: 'sum' [, 'area', 'max', 'min']; numpy arrays, or
        torch tensors left on the GPU with device_out=True.  pinned=True: the results land in page-locked arrays the plan
        keeps and hands out again at every call (CDFT_MEM_PINNED: no staging, no fresh pages); want_area=False: the mean
        form without its area, which area() gives once per mesh."""
        pz, sz, kz = _arg(zv, np.float32, "zv")
        nlev = kz.shape[0]
        shape = (self.nb, nlev, self.npj)
        if device_out:
            import torch
            new = lambda dt, k: torch.empty(shape, dtype=dt, device="cuda")         # noqa: E731
            ptr = lambda t: C.c_void_p(t.data_ptr())                                 # noqa: E731
            r8, r4, space = torch.float64, torch.float32, MEM_DEVICE
        elif pinned:
            if getattr(self, "_pin_shape", None) != shape:
                self._pin = {}
                self._pin_shape = shape

            def new(dt, k):
                if k not in self._pin:
                    self._pin[k] = PinnedArray(shape, dt)
                return self._pin[k].a
            ptr = lambda a: C.c_void_p(a.ctypes.data)                                # noqa: E731
            r8, r4, space = np.float64, np.float32, MEM_PINNED
        else:
            new = lambda dt, k: np.empty(shape, dt)                                  # noqa: E731
            ptr = lambda a: C.c_void_p(a.ctypes.data)                                # noqa: E731
            r8, r4, space = np.float64, np.float32, MEM_HOST
        out = {"sum": new(r8, "sum")}
        pa = px = pn = _null()
        if mean_form and want_area:
            out["area"] = new(r8, "area")
            pa = ptr(out["area"])
        if minmax:
            out["max"], out["min"] = new(r4, "max"), new(r4, "min")
            px, pn = ptr(out["max"]), ptr(out["min"])
        st = stream or current_stream_ptr()
        check(_lib.load().cdft_zonal_compute(self._h, pz, nlev, sz, 1 if mean_form else 0, ptr(out["sum"]), pa, px, pn, space, st))
        if pinned:
            check(_lib.load().cdft_stream_sync(st))
        return out


def mocsig(zv, zt, zs, e1v, ibmask, nbins, sigmin, sigstp, pref=0.0, eos=0, e3v=None, e31d=None, zveiv=None, gdep=None,
           spv=(0.0, 0.0, 0.0), isodep=False, stream=None):
    """cdft_mocsig_compute -> dmoc (npj,nbins,nb) r8 [, depi the same shape with isodep].  Host arrays, or CUDA tensors
    for every field (gdep stays a host array): the result is then a CUDA tensor too."""
    P = lambda x: _null() if x is None else C.c_void_p(x.ctypes.data)   # noqa: E731
    if _is_tensor(zv) and zv.is_cuda:
        keep = [_arg(x, np.float32, n) if x is not None else (_null(), MEM_DEVICE, None)
                for x, n in ((zv, "zv"), (zveiv, "zveiv"), (zt, "zt"), (zs, "zs"), (e1v, "e1v"), (e3v, "e3v"), (e31d, "e31d"))]
        pib, sib, kib = _arg(ibmask, np.int16, "ibmask")
        if any(sp != MEM_DEVICE for _, sp, _ in keep) or sib != MEM_DEVICE:
            raise ValueError("mocsig: every field must live in the same memory space")
        npk, npj, npi = zv.shape
        nb = ibmask.shape[-1]
        dmoc = torch.empty((npj, nbins, nb), dtype=torch.float64, device=zv.device)
        depi = torch.empty((npj, nbins, nb), dtype=torch.float64, device=zv.device) if isodep else None
        gd = None if gdep is None else np.ascontiguousarray(gdep, np.float32)
        ptr = [k[0] for k in keep]
        check(_lib.load().cdft_mocsig_compute(ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5], ptr[6], pib, P(gd), npi, npj, npk, nb,
                                              float(spv[0]), float(spv[1]), float(spv[2]), float(pref), eos, nbins, float(sigmin),
                                              float(sigstp), MEM_DEVICE, C.c_void_p(dmoc.data_ptr()),
                                              _null() if depi is None else C.c_void_p(depi.data_ptr()), MEM_DEVICE,
                                              stream or current_stream_ptr()))
        return (dmoc, depi) if isodep else dmoc
    f = lambda x: None if x is None else np.ascontiguousarray(x, np.float32)   # noqa: E731
    zv, zt, zs, e1v, e3v, e31d, zveiv, gdep = map(f, (zv, zt, zs, e1v, e3v, e31d, zveiv, gdep))
    ib = np.ascontiguousarray(ibmask, np.int16)
    npk, npj, npi = zv.shape
    nb = ib.shape[-1]
    dmoc = np.empty((npj, nbins, nb), np.float64)
    depi = np.empty((npj, nbins, nb), np.float64) if isodep else None
    check(_lib.load().cdft_mocsig_compute(P(zv), P(zveiv), P(zt), P(zs), P(e1v), P(e3v), P(e31d), P(ib), P(gdep), npi, npj, npk,
                                          nb, float(spv[0]), float(spv[1]), float(spv[2]), float(pref), eos, nbins,
                                          float(sigmin), float(sigstp), MEM_HOST, P(dmoc), P(depi), MEM_HOST,
                                          stream or current_stream_ptr()))
    return (dmoc, depi) if isodep else dmoc


def vint_cumul(zt, tmask, kout, hit, clip, rdep2, npko, sclf=1.0, mean=False, e3t=None, e31d=None, stream=None):
    """cdft_vint_cumul (host arrays): zt, tmask (npk,npj,npi) r4 -> (npko,npj,npi) r4."""
    a = [np.ascontiguousarray(zt, np.float32), None if e3t is None else np.ascontiguousarray(e3t, np.float32),
         None if e31d is None else np.ascontiguousarray(e31d, np.float32), np.ascontiguousarray(tmask, np.float32)]
    sch = [np.ascontiguousarray(kout, np.int32), np.ascontiguousarray(hit, np.int32), np.ascontiguousarray(clip, np.float32),
           np.ascontiguousarray(rdep2, np.float32)]
    npk, npj, npi = a[0].shape
    out = np.empty((npko, npj, npi), np.float32)
    ptr = [_null() if x is None else C.c_void_p(x.ctypes.data) for x in a]
    check(_lib.load().cdft_vint_cumul(*ptr, npi, npj, npk, *[C.c_void_p(x.ctypes.data) for x in sch], npko, float(sclf),
                                      1 if mean else 0, MEM_HOST, C.c_void_p(out.ctypes.data), MEM_HOST,
                                      stream or current_stream_ptr()))
    return out


def vertmean(zv, zmask, gdep_first, gdep_last, rdep_up, rdep_down, e3=None, e31d=None, stream=None):
    """cdft_vertmean (host arrays): the levels ik1..ik2 stacked (nlev,npj,npi) -> (npj,npi) r8."""
    a = [np.ascontiguousarray(zv, np.float32), None if e3 is None else np.ascontiguousarray(e3, np.float32),
         None if e31d is None else np.ascontiguousarray(e31d, np.float32), np.ascontiguousarray(zmask, np.float32)]
    nlev, npj, npi = a[0].shape
    out = np.empty((npj, npi), np.float64)
    ptr = [_null() if x is None else C.c_void_p(x.ctypes.data) for x in a]
    check(_lib.load().cdft_vertmean(*ptr, npi, npj, nlev, float(gdep_first), float(gdep_last), float(rdep_up), float(rdep_down),
                                    MEM_HOST, C.c_void_p(out.ctypes.data), MEM_HOST, stream or current_stream_ptr()))
    return out


def vint_transport(vel, m2d, e3=None, e31d=None, stream=None):
    """cdft_vint_transport: vel (npk,npj,npi) r4 -> (npj,npi) r8 (host arrays or CUDA tensors)."""
    pv, sv, kv = _arg(vel, np.float32, "vel")
    pm, sm, km = _arg(m2d, np.float32, "m2d")
    pe, se, ke = _opt(e3, "e3")
    p1, s1, k1 = _opt(e31d, "e31d")
    sp = _same_space(sv, sm, se, s1)
    npk, npj, npi = kv.shape
    if sp == MEM_DEVICE:
        out = torch.empty((npj, npi), dtype=torch.float64, device=kv.device)
        po = C.c_void_p(out.data_ptr())
    else:
        out = np.empty((npj, npi), np.float64)
        po = C.c_void_p(out.ctypes.data)
    check(_lib.load().cdft_vint_transport(pv, pm, pe, p1, npi, npj, npk, sp, po, sp, stream or current_stream_ptr()))
    return out


def vint_slope_rotate(dtrpu, dtrpv, hdepw, e1u, e2v, tmask, stream=None):
    """cdft_vint_slope_rotate (cdfvtrp -bathy): (npj,npi) r8 transports + r4 planes -> (along, cross) r8."""
    pu, su, ku = _arg(dtrpu, np.float64, "dtrpu")
    pv, sv, kv = _arg(dtrpv, np.float64, "dtrpv")
    ph, sh, kh = _arg(hdepw, np.float32, "hdepw")
    p1, s1, k1 = _arg(e1u, np.float32, "e1u")
    p2, s2, k2 = _arg(e2v, np.float32, "e2v")
    pm, sm, km = _arg(tmask, np.float32, "tmask")
    sp = _same_space(su, sv, sh, s1, s2, sm)
    npj, npi = ku.shape
    if sp == MEM_DEVICE:
        a = torch.empty((npj, npi), dtype=torch.float64, device=ku.device)
        c = torch.empty_like(a)
        pa, pc = C.c_void_p(a.data_ptr()), C.c_void_p(c.data_ptr())
    else:
        a, c = np.empty((npj, npi), np.float64), np.empty((npj, npi), np.float64)
        pa, pc = C.c_void_p(a.ctypes.data), C.c_void_p(c.ctypes.data)
    check(_lib.load().cdft_vint_slope_rotate(pu, pv, ph, p1, p2, pm, npi, npj, sp, pa, pc, sp, stream or current_stream_ptr()))
    return a, c


def vhst_transport(xt, xs, m2d, e3=None, e31d=None, stream=None):
    """cdft_vhst_transport: one pair of VT-file variables -> (heat, salt), each (npj,npi) r8."""
    pt, st_, kt = _arg(xt, np.float32, "xt")
    ps, ss, ks = _arg(xs, np.float32, "xs")
    pm, sm, km = _arg(m2d, np.float32, "m2d")
    pe, se, ke = _opt(e3, "e3")
    p1, s1, k1 = _opt(e31d, "e31d")
    sp = _same_space(st_, ss, sm, se, s1)
    npk, npj, npi = kt.shape
    if sp == MEM_DEVICE:
        h = torch.empty((npj, npi), dtype=torch.float64, device=kt.device)
        s = torch.empty_like(h)
        ph, pss = C.c_void_p(h.data_ptr()), C.c_void_p(s.data_ptr())
    else:
        h, s = np.empty((npj, npi), np.float64), np.empty((npj, npi), np.float64)
        ph, pss = C.c_void_p(h.ctypes.data), C.c_void_p(s.ctypes.data)
    check(_lib.load().cdft_vhst_transport(pt, ps, pm, pe, p1, npi, npj, npk, sp, ph, pss, sp, stream or current_stream_ptr()))
    return h, s


class CurlPlan(_Plan):
    """cdfcurl (src/cdfcurl.f90:262-301): cdft_curl_* of the C ABI."""

    _destroy = "cdft_curl_destroy"

    def __init__(self, npiglo, npjglo):
        self.npi, self.npj = npiglo, npjglo
        self._h = C.c_void_p()
        check(_lib.load().cdft_curl_create(C.byref(self._h), npiglo, npjglo))

    def set_mesh(self, e1u, e2v, e1f, e2f, ff=None, lperio=False, stream=None):
        a = [_arg(x, np.float32, n) for x, n in zip((e1u, e2v, e1f, e2f), ("e1u", "e2v", "e1f", "e2f"))]
        pf, sf, kf = _arg(ff, np.float64, "ff") if ff is not None else (_null(), None, None)
        sp = _same_space(*[x[1] for x in a], sf)
        check(_lib.load().cdft_curl_set_mesh(self._h, a[0][0], a[1][0], a[2][0], a[3][0], pf, 1 if lperio else 0, sp,
                                             stream or current_stream_ptr()))

    def compute(self, un, vn, ltpoint=False, out_r8=True, stream=None):
        """un, vn (nlev,npj,npi) r4 -> curl (nlev,npj,npi) r8 or r4 in the inputs' memory space."""
        pu, su, ku = _arg(un, np.float32, "un")
        pv, sv, kv = _arg(vn, np.float32, "vn")
        sp = _same_space(su, sv)
        nlev = int(np.prod(ku.shape)) // (self.npi * self.npj)
        shape = (nlev, self.npj, self.npi)
        if sp == MEM_DEVICE:
            out = torch.empty(shape, dtype=torch.float64 if out_r8 else torch.float32, device=un.device)
            po = C.c_void_p(out.data_ptr())
        else:
            out = np.empty(shape, np.float64 if out_r8 else np.float32)
            po = C.c_void_p(out.ctypes.data)
        check(_lib.load().cdft_curl_compute(self._h, pu, pv, nlev, sp, 1 if ltpoint else 0, po, 1 if out_r8 else 0,
                                            sp, stream or current_stream_ptr()))
        return out


__all__ = ["MocPlan", "HeatcPlan", "MeanPlan", "ZonalPlan", "MhstPlan", "PsiPlan", "CurlPlan", "vt_compute", "mocsig", "vint_cumul", "vertmean", "vint_transport", "vhst_transport", "sigmai", "sigma0", "sig_tool", "eos_field", "eos_point", "alpha", "beta", "albet", "spice", "eosbn2", "EOS_80", "EOS_TEOS10", "EOS_JMD94", "EOS_NTR", "EOS_MCD87", "EOS_FAST_FMA", "Timer", "CdftError", "device_count", "current_stream_ptr", "MEM_HOST", "MEM_DEVICE"]
