"""ctypes binding of the C ABI declared in include/cdftools_hip.h.

The library is the product: there is no Python or CPU fallback.  A missing shared object, a
missing symbol or a missing GPU is an error the caller sees (CdftError), never a silent detour.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CDFT_LIB: another build of the same library (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get("CDFT_LIB") or os.path.join(HERE, "lib", "libcdftools_hip.so")

CDFT_OK = 0
CDFT_ERR_ARG, CDFT_ERR_HIP, CDFT_ERR_ALLOC, CDFT_ERR_STATE, CDFT_ERR_NODEVICE = 1, 2, 3, 4, 5
MEM_HOST, MEM_DEVICE, MEM_PINNED = 0, 1, 2


class CdftError(RuntimeError):
    """A C-ABI call returned a non-zero status (the Fortran host turns this into STOP 98)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libcdftools_hip status {code}: {message}")
        self.code = code
        self.message = message


_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_pvp = C.POINTER(C.c_void_p)

# symbol -> (restype, argtypes); must list every entry point of include/cdftools_hip.h
SIGNATURES = {
    "cdft_version": (_i, []),
    "cdft_last_error": (C.c_char_p, []),
    "cdft_device_count": (_i, [C.POINTER(_i)]),
    "cdft_set_device": (_i, [_i]),
    "cdft_malloc": (_i, [_pvp, _sz]),
    "cdft_free": (_i, [_vp]),
    "cdft_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "cdft_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "cdft_stream_sync": (_i, [_vp]),
    "cdft_timer_create": (_i, [_pvp]),
    "cdft_timer_start": (_i, [_vp, _vp]),
    "cdft_timer_stop": (_i, [_vp, _vp]),
    "cdft_timer_elapsed_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "cdft_timer_destroy": (_i, [_vp]),
    "cdft_moc_create": (_i, [_pvp, _i, _i, _i, _i]),
    "cdft_shard_from_env": (_i, [C.POINTER(_i), C.POINTER(_i), _i]),
    "cdft_shard_set_collective": (_i, [_i]),
    "cdft_shard_get_collective": (_i, [C.POINTER(_i)]),
    "cdft_host_alloc": (_i, [_pvp, _sz]),
    "cdft_host_free": (_i, [_vp]),
    "cdft_mem_info": (_i, [C.POINTER(_sz), C.POINTER(_sz)]),
    "cdft_feed_create": (_i, [_pvp, _sz, _i]),
    "cdft_feed_resize": (_i, [_vp, _sz, _i]),
    "cdft_feed_slot": (_i, [_vp, _pvp, C.POINTER(_sz)]),
    "cdft_feed_push": (_i, [_vp, _vp, _sz, _sz]),
    "cdft_feed_push_rows": (_i, [_vp, _vp, _sz, _sz, _sz, _sz]),
    "cdft_feed_push_i1_as_r4": (_i, [_vp, _vp, _sz, _sz]),
    "cdft_feed_stream": (_i, [_vp, _pvp]),
    "cdft_feed_wait": (_i, [_vp]),
    "cdft_feed_destroy": (_i, [_vp]),
    "cdft_moc_create_sharded": (_i, [_pvp, _i, _i, _i, _i, _i, C.POINTER(_i)]),
    "cdft_moc_set_mesh": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_moc_set_mesh_vmask": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_moc_compute": (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _vp]),
    "cdft_moc_decomp_set_mesh": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_moc_mask_e3v": (_i, [_vp, _vp, C.c_int64, _vp]),
    "cdft_moc_decomp": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_moc_set_tuning": (_i, [_vp, _i, _i, _i]),
    "cdft_moc_describe": (_i, [_vp, C.c_char_p, _sz]),
    "cdft_moc_profile_enable": (_i, [_vp, _i]),
    "cdft_moc_profile_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    "cdft_moc_profile_read_each": (_i, [_vp, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "cdft_moc_profile_read_block": (_i, [_vp, _i, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "cdft_moc_block_info": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "cdft_moc_reserve": (_i, [_vp, _i, _vp]),
    "cdft_moc_stage": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "cdft_moc_compute_staged": (_i, [_vp, _vp, _i, _vp]),
    "cdft_moc_set_basins": (_i, [_vp, _vp, _vp, _i, _vp]),
    "cdft_moc_stream_begin": (_i, [_vp, _i, _i, _vp]),
    "cdft_moc_stream_dest": (_i, [_vp, _i, _i, _pvp, C.POINTER(_sz)]),
    "cdft_moc_stream_vmask_dest": (_i, [_vp, _i, _pvp]),
    "cdft_moc_stream_compute": (_i, [_vp, _i, _i, _i, _vp]),
    "cdft_moc_stream_end": (_i, [_vp, _vp, _i, _vp]),
    "cdft_moc_compute_staged_windows": (_i, [_vp, _i, _vp, _i, _vp]),
    "cdft_moc_stage_open": (_i, [_vp, _i, _vp]),
    "cdft_moc_stage_dest": (_i, [_vp, _i, _i, _pvp, C.POINTER(_sz)]),
    "cdft_moc_stage_close": (_i, [_vp]),
    "cdft_moc_release_staged": (_i, [_vp]),
    "cdft_moc_stats": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "cdft_moc_stage_rows": (_i, [_vp, _vp, _i64, _i64, _i, _i, _vp]),
    "cdft_moc_compute_pitched": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _i, _vp]),
    "cdft_moc_destroy": (_i, [_vp]),
    "cdft_eos_sigmai": (_i, [_vp, _vp, _i64, C.c_float, _i, _i, _vp, _i, _i, _vp]),
    "cdft_eos_sigmai2d": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _i, _vp]),
    "cdft_eos_sig_tool": (_i, [_vp, _vp, _i64, C.c_float, C.c_float, _i, _i, _vp, _i, _i, _vp]),
    "cdft_eos_release": (_i, []),
    "cdft_eos_field": (_i, [_i, _vp, _vp, _vp, C.c_float, _i64, _i, _vp, _i, _i, _vp]),
    "cdft_eos_bn2": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i64, _i, _vp, _i, _i, _vp]),
    "cdft_eos_point": (_i, [_i, C.c_double, C.c_double, C.c_double, _i, C.POINTER(C.c_double)]),
    "cdft_eos_more_release": (_i, []),
    "cdft_heatc_create": (_i, [_pvp, _i, _i, _i]),
    "cdft_heatc_create_sharded": (_i, [_pvp, _i, _i, _i, _i, C.POINTER(_i)]),
    "cdft_heatc_set_mesh": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "cdft_heatc_compute": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "cdft_heatc_set_sum_mode": (_i, [_vp, _i]),
    "cdft_heatc_chain": (_i, [_vp, _vp, _vp]),
    "cdft_heatc_destroy": (_i, [_vp]),
    "cdft_mhst_create": (_i, [_pvp, _i, _i, _i, _i]),
    "cdft_mhst_create_sharded": (_i, [_pvp, _i, _i, _i, _i, _i, C.POINTER(_i)]),
    "cdft_mhst_set_mesh": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_mhst_compute": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "cdft_mhst_destroy": (_i, [_vp]),
    "cdft_vint_transport": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "cdft_vint_slope_rotate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "cdft_vhst_transport": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "cdft_vint_release": (_i, []),
    "cdft_mean_create": (_i, [_pvp, _i, _i, _i, _i]),
    "cdft_mean_set_mesh": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_mean_compute": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "cdft_mean_set_form": (_i, [_vp, _i]),
    "cdft_mean_destroy": (_i, [_vp]),
    "cdft_zonal_create": (_i, [_pvp, _i, _i, _i, _i]),
    "cdft_zonal_set_mesh": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_zonal_compute": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_zonal_area": (_i, [_vp, _i, _vp, _i, _vp]),
    "cdft_zonal_destroy": (_i, [_vp]),
    "cdft_vint_cumul": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, C.c_float, _i, _i, _vp, _i, _vp]),
    "cdft_vertmean": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, C.c_float, C.c_float, C.c_float, _i, _vp, _i, _vp]),
    "cdft_column_release": (_i, []),
    "cdft_mocsig_compute": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_float, C.c_float, C.c_float,
                                 C.c_float, _i, _i, C.c_float, C.c_float, _i, _vp, _vp, _i, _vp]),
    "cdft_mocsig_release": (_i, []),
    "cdft_psi_create": (_i, [_pvp, _i, _i, _i]),
    "cdft_psi_create_sharded": (_i, [_pvp, _i, _i, _i, _i, C.POINTER(_i)]),
    "cdft_psi_compute": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "cdft_psi_integrate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cdft_psi_scan_u": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp]),
    "cdft_psi_scan_open": (_i, [_vp, _i, _vp]),
    "cdft_psi_value": (_i, [_vp, _i, _i, C.POINTER(C.c_double), _vp]),
    "cdft_psi_finish": (_i, [_vp, _i, C.c_double, _i, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "cdft_psi_destroy": (_i, [_vp]),
    "cdft_vt_compute": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                             _i, _i, _vp]),
    "cdft_stencil_release": (_i, []),
    "cdft_vt_compute_sharded": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_i)]),
    "cdft_curl_create": (_i, [_pvp, _i, _i]),
    "cdft_curl_set_mesh": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "cdft_curl_compute": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "cdft_curl_destroy": (_i, [_vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libcdftools_hip.so (built in-tree by __graft_entry__.build()) and type its symbols."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CdftError(
            CDFT_ERR_STATE,
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cdftools_amd/csrc`); there is no fallback path",
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  When Python callers
    # also use torch (device memory, streams), torch must be loaded first so that this library
    # binds to the same runtime instance and torch's stream handles are valid here.  (The Fortran
    # host programs link the system runtime directly and never see torch.)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less callers use the system runtime
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of step
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != CDFT_OK:
        msg = load().cdft_last_error()
        raise CdftError(status, msg.decode("utf-8", "replace") if msg else "(no message)")
