// common.hpp -- shared host-side helpers of libcdftools_hip.so (error channel, staging buffers).
// gfx950 only; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/cdftools_hip.h"

namespace cdft {

// thread-local message of the last failing call (read through cdft_last_error()).
void set_error(const char *fmt, ...);
int  fail(int code, const char *fmt, ...);
// returns CDFT_OK when a HIP device is usable, CDFT_ERR_NODEVICE otherwise (message set).
int  require_device();

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// A device buffer that grows on demand (used to stage host arrays through HBM).
struct DevBuf {
    void  *ptr   = nullptr;
    size_t bytes = 0;
    int  reserve(size_t n);   // CDFT_OK / CDFT_ERR_ALLOC
    void release();
};

// Bring `n` bytes at `src` (host or device) into HBM.  Returns the device pointer to use:
// `src` itself when it is already in HBM, otherwise `stage.ptr` after an async H2D copy.
int to_device(const void *src, size_t n, int memspace, DevBuf &stage, hipStream_t st,
              const void **dev);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) made once per kernel and device instead of before every launch (the call
// goes to the runtime under its lock every time; a launch that follows it cannot have been queued behind the previous kernel).
int dyn_lds_once(const void *kernel, int bytes);

// cdfmoc -decomp's r8 zonal sums on the zonal class kernel (zonal.hip): see there
bool zonal_class_fits_moc(int npi, int nclass);
int zonal_class_moc_r8(const float *w, const double *field, const uint8_t *cls, const uint8_t *pat, int nclass, int npi, int npj, int nlev, int nb,
                       double *raw, hipStream_t st);

// A named range around a C-ABI call.  CDFT_ROCTX=1: a roctx range for rocprofv3 --marker-trace (the roctx library is looked
// up at run time: no link dependency, nothing when it is absent).  CDFT_TIMING=1: the wall time of the call (as the caller
// sees it: a call that returns host results includes its stream synchronisation) is added to a per-entry total, printed
// on stderr when the process ends.  Both are read from the environment when the first range is made; neither touches stdout.
struct TraceRange {
    explicit TraceRange(const char *name);
    ~TraceRange();
    const char *name;
    bool on, timed;
    double t0;
};
// While one is alive on a thread, the ranges made on that thread are not added to the CDFT_TIMING totals (they still
// show as roctx ranges): the row blocks of a sharded plan call the very entry points the caller's range already times,
// side by side on their own threads -- counted too, a call on four devices read as five calls and five times its wall time.
struct TraceMute {
    TraceMute();
    ~TraceMute();
};

// Streaming loads.  Every kernel of this path reads its 3-D fields exactly once; a nontemporal
// load tells L2 / Infinity Cache not to keep those lines (measured with tools/hbm_probe.hip on
// MI355X: 7.0-7.2 TB/s for a nontemporal read stream against 6.3-6.5 TB/s with plain loads).
typedef float f4a __attribute__((ext_vector_type(4)));               // 16-byte aligned address
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // any r4 address
__device__ __forceinline__ f4a ld_stream4(const float *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f4a *>(p));
}
__device__ __forceinline__ f4u ld_stream4u(const float *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f4u *>(p));
}
template <typename T>
__device__ __forceinline__ T ld_stream(const T *p) { return __builtin_nontemporal_load(p); }

// XCD bands.  Workgroups go round the 8 XCDs in launch order (workgroup b runs on XCD b % 8, each XCD with an L2 of its own).
// A kernel whose workgroups sweep memory in workgroup order takes xcd_band(on) as its workgroup index instead of blockIdx.x:
// XCD x then sweeps the x-th eighth of the work in order, and what two neighbouring workgroups both touch -- the line a row or a
// 16 KiB run ends in -- is asked for twice from the SAME L2.  The grid must be a multiple of 8 (xcd_band_grid); indices past the
// work come back >= the caller's count and are skipped by the bounds checks the kernels have anyway.  Speed only.
// Measured in one process on one set of fields (tools/ab_env.py): cdfmoc 9.86 -> 9.52 ms, cdfheatc 1.678 -> 1.645 ms at eORCA12.
__device__ __forceinline__ int64_t xcd_band(int on) {
    const int64_t b = blockIdx.x;
    return on ? (b & 7) * (int64_t)(gridDim.x >> 3) + (b >> 3) : b;
}
inline unsigned xcd_band_grid(int64_t nblocks, int on) { return (unsigned)(on ? (nblocks + 7) / 8 * 8 : nblocks); }
// CDFT_XCD_BANDS=0 switches every band mapping of the library off (A/B hook, read at every launch)
inline int xcd_bands_on() { const char *e = getenv("CDFT_XCD_BANDS"); return (e && e[0] == '0') ? 0 : 1; }

}  // namespace cdft

#define CDFT_HIP(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return cdft::fail(CDFT_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,  \
                              hipGetErrorString(e_));                                      \
    } while (0)

#define CDFT_TRY(call)                   \
    do {                                 \
        int s_ = (call);                 \
        if (s_ != CDFT_OK) return s_;    \
    } while (0)

#define CDFT_LAUNCH_CHECK()                                                                \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess)                                                              \
            return cdft::fail(CDFT_ERR_HIP, "%s:%d: kernel launch -> %s", __FILE__,        \
                              __LINE__, hipGetErrorString(e_));                            \
    } while (0)
