// common.hpp -- shared host-side helpers of libcdftools_hip.so (error channel, staging buffers).
// gfx950 only; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
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
