// runtime.hip -- device/runtime entry points of the C ABI (include/cdftools_hip.h, "runtime").
#include <dlfcn.h>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include "common.hpp"
#include <utility>
#include <vector>

namespace cdft {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_NODEVICE,
                    "no HIP device available (hipGetDeviceCount: %s, count %d); "
                    "libcdftools_hip has no CPU fallback",
                    hipGetErrorString(e), n);
    }
    return CDFT_OK;
}

int dyn_lds_once(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const void *, int>, int>> done;      // ((kernel, device), bytes asked for)
    int dev = 0;
    CDFT_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (auto &d : done)
        if (d.first.first == kernel && d.first.second == dev) {
            if (d.second >= bytes) return CDFT_OK;
            CDFT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            d.second = bytes;
            return CDFT_OK;
        }
    CDFT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.push_back({{kernel, dev}, bytes});
    return CDFT_OK;
}

int DevBuf::reserve(size_t n) {
    if (n <= bytes) return CDFT_OK;
    release();
    if (hipMalloc(&ptr, n) != hipSuccess) {
        (void)hipGetLastError();
        ptr = nullptr;
        return fail(CDFT_ERR_ALLOC, "hipMalloc of %zu bytes failed", n);
    }
    bytes = n;
    return CDFT_OK;
}

void DevBuf::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

int to_device(const void *src, size_t n, int memspace, DevBuf &stage, hipStream_t st,
              const void **dev) {
    if (memspace == CDFT_MEM_DEVICE) {
        *dev = src;
        return CDFT_OK;
    }
    if (memspace != CDFT_MEM_HOST && memspace != CDFT_MEM_PINNED) return fail(CDFT_ERR_ARG, "bad memory-space tag %d", memspace);
    CDFT_TRY(stage.reserve(n));
    CDFT_HIP(hipMemcpyAsync(stage.ptr, src, n, hipMemcpyHostToDevice, st));
    *dev = stage.ptr;
    return CDFT_OK;
}

struct Timer {
    hipEvent_t t0, t1;
};

namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)(void);
roctx_push_t g_push = nullptr;
roctx_pop_t  g_pop = nullptr;
bool roctx_ready() {
    static const bool ok = [] {
        const char *e = getenv("CDFT_ROCTX");
        if (!e || e[0] != '1') return false;
        void *h = nullptr;
        for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
            if ((h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!h) return false;
        g_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
        g_pop  = (roctx_pop_t)dlsym(h, "roctxRangePop");
        return g_push && g_pop;
    }();
    return ok;
}
}  // namespace

namespace {
struct TimingRow { const char *name; long calls; double seconds; };
TimingRow g_rows[32];
int g_nrows = 0;
std::mutex g_timing_mutex;
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void timing_report() {
    for (int i = 0; i < g_nrows; ++i)
        fprintf(stderr, " CDFT_TIMING library: %-24s %6ld call%s %10.3f ms\n", g_rows[i].name, g_rows[i].calls, g_rows[i].calls == 1 ? " " : "s",
                g_rows[i].seconds * 1e3);
}
bool timing_ready() {
    static const bool ok = [] {
        const char *e = getenv("CDFT_TIMING");
        if (!e || e[0] != '1') return false;
        atexit(timing_report);
        return true;
    }();
    return ok;
}
}  // namespace

static thread_local int g_trace_mute = 0;
TraceMute::TraceMute() { ++g_trace_mute; }
TraceMute::~TraceMute() { --g_trace_mute; }

TraceRange::TraceRange(const char *nm) : name(nm), on(roctx_ready()), timed(timing_ready() && g_trace_mute == 0), t0(0.0) {
    if (on) (void)g_push(name);
    if (timed) t0 = now_s();
}
TraceRange::~TraceRange() {
    if (timed) {
        const double dt = now_s() - t0;
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        int i = 0;
        while (i < g_nrows && g_rows[i].name != name) ++i;       // (the names are string literals: one address each)
        if (i == g_nrows && g_nrows < 32) g_rows[g_nrows++] = TimingRow{name, 0, 0.0};
        if (i < g_nrows) { g_rows[i].calls += 1; g_rows[i].seconds += dt; }
    }
    if (on) (void)g_pop();
}

}  // namespace cdft

using namespace cdft;

extern "C" {

int cdft_version(void) { return 100; }

const char *cdft_last_error(void) { return g_err; }

int cdft_device_count(int *count) {
    if (!count) return fail(CDFT_ERR_ARG, "cdft_device_count: NULL argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return CDFT_OK;
}

int cdft_set_device(int device) {
    CDFT_TRY(require_device());
    CDFT_HIP(hipSetDevice(device));
    return CDFT_OK;
}

int cdft_malloc(void **ptr, size_t bytes) {
    if (!ptr) return fail(CDFT_ERR_ARG, "cdft_malloc: NULL argument");
    CDFT_TRY(require_device());
    if (hipMalloc(ptr, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return fail(CDFT_ERR_ALLOC, "hipMalloc of %zu bytes failed", bytes);
    }
    return CDFT_OK;
}

int cdft_free(void *ptr) {
    if (ptr) CDFT_HIP(hipFree(ptr));
    return CDFT_OK;
}

int cdft_mem_info(size_t *free_bytes, size_t *total_bytes) {
    CDFT_TRY(require_device());
    size_t f = 0, t = 0;
    CDFT_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return CDFT_OK;
}

int cdft_host_alloc(void **ptr, size_t bytes) {
    if (!ptr) return fail(CDFT_ERR_ARG, "cdft_host_alloc: NULL argument");
    CDFT_TRY(require_device());
    if (hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return fail(CDFT_ERR_ALLOC, "hipHostMalloc of %zu bytes failed", bytes);
    }
    return CDFT_OK;
}

int cdft_host_free(void *ptr) {
    if (ptr) CDFT_HIP(hipHostFree(ptr));
    return CDFT_OK;
}

// ------------------------------------------------------------------------------------------------ host feed
// A ring of page-locked slots and a stream: the host fills a slot (a file read), queues its copy to HBM and goes on
// filling the next one while the DMA runs.  A slot is handed out again only when the copy that last used it is done.
struct cdft_feed {
    std::vector<void *> slot;
    std::vector<hipEvent_t> done;
    std::vector<char> busy;
    size_t slot_bytes = 0;
    int cur = 0;                  // the slot being filled
    hipStream_t st = nullptr;
    // cdft_feed_push_rows in its "bounce" form: the slot travels as one linear DMA into one of two HBM buffers, and a kernel on
    // the same stream spreads its rows (5 TB/s: nothing beside the DMA's 57 GB/s)
    cdft::DevBuf bounce[2];
    int bcur = 0;
};

// rows of `words` 4-byte words, dense in src, dst_pitch words apart in dst
__global__ void __launch_bounds__(256) feed_rows_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t words,
                                                        int64_t dst_pitch, int64_t nrows) {
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const uint32_t *s = src + r * words;
        uint32_t *d = dst + r * dst_pitch;
        for (int64_t i = threadIdx.x; i < words; i += 256) d[i] = __builtin_nontemporal_load(s + i);
    }
}

// n signed bytes (what a NetCDF byte variable holds) widened to r4 at dst: the conversion netCDF's nf90_get_var makes on the host
__global__ void __launch_bounds__(256) feed_i1_r4_kernel(const int8_t *__restrict__ src, float *__restrict__ dst, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
        const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(src) + q);
        float4 v;
        v.x = (float)(int8_t)(w & 0xffu); v.y = (float)(int8_t)((w >> 8) & 0xffu); v.z = (float)(int8_t)((w >> 16) & 0xffu); v.w = (float)(int8_t)(w >> 24);
        float *o = dst + 4 * q;
        if ((reinterpret_cast<uintptr_t>(o) & 15u) == 0) *reinterpret_cast<float4 *>(o) = v;
        else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < n - 4 * n4) dst[4 * n4 + threadIdx.x] = (float)src[4 * n4 + threadIdx.x];
}

int cdft_feed_create(cdft_feed **feed, size_t slot_bytes, int nslots) {
    if (!feed) return fail(CDFT_ERR_ARG, "cdft_feed_create: NULL argument");
    *feed = nullptr;
    if (slot_bytes < 1 || nslots < 2 || nslots > 64) return fail(CDFT_ERR_ARG, "cdft_feed_create: %zu bytes x %d slots", slot_bytes, nslots);
    CDFT_TRY(require_device());
    cdft_feed *f = new cdft_feed;
    f->slot_bytes = slot_bytes;
    hipError_t e = hipStreamCreateWithFlags(&f->st, hipStreamNonBlocking);
    for (int i = 0; i < nslots && e == hipSuccess; ++i) {
        void *h = nullptr;
        hipEvent_t ev = nullptr;
        e = hipHostMalloc(&h, slot_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (h) f->slot.push_back(h);
        if (ev) f->done.push_back(ev);
        f->busy.push_back(0);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)cdft_feed_destroy(f);
        return fail(CDFT_ERR_ALLOC, "cdft_feed_create: %d page-locked slots of %zu bytes -> %s", nslots, slot_bytes, hipGetErrorString(e));
    }
    *feed = f;
    return CDFT_OK;
}

int cdft_feed_resize(cdft_feed *f, size_t slot_bytes, int nslots) {
    if (!f) return fail(CDFT_ERR_ARG, "cdft_feed_resize: NULL argument");
    if (slot_bytes < 1 || nslots < 2 || nslots > 64) return fail(CDFT_ERR_ARG, "cdft_feed_resize: %zu bytes x %d slots", slot_bytes, nslots);
    CDFT_HIP(hipStreamSynchronize(f->st));                      // nothing in flight: the slots can go
    // the new ring is built beside the old one and swapped in only when it is complete: a failure leaves the old ring as it was
    std::vector<hipEvent_t> more;
    std::vector<void *> fresh;
    hipError_t e = hipSuccess;
    while (e == hipSuccess && (int)(f->done.size() + more.size()) < nslots) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) more.push_back(ev);
    }
    for (int i = 0; i < nslots && e == hipSuccess; ++i) {
        void *h = nullptr;
        e = hipHostMalloc(&h, slot_bytes, hipHostMallocDefault);
        if (e == hipSuccess) fresh.push_back(h);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (void *h : fresh) (void)hipHostFree(h);
        for (hipEvent_t ev : more) (void)hipEventDestroy(ev);
        return fail(CDFT_ERR_ALLOC, "cdft_feed_resize: %d page-locked slots of %zu bytes -> %s (the ring is unchanged)", nslots, slot_bytes,
                    hipGetErrorString(e));
    }
    for (void *h : f->slot) (void)hipHostFree(h);
    f->slot.swap(fresh);
    f->done.insert(f->done.end(), more.begin(), more.end());
    f->busy.assign((size_t)nslots, 0);
    f->slot_bytes = slot_bytes;
    f->cur = 0;
    return CDFT_OK;
}

int cdft_feed_slot(cdft_feed *f, void **host, size_t *bytes) {
    if (!f || !host) return fail(CDFT_ERR_ARG, "cdft_feed_slot: NULL argument");
    if (f->slot.empty()) return fail(CDFT_ERR_STATE, "cdft_feed_slot: the ring has no slots");
    if (f->busy[f->cur]) {
        CDFT_HIP(hipEventSynchronize(f->done[f->cur]));
        f->busy[f->cur] = 0;
    }
    *host = f->slot[f->cur];
    if (bytes) *bytes = f->slot_bytes;
    return CDFT_OK;
}

int cdft_feed_push(cdft_feed *f, void *dst, size_t dst_offset, size_t bytes) {
    if (!f || !dst) return fail(CDFT_ERR_ARG, "cdft_feed_push: NULL argument");
    if (f->slot.empty()) return fail(CDFT_ERR_STATE, "cdft_feed_push: the ring has no slots");
    if (bytes > f->slot_bytes) return fail(CDFT_ERR_ARG, "cdft_feed_push: %zu bytes from a slot of %zu", bytes, f->slot_bytes);
    CDFT_HIP(hipMemcpyAsync((char *)dst + dst_offset, f->slot[f->cur], bytes, hipMemcpyHostToDevice, f->st));
    CDFT_HIP(hipEventRecord(f->done[f->cur], f->st));
    f->busy[f->cur] = 1;
    f->cur = (f->cur + 1) % (int)f->slot.size();
    return CDFT_OK;
}

int cdft_feed_push_rows(cdft_feed *f, void *dst, size_t dst_offset, size_t dst_pitch, size_t row_bytes, size_t nrows) {
    if (!f || !dst) return fail(CDFT_ERR_ARG, "cdft_feed_push_rows: NULL argument");
    if (f->slot.empty()) return fail(CDFT_ERR_STATE, "cdft_feed_push_rows: the ring has no slots");
    if (row_bytes < 1 || nrows < 1 || dst_pitch < row_bytes || row_bytes * nrows > f->slot_bytes)
        return fail(CDFT_ERR_ARG, "cdft_feed_push_rows: %zu rows of %zu bytes (pitch %zu) from a slot of %zu", nrows, row_bytes, dst_pitch,
                    f->slot_bytes);
    char *d = (char *)dst + dst_offset;
    // CDFT_FEED_ROWS=2d: one pitched copy straight from the slot (hipMemcpy2DAsync); default: the bounce form (struct cdft_feed)
    const char *e = getenv("CDFT_FEED_ROWS");
    const bool words_ok = row_bytes % 4 == 0 && dst_pitch % 4 == 0 && (uintptr_t)d % 4 == 0;
    if ((e && e[0] == '2') || !words_ok) {
        CDFT_HIP(hipMemcpy2DAsync(d, dst_pitch, f->slot[f->cur], row_bytes, row_bytes, nrows, hipMemcpyHostToDevice, f->st));
    } else {
        cdft::DevBuf &b = f->bounce[f->bcur];
        f->bcur ^= 1;
        CDFT_TRY(b.reserve(f->slot_bytes));
        CDFT_HIP(hipMemcpyAsync(b.ptr, f->slot[f->cur], row_bytes * nrows, hipMemcpyHostToDevice, f->st));
        const unsigned grid = (unsigned)(nrows < 16384 ? nrows : 16384);
        hipLaunchKernelGGL(feed_rows_kernel, dim3(grid), dim3(256), 0, f->st, (const uint32_t *)b.ptr, (uint32_t *)d,
                           (int64_t)(row_bytes / 4), (int64_t)(dst_pitch / 4), (int64_t)nrows);
        CDFT_LAUNCH_CHECK();
    }
    CDFT_HIP(hipEventRecord(f->done[f->cur], f->st));
    f->busy[f->cur] = 1;
    f->cur = (f->cur + 1) % (int)f->slot.size();
    return CDFT_OK;
}

int cdft_feed_push_i1_as_r4(cdft_feed *f, float *dst, size_t dst_offset, size_t n) {
    if (!f || !dst) return fail(CDFT_ERR_ARG, "cdft_feed_push_i1_as_r4: NULL argument");
    if (n == 0 || n > f->slot_bytes) return fail(CDFT_ERR_ARG, "cdft_feed_push_i1_as_r4: %zu bytes for a slot of %zu", n, f->slot_bytes);
    if (f->slot.empty()) return fail(CDFT_ERR_STATE, "cdft_feed_push_i1_as_r4: the ring has no slots");
    DevBuf &b = f->bounce[f->bcur];                              // (the two buffers of cdft_feed_push_rows: stream order keeps them apart)
    f->bcur ^= 1;
    CDFT_TRY(b.reserve(f->slot_bytes));
    CDFT_HIP(hipMemcpyAsync(b.ptr, f->slot[f->cur], n, hipMemcpyHostToDevice, f->st));
    const unsigned grid = (unsigned)std::min<size_t>((n / 4 + 255) / 256 + 1, 16384);
    hipLaunchKernelGGL(feed_i1_r4_kernel, dim3(grid), dim3(256), 0, f->st, (const int8_t *)b.ptr, dst + dst_offset, (int64_t)n);
    CDFT_LAUNCH_CHECK();
    CDFT_HIP(hipEventRecord(f->done[f->cur], f->st));
    f->busy[f->cur] = 1;
    f->cur = (f->cur + 1) % (int)f->slot.size();
    return CDFT_OK;
}

int cdft_feed_stream(cdft_feed *f, void **stream) {
    if (!f || !stream) return fail(CDFT_ERR_ARG, "cdft_feed_stream: NULL argument");
    *stream = (void *)f->st;
    return CDFT_OK;
}

int cdft_feed_wait(cdft_feed *f) {
    if (!f) return fail(CDFT_ERR_ARG, "cdft_feed_wait: NULL argument");
    CDFT_HIP(hipStreamSynchronize(f->st));
    for (char &b : f->busy) b = 0;
    return CDFT_OK;
}

int cdft_feed_destroy(cdft_feed *f) {
    if (!f) return CDFT_OK;
    if (f->st) (void)hipStreamSynchronize(f->st);
    for (hipEvent_t e : f->done) (void)hipEventDestroy(e);
    for (void *h : f->slot) (void)hipHostFree(h);
    f->bounce[0].release(); f->bounce[1].release();
    if (f->st) (void)hipStreamDestroy(f->st);
    (void)hipGetLastError();
    delete f;
    return CDFT_OK;
}

int cdft_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    CDFT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return CDFT_OK;
}

int cdft_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    CDFT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
    return CDFT_OK;
}

int cdft_stream_sync(void *stream) {
    CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
    return CDFT_OK;
}

int cdft_timer_create(void **timer) {
    if (!timer) return fail(CDFT_ERR_ARG, "cdft_timer_create: NULL argument");
    CDFT_TRY(require_device());
    Timer *t = new Timer;
    CDFT_HIP(hipEventCreate(&t->t0));
    CDFT_HIP(hipEventCreate(&t->t1));
    *timer = t;
    return CDFT_OK;
}

int cdft_timer_start(void *timer, void *stream) {
    if (!timer) return fail(CDFT_ERR_ARG, "cdft_timer_start: NULL timer");
    CDFT_HIP(hipEventRecord(static_cast<Timer *>(timer)->t0, as_stream(stream)));
    return CDFT_OK;
}

int cdft_timer_stop(void *timer, void *stream) {
    if (!timer) return fail(CDFT_ERR_ARG, "cdft_timer_stop: NULL timer");
    CDFT_HIP(hipEventRecord(static_cast<Timer *>(timer)->t1, as_stream(stream)));
    return CDFT_OK;
}

int cdft_timer_elapsed_ms(void *timer, float *ms) {
    if (!timer || !ms) return fail(CDFT_ERR_ARG, "cdft_timer_elapsed_ms: NULL argument");
    Timer *t = static_cast<Timer *>(timer);
    CDFT_HIP(hipEventSynchronize(t->t1));
    CDFT_HIP(hipEventElapsedTime(ms, t->t0, t->t1));
    return CDFT_OK;
}

int cdft_timer_destroy(void *timer) {
    if (!timer) return CDFT_OK;
    Timer *t = static_cast<Timer *>(timer);
    (void)hipEventDestroy(t->t0);
    (void)hipEventDestroy(t->t1);
    delete t;
    return CDFT_OK;
}

}  // extern "C"
