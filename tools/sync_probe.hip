// sync_probe.hip -- what a host result costs beside the kernels: N rounds of (a small kernel, then the host waits for it), the host
// waiting (a) in hipStreamSynchronize, (b) on a word the kernel writes into page-locked, device-mapped host memory (system-scope
// store after a fence), (c) hipEventSynchronize on a disable-timing event.  Prints microseconds per round; the kernel itself runs ~3 us.
//     hipcc -O2 --offload-arch=gfx950 tools/sync_probe.hip -o tools/bin/sync_probe && tools/bin/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void work(float *d, int n, volatile unsigned *flag, unsigned seq) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += d[i];
    d[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0 && flag) {
        __threadfence_system();
        __hip_atomic_store((unsigned *)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    float *d; CK(hipMalloc(&d, 1 << 20));
    CK(hipMemset(d, 0, 1 << 20));
    unsigned *hflag, *dflag;
    CK(hipHostMalloc((void **)&hflag, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&dflag, hflag, 0));
    *hflag = 0;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int w = 0; w < 200; ++w) { hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, d, 4096, nullptr, 0u); CK(hipStreamSynchronize(st)); }
    double t0 = now();
    for (int r = 0; r < rounds; ++r) { hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, d, 4096, nullptr, 0u); CK(hipStreamSynchronize(st)); }
    const double a = (now() - t0) / rounds * 1e6;
    t0 = now();
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, d, 4096, dflag, (unsigned)(r + 1));
        while (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != (unsigned)(r + 1)) _mm_pause();
    }
    const double b = (now() - t0) / rounds * 1e6;
    CK(hipStreamSynchronize(st));
    t0 = now();
    for (int r = 0; r < rounds; ++r) { hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, d, 4096, nullptr, 0u); CK(hipEventRecord(ev, st)); CK(hipEventSynchronize(ev)); }
    const double c = (now() - t0) / rounds * 1e6;
    // back to back, one wait at the end: what the launches themselves cost
    t0 = now();
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, d, 4096, nullptr, 0u);
    CK(hipStreamSynchronize(st));
    const double e = (now() - t0) / rounds * 1e6;
    printf("{\"rounds\": %d, \"us_per_round\": {\"hipStreamSynchronize\": %.2f, \"flag_in_mapped_host_memory\": %.2f, \"hipEventSynchronize\": %.2f, \"back_to_back_no_wait\": %.2f}}\n",
           rounds, a, b, c, e);
    return 0;
}
