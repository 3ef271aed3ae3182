// What the HIP runtime costs a process before its first kernel: the first hipGetDeviceCount (the runtime and the driver's topology),
// the first call that needs the device's context (hipMemGetInfo), a stream, the first kernel launch (its code object).
//   hipcc --offload-arch=gfx950 -O2 tools/init_probe.hip -o /tmp/init_probe && /tmp/init_probe      (on the GPU box)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int *p) { if (p) *p = 1; }
int main() {
    double t0 = now();
    int n = 0; hipGetDeviceCount(&n);
    double t1 = now();
    size_t f = 0, t = 0; hipMemGetInfo(&f, &t);
    double t2 = now();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t3 = now();
    int *d = nullptr; hipMalloc(&d, 4);
    double t4 = now();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s);
    double t5 = now();
    void *h = nullptr; hipHostMalloc(&h, 256u << 20, hipHostMallocDefault);
    double t6 = now();
    hipHostFree(h);
    double t7 = now();
    printf("{\"devices\": %d, \"device_count_ms\": %.1f, \"mem_info_ms\": %.1f, \"stream_ms\": %.1f, \"malloc_ms\": %.1f, \"first_launch_ms\": %.1f, "
           "\"host_malloc_256MB_ms\": %.1f, \"host_free_ms\": %.1f}\n", n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3,
           (t6 - t5) * 1e3, (t7 - t6) * 1e3);
    return 0;
}
