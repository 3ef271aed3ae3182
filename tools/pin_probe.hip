// pin_probe.hip -- what page-locked host memory costs and buys on this box (tools, not product):
//   hipHostMalloc time per GB, H2D rate from pageable and from pinned memory, and first-touch cost of the host pages.
//   hipcc --offload-arch=gfx950 -O2 tools/pin_probe.hip -o tools/bin/pin_probe && tools/bin/pin_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t GB = 1ull << 30;
    void *d = nullptr;
    hipMalloc(&d, 4 * GB);
    for (size_t n : {GB / 4, GB, 4 * GB}) {
        double t0 = now();
        void *h = nullptr;
        hipHostMalloc(&h, n, hipHostMallocDefault);
        double t1 = now();
        memset(h, 1, n);
        double t2 = now();
        hipMemcpy(d, h, n, hipMemcpyHostToDevice);
        double t3 = now();
        hipMemcpy(d, h, n, hipMemcpyHostToDevice);
        double t4 = now();
        hipHostFree(h);
        double t5 = now();
        void *p = malloc(n);
        memset(p, 1, n);
        double t6 = now();
        hipMemcpy(d, p, n, hipMemcpyHostToDevice);
        double t7 = now();
        hipHostRegister(p, n, hipHostRegisterDefault);
        double t8 = now();
        hipMemcpy(d, p, n, hipMemcpyHostToDevice);
        double t9 = now();
        hipHostUnregister(p);
        free(p);
        printf("{\"GB\": %.2f, \"hipHostMalloc_s\": %.4f, \"first_touch_s\": %.4f, \"h2d_pinned_GBs\": %.1f, \"h2d_pinned_again_GBs\": %.1f, "
               "\"hipHostFree_s\": %.4f, \"malloc_touch_s\": %.4f, \"h2d_pageable_GBs\": %.1f, \"hipHostRegister_s\": %.4f, \"h2d_registered_GBs\": %.1f}\n",
               (double)n / GB, t1 - t0, t2 - t1, n / (t3 - t2) / 1e9, n / (t4 - t3) / 1e9, t5 - t4, t6 - t5, n / (t7 - t6) / 1e9, t8 - t7,
               n / (t9 - t8) / 1e9);
    }
    return 0;
}
