// Stand-alone HIP sanity check (no torch): device count + name, as the Fortran host programs see it.
#include <hip/hip_runtime.h>
#include <cstdio>
int main() {
    int n = -1;
    hipError_t e = hipGetDeviceCount(&n);
    printf("hipGetDeviceCount -> %s, n=%d\n", hipGetErrorString(e), n);
    if (e == hipSuccess && n > 0) {
        hipDeviceProp_t p;
        hipGetDeviceProperties(&p, 0);
        printf("device 0: %s, %d CUs, %.1f GB, arch %s\n", p.name, p.multiProcessorCount,
               p.totalGlobalMem / 1e9, p.gcnArchName);
    }
    return e == hipSuccess && n > 0 ? 0 : 1;
}
