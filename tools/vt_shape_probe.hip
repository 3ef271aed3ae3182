// What the memory system gives cdfvT's traffic without its arithmetic: a workgroup reads 1024 cells of 4 fields for NFR frames
// (a frame apart in memory) and writes 4 results of 1024 cells once.  Variants of who stores and how long a workgroup lives:
//   0  one tile per workgroup, the four stores at its end (the kernel's shape)
//   1  TPW tiles per workgroup in turn; the first frame of the next tile is requested before the stores of this one
//   2  as 1, the results handed through LDS to a fifth wave that only stores
// Build: hipcc -O3 --offload-arch=gfx950 tools/vt_shape_probe.hip -o tools/bin/vt_shape_probe ; run: vt_shape_probe [cells per frame] [frames]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

struct Args { const float *f[4]; float *o[4]; int64_t n, fstride; int nfr; int64_t ntiles; };

__device__ __forceinline__ int64_t band_item(int64_t total) {
    const int64_t b = blockIdx.x, per = (total + 7) / 8;
    return (b & 7) * per + (b >> 3);
}

template <int VARIANT, int TPW>
__global__ void __launch_bounds__(VARIANT == 2 ? 320 : 256) vt_shape_kernel(Args a) {
    __shared__ f4 hand[2][4][256];
    const int tid = threadIdx.x;
    const int64_t ngroups = (a.ntiles + TPW - 1) / TPW, g = band_item(ngroups);
    if (g >= ngroups) return;
    const int64_t t0 = g * TPW, t1 = t0 + TPW < a.ntiles ? t0 + TPW : a.ntiles;
    if (VARIANT == 2 && tid >= 256) {                              // the storing wave
        const int lane = tid - 256;
        int it = 0;
        for (int64_t t = t0; t < t1; ++t, ++it) {
            __syncthreads();                                      // tile t's results are in hand[it & 1]
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_nontemporal_store(hand[it & 1][q][r * 64 + lane], reinterpret_cast<f4 *>(a.o[q] + (t * 256 + r * 64 + lane) * 4));
        }
        return;
    }
    f4 cur[4], nxt[4], acc[4];
    auto fetch = [&](f4 (&x)[4], int64_t t, int f) {
        const int64_t e = (int64_t)f * a.fstride + (t * 256 + tid) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(a.f[q] + e));
    };
    fetch(cur, t0, 0);
    int it = 0;
    for (int64_t t = t0; t < t1; ++t, ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f4{0, 0, 0, 0};
        for (int f = 0; f < a.nfr; ++f) {
            if (f + 1 < a.nfr) fetch(nxt, t, f + 1);
            else if (VARIANT != 0 && t + 1 < t1) fetch(nxt, t + 1, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += cur[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        }
        if (VARIANT == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hand[it & 1][q][tid] = acc[q];
            __syncthreads();
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(acc[q], reinterpret_cast<f4 *>(a.o[q] + (t * 256 + tid) * 4));
        }
    }
}

template <int VARIANT, int TPW>
void run(const Args &a) {
    const int64_t ngroups = (a.ntiles + TPW - 1) / TPW;
    const dim3 grid((unsigned)(((ngroups + 7) / 8) * 8)), block(VARIANT == 2 ? 320 : 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((vt_shape_kernel<VARIANT, TPW>), grid, block, 0, 0, a);
    CK(hipDeviceSynchronize());
    float sum = 0.f;
    const int reps = 5;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((vt_shape_kernel<VARIANT, TPW>), grid, block, 0, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        sum += ms;
    }
    const double bytes = (double)a.ntiles * 1024 * (16.0 * a.nfr + 16.0);
    printf("{\"variant\": %d, \"tiles_per_workgroup\": %d, \"frames\": %d, \"ms\": %.4f, \"TBps\": %.3f, \"of_8TBps\": %.3f}\n", VARIANT, TPW, a.nfr,
           sum / reps, bytes / (sum / reps * 1e-3) / 1e12, bytes / (sum / reps * 1e-3) / 8e12);
    fflush(stdout);
}

int main(int argc, char **argv) {
    Args a;
    a.n = argc > 1 ? atoll(argv[1]) : 1168884900LL;               // eORCA12: 4322 x 3606 x 75
    a.nfr = argc > 2 ? atoi(argv[2]) : 6;
    a.n = a.n / 1024 * 1024;
    a.fstride = a.n; a.ntiles = a.n / 1024;
    for (int q = 0; q < 4; ++q) {
        CK(hipMalloc((void **)&a.f[q], (size_t)a.n * a.nfr * 4)); CK(hipMemset((void *)a.f[q], 0, (size_t)a.n * a.nfr * 4));
        CK(hipMalloc((void **)&a.o[q], (size_t)a.n * 4));
    }
    printf("{\"cells_per_frame\": %lld, \"frames\": %d}\n", (long long)a.n, a.nfr);
    run<0, 1>(a);
    run<1, 2>(a); run<1, 4>(a); run<1, 8>(a); run<1, 32>(a);
    run<2, 2>(a); run<2, 4>(a); run<2, 8>(a); run<2, 32>(a);
    return 0;
}
