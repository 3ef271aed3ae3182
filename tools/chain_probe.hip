// What one wave pays per step of a sequential r8 chain (cdfpsi's scan along j): the dependent v_add_f64 alone, with its term
// read from LDS (8 rows ahead), and with the result written back to LDS.   hipcc -O3 --offload-arch=gfx950 tools/chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE, int ACTIVE>   // 0: adds alone  1: + LDS reads  2: + LDS writes (in place)
__global__ void __launch_bounds__(1024) chain_kernel(double *out, int passes, double seed, int off) {
    extern __shared__ double dyn[];
    double *rows = dyn + off;
    const int lane = threadIdx.x, c = lane % 32;
    for (int r = lane; r < 128 * 32; r += blockDim.x) rows[r] = seed + r;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    double acc = seed;
    if (lane < ACTIVE) {
        for (int p = 0; p < passes; ++p) {
            double t[8], tn[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = MODE ? rows[r * 32 + c] : seed + r;
            for (int r0 = 0; r0 < 128; r0 += 8) {
                if (MODE && r0 + 8 < 128) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) tn[r] = rows[(r0 + 8 + r) * 32 + c];
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    acc = __dadd_rn(acc, -t[r]);
                    if (MODE == 2) rows[(r0 + r) * 32 + c] = acc;
                }
                if (MODE) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) t[r] = tn[r];
                }
            }
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE, int ACTIVE>
int run(double *out, int blocks, const char *what, int lds_kb = 32, int threads = 64) {
    const size_t shmem = (size_t)lds_kb * 1024; const int off = (lds_kb - 32) * 128;
    CK(hipFuncSetAttribute((const void *)chain_kernel<MODE, ACTIVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const int passes = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((chain_kernel<MODE, ACTIVE>), dim3(blocks), dim3(threads), shmem, 0, out, passes, 1.0, off);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((chain_kernel<MODE, ACTIVE>), dim3(blocks), dim3(threads), shmem, 0, out, passes, 1.0, off);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"what\": \"%s\", \"threads\": %d, \"lds_kb\": %d, \"active_lanes\": %d, \"workgroups\": %d, \"steps\": %d, \"ns_per_step\": %.2f}\n", what, threads, lds_kb, ACTIVE, blocks, passes * 128,
           ms * 1e6 / (passes * 128.0));
    return 0;
}

typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_kernel(const f4v *a, size_t n, float *o) {
    float acc = 0.f;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (size_t)gridDim.x * 256) { const f4v v = __builtin_nontemporal_load(a + p); acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) o[0] = acc;
}

// a short chain right behind a kernel that streams from HBM (what the scan sees behind vint_kernel): ns per step from the
// difference of two lengths
int short_after_stream(double *out) {
    f4v *buf; float *o;
    const size_t n = (size_t)1 << 26;                             // 1 GiB
    CK(hipMalloc((void **)&buf, n * 16)); CK(hipMemset(buf, 0, n * 16)); CK(hipMalloc((void **)&o, 64));
    CK(hipFuncSetAttribute((const void *)chain_kernel<2, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int with_stream = 0; with_stream < 2; ++with_stream) {
        float t[2] = {0.f, 0.f};
        const int len[2] = {10, 40};
        for (int rep = 0; rep < 20; ++rep)
            for (int k = 0; k < 2; ++k) {
                if (with_stream) hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, buf, n, o);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((chain_kernel<2, 32>), dim3(46), dim3(64), 32768, 0, out, len[k], 1.0, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 4) t[k] += ms / 16;
            }
        printf("{\"what\": \"a chain of 1280 / 5120 steps %s\", \"us_short\": %.2f, \"us_long\": %.2f, \"ns_per_step\": %.2f}\n",
               with_stream ? "right behind a 1 GiB streaming kernel" : "on an idle GPU", t[0] * 1e3, t[1] * 1e3, (t[1] - t[0]) * 1e6 / (30 * 128.0));
    }
    return 0;
}

int main() {
    double *out;
    CK(hipMalloc((void **)&out, 8 * 64 * 512));
    if (short_after_stream(out)) return 1;
    for (int blocks : {1, 46, 136}) {
        run<0, 32>(out, blocks, "dependent v_add_f64 alone");
        run<1, 32>(out, blocks, "+ term from LDS, 8 rows ahead");
        run<2, 32>(out, blocks, "+ result written to LDS");
        run<2, 64>(out, blocks, "+ result written to LDS");
        run<2, 32>(out, blocks, "+ result written to LDS", 64);
        run<2, 32>(out, blocks, "+ result written to LDS", 115);
        run<2, 32>(out, blocks, "+ result written to LDS", 150);
        run<2, 32>(out, blocks, "+ result written to LDS", 115, 1024);
        run<2, 32>(out, blocks, "+ result written to LDS", 32, 256);
    }
    return 0;
}
