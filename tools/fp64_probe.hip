// fp64_probe.hip -- what the vector pipeline of this chip sustains in r8 (tools, not product): independent chains of
// v_mul_f64 + v_add_f64 (what a non-contracted Horner evaluation issues) and of v_fma_f64, every lane of every SIMD busy.
// The EOS kernels are priced against these numbers and against the nominal ones (tools/bench_tools.py: 39.3 T instructions/s
// = 256 CUs x 4 SIMDs x 16 r8 lanes per cycle x 2.4 GHz, i.e. the 78.6 TFLOP/s FMA peak at one flop per instruction).
//   hipcc --offload-arch=gfx950 -O2 tools/fp64_probe.hip -o tools/bin/fp64_probe && tools/bin/fp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int FMA>
__global__ void __launch_bounds__(256) chains(double *out, int iters, double a, double b) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (double)(threadIdx.x + i) * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (FMA) x[i] = __fma_rn(x[i], a, b);
            else     x[i] = __dadd_rn(__dmul_rn(x[i], a), b);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 16, threads = 256, iters = 20000;
    double *d;
    (void)hipMalloc(&d, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int fma = 0; fma < 2; ++fma) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (fma) hipLaunchKernelGGL(chains<1>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999999, 1e-9);
            else     hipLaunchKernelGGL(chains<0>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999999, 1e-9);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks * threads * iters * 8 * (fma ? 1 : 2);
            printf("{\"kernel\": \"%s\", \"ms\": %.3f, \"T_instructions_per_s\": %.2f, \"TFLOP_per_s\": %.2f}\n", fma ? "v_fma_f64 chains" : "v_mul_f64 + v_add_f64 chains",
                   ms, instr / ms / 1e9, instr * (fma ? 2 : 1) / ms / 1e9);
        }
    }
    return 0;
}
