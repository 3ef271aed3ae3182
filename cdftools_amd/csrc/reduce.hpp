// reduce.hpp -- fixed-order wave / workgroup reductions of r8 partial sums (no atomics, so every
// kernel that uses them is run-to-run deterministic).
#pragma once
#include <hip/hip_runtime.h>

namespace cdft {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;                                     // every lane holds the wave total
}

// Sum of `v` over the workgroup (blockDim.x a multiple of 64, <= 1024), combined in wave order.
// `scratch` needs 16 doubles of LDS; the result is valid in every thread.
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();                              // scratch may still be read from a previous call
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < nw; ++k) s += scratch[k];
    return s;
}

}  // namespace cdft
