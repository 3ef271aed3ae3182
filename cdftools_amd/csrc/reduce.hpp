// reduce.hpp -- fixed-order wave / workgroup reductions of r8 partial sums (no atomics, so every
// kernel that uses them is run-to-run deterministic).
#pragma once
#include <hip/hip_runtime.h>

// The DPP control words below (row_bcast:15 / :31, row mirrors) are those of the GCN / CDNA wave64 encoding; this library
// is built for gfx950 alone (csrc/Makefile) and says so here rather than miscompiling quietly elsewhere (RDNA has no
// row_bcast).  Device passes only: the host pass of hipcc defines no __gfx*__ macro.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "cdftools_amd/csrc/reduce.hpp: the DPP wave reductions are written for CDNA (gfx950); build with --offload-arch=gfx950"
#endif

namespace cdft {

// One step of the reduction: v + (v moved across lanes by the DPP control word).  Data-parallel primitives run in the
// vector pipeline (two v_mov_dpp and a v_add_f64), where a __shfl_xor of a double is two ds_bpermute round trips
// through the LDS crossbar -- the level kernels do several of these per level and wave.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);          // (rows outside ROW_MASK add +0.0: a total of -0.0 terms comes out +0.0 -- as a
                                                  //  Fortran SUM, which starts from +0, returns it)
}

// Fixed order: quads, half rows, rows of 16, then the four rows in order -- the same tree on every call.
// (Round 1 reduced with __shfl_xor butterflies: another tree, so r8 sums formed since differ from round-1 output in the
// last bits -- inside the 1e-13 * sum|terms| every regrouped sum is held to, and identical from run to run as before.)
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0xb1, 0xf>(v);                    // quad_perm:[1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);                    // quad_perm:[2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);                   // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);                   // row_mirror: every lane holds its row of 16
    v = dpp_add<0x142, 0xa>(v);                   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);                   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);              // every lane holds the wave total
}

// Eight wave sums at a time through LDS: lane l adds the values of sum l / 8 held by lanes 8 (l % 8) .. 8 (l % 8) + 7, three
// DPP steps add the eight lanes of a sum -- 33 instructions for 8 sums where 8 x wave_sum() take 160.  `scr` is this wave's
// scratch of 8 x kSum8Stride doubles.  The total of sum n is returned in the lanes 8 n .. 8 n + 7.  A fixed order.
constexpr int kSum8Stride = 65;
__device__ __forceinline__ double wave_sum8(const double (&v)[8], double *scr, int lane) {
#pragma unroll
    for (int j = 0; j < 8; ++j) scr[j * kSum8Stride + lane] = v[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = lane >> 3, l8 = lane & 7;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += scr[n * kSum8Stride + l8 * 8 + i];
    t = dpp_add<0xb1, 0xf>(t);                    // quad_perm:[1,0,3,2]
    t = dpp_add<0x4e, 0xf>(t);                    // quad_perm:[2,3,0,1]
    t = dpp_add<0x141, 0xf>(t);                   // row_half_mirror
    __builtin_amdgcn_wave_barrier();              // (scr may be written again)
    return t;
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
