// heatc.hip -- cdfheatc masked volume integrals (C ABI section "cdfheatc").
// Reference loop src/cdfheatc.f90:205-240, per level:
//     dsurf  = sum(e1t * e2t       * tmask)
//     dvol2d = sum(e1t * e2t * e3t * tmask)
//     dsum2d = sum(e1t * e2t * e3t * temp * tmask)
// The array expressions are r4 (left to right); the reference also SUMs in r4, which makes its own
// answer compiler-dependent at ~1e-4.  Here every r4 product is formed exactly
// as written and accumulated in r8 in a fixed order (more accurate, deterministic).
//
// HBM-bound: per cell temp 4 B + e3t 4 B + tmask 1 B (+ 8/npk) = 9.11 algorithmic bytes; the 2-D
// area e1t*e2t is re-read from L2 / Infinity Cache for every level.
#include <cstdlib>
#include <vector>
#include "common.hpp"
#include "reduce.hpp"
#include "shard.hpp"

using namespace cdft;

namespace cdft { namespace shard {
int heatc_parts(ShardSet &sh);
int heatc_set_mesh(ShardSet &sh, const float *e1t, const float *e2t, const float *e3t, const float *e31d, const void *tmask,
                   int mask_is_r4);
int heatc_compute(ShardSet &sh, const float *temp, const float *rmxldep, const float *gdepw, int mxloption, double *sums,
                  int out_space, hipStream_t out_stream, int sum_mode);
} }

struct cdft_heatc_plan {
    cdft::ShardSet *sh = nullptr;   // set: a shell over one plan per row block (shard.hpp)
    int npi = 0, npj = 0, npk = 0;
    float   *area = nullptr;   // (npi,npj) e1t*e2t in r4
    float   *e3t  = nullptr;   // (npi,npj,npk) or null when full-step
    float   *e31d = nullptr;   // (npk) full-step thickness, or null
    uint8_t *mask8 = nullptr;  // (npi,npj,npk) 0/1 bytes, or
    float   *mask4 = nullptr;  // (npi,npj,npk) r4 mask (arbitrary values, -M option)
    float   *gdepw = nullptr;  // (npk)
    double  *partial = nullptr;
    double  *hsums = nullptr;     // pinned, device-mapped host memory: the final kernel writes the result where the host reads it
    double  *hsums_dev = nullptr; // (its device address): no device-to-host copy call -- 5 % of a 0.2 ms call at eORCA025
    int nblk = 0;
    bool mesh_set = false;
    int sum_mode = 0;          // CDFT_HEATC_SUM_R8 | CDFT_HEATC_SUM_R4SEQ
    std::vector<float> carry;  // r4seq: the chains' start values (3*npk), empty = zeros; consumed by the next compute
    std::vector<float> last4;  // r4seq: the r4 sums of the last compute (what the next row block starts from)
    DevBuf st_a, st_b, st_temp, st_mxl, st_out, st_c4, st_s4;
};

namespace {

constexpr int kThreads = 256;

struct HeatcArgs {
    const float *area, *e3t, *e31d, *temp, *mxl, *gdepw;
    const uint8_t *mask8;
    const float *mask4;
    int64_t nij;
    int npk, mxlopt, nblk;
    double *partial;           // [npk][nblk][3]
};

__global__ void heatc_area_kernel(const float *__restrict__ e1t, const float *__restrict__ e2t,
                                  float *__restrict__ area, int64_t nij) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nij) area[e] = __fmul_rn(e1t[e], e2t[e]);
}

template <int VEC>
struct Vec;
typedef float f2a __attribute__((ext_vector_type(2)));
typedef unsigned char u4a __attribute__((ext_vector_type(4)));
typedef unsigned char u2a __attribute__((ext_vector_type(2)));
template <> struct Vec<4> { typedef f4a f; typedef u4a u; };
template <> struct Vec<2> { typedef f2a f; typedef u2a u; };
template <> struct Vec<1> { typedef float f;  typedef unsigned char u; };

template <int VEC> __device__ __forceinline__ void unpack(const typename Vec<VEC>::f &v, float (&o)[VEC]);
template <> __device__ __forceinline__ void unpack<4>(const f4a &v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void unpack<2>(const f2a &v, float (&o)[2]) { o[0] = v.x; o[1] = v.y; }
template <> __device__ __forceinline__ void unpack<1>(const float &v, float (&o)[1]) { o[0] = v; }
template <int VEC> __device__ __forceinline__ void unpacku(const typename Vec<VEC>::u &v, float (&o)[VEC]);
template <> __device__ __forceinline__ void unpacku<4>(const u4a &v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void unpacku<2>(const u2a &v, float (&o)[2]) { o[0] = v.x; o[1] = v.y; }
template <> __device__ __forceinline__ void unpacku<1>(const unsigned char &v, float (&o)[1]) { o[0] = v; }

constexpr int kVPT = 4;   // vectors per lane: each wave reads kVPT contiguous KiB-sized runs per field and level

// One workgroup owns kThreads*kVPT vectors of the horizontal plane for ALL levels: the area
// e1t*e2t (and the mixed-layer depth) stays in registers, every level streams temp / e3t / tmask
// past it.  Per level each wave reduces its three sums and parks them in LDS; at the end the waves
// are combined in wave order into partial[k][block][3].
template <int VEC, bool MASK8, bool FULL>
__global__ void __launch_bounds__(kThreads) heatc_level_kernel(HeatcArgs a) {
    extern __shared__ double wsum[];                   // [nwaves][npk][3]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = kThreads / 64;
    const int64_t nvec = a.nij / VEC;                  // nij % VEC == 0 by construction
    typedef typename Vec<VEC>::f FV;
    typedef typename Vec<VEC>::u UV;
    int64_t q[kVPT];
    bool ok[kVPT];
    float ar[kVPT][VEC], mx[kVPT][VEC];
#pragma unroll
    for (int v = 0; v < kVPT; ++v) {
        q[v] = (((int64_t)blockIdx.x * nwaves + wave) * kVPT + v) * 64 + lane;
        ok[v] = q[v] < nvec;
        if (!ok[v]) q[v] = 0;
        unpack<VEC>(reinterpret_cast<const FV *>(a.area)[q[v]], ar[v]);
        if (a.mxlopt) unpack<VEC>(reinterpret_cast<const FV *>(a.mxl)[q[v]], mx[v]);
        if (!ok[v])
#pragma unroll
            for (int c = 0; c < VEC; ++c) ar[v][c] = 0.f;       // contributes exact zeros
    }
    for (int k = 0; k < a.npk; ++k) {
        const int64_t base = (int64_t)k * nvec;        // vector offset of level k in a 3-D field
        const float e3k = FULL ? a.e31d[k] : 0.f;
        const float dw  = a.mxlopt ? a.gdepw[k] : 0.f;
        float e3[kVPT][VEC], tt[kVPT][VEC], mk[kVPT][VEC];
#pragma unroll
        for (int v = 0; v < kVPT; ++v) {
            unpack<VEC>(ld_stream(reinterpret_cast<const FV *>(a.temp) + base + q[v]), tt[v]);
            if (FULL) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) e3[v][c] = e3k;
            } else {
                unpack<VEC>(ld_stream(reinterpret_cast<const FV *>(a.e3t) + base + q[v]), e3[v]);
            }
            if (MASK8) unpacku<VEC>(ld_stream(reinterpret_cast<const UV *>(a.mask8) + base + q[v]), mk[v]);
            else       unpack<VEC>(ld_stream(reinterpret_cast<const FV *>(a.mask4) + base + q[v]), mk[v]);
        }
        double s_surf = 0.0, s_vol = 0.0, s_sum = 0.0;
#pragma unroll
        for (int v = 0; v < kVPT; ++v)
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float e = e3[v][c];
                if (a.mxlopt == 1)                        // src/cdfheatc.f90:220
                    e = fmaxf(0.f, fminf(e, __fsub_rn(mx[v][c], dw)));
                else if (a.mxlopt == -1)                  // :222
                    e = fminf(e, fmaxf(0.f, __fsub_rn(__fadd_rn(dw, e), mx[v][c])));
                const float mkc = ok[v] ? mk[v][c] : 0.f;
                const float ttc = ok[v] ? tt[v][c] : 0.f;
                const float pv = __fmul_rn(ar[v][c], e);                       // (e1t*e2t)*e3t
                s_surf += (double)__fmul_rn(ar[v][c], mkc);                    // :225
                s_vol  += (double)__fmul_rn(pv, mkc);                          // :226
                s_sum  += (double)__fmul_rn(__fmul_rn(pv, ttc), mkc);          // :229
            }
        s_surf = wave_sum(s_surf); s_vol = wave_sum(s_vol); s_sum = wave_sum(s_sum);
        if (lane == 0) {
            double *w = wsum + ((int64_t)wave * a.npk + k) * 3;
            w[0] = s_surf; w[1] = s_vol; w[2] = s_sum;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < a.npk * 3; t += kThreads) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += wsum[(int64_t)w * a.npk * 3 + t];
        const int k = t / 3, c = t - 3 * k;
        a.partial[((int64_t)k * a.nblk + blockIdx.x) * 3 + c] = s;
    }
}

// The same for planes whose size is not a multiple of 4 cells (eORCA025: 1442 x 1207) or whose arrays are not
// 16-byte aligned: 16-byte loads at r4 alignment (a level then starts anywhere), the plane covered by vectors at cells
// 0, 4, 8, ... plus, when nij is not a multiple of 4, one vector anchored at nij-4 whose leading cells repeat the
// previous vector's and count as zeros -- every load is a full vector inside the level, without a branch.
typedef unsigned char u4u __attribute__((ext_vector_type(4), aligned(1)));

template <bool MASK8, bool FULL>
__global__ void __launch_bounds__(kThreads) heatc_level_u4_kernel(HeatcArgs a) {
    extern __shared__ double wsum[];                   // [nwaves][npk][3]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = kThreads / 64;
    const int64_t nfull = a.nij >> 2;                  // vectors 0..nfull-1 start at 4v
    int64_t e0[kVPT];
    bool own[kVPT][4];
    float ar[kVPT][4], mx[kVPT][4];
#pragma unroll
    for (int v = 0; v < kVPT; ++v) {
        const int64_t q = (((int64_t)blockIdx.x * nwaves + wave) * kVPT + v) * 64 + lane;
        e0[v] = min(4 * q, a.nij - 4);
        const f4u av = *reinterpret_cast<const f4u *>(a.area + e0[v]);
        const float a4[4] = {av.x, av.y, av.z, av.w};
        float m4[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.mxlopt) { const f4u mv = *reinterpret_cast<const f4u *>(a.mxl + e0[v]); m4[0] = mv.x; m4[1] = mv.y; m4[2] = mv.z; m4[3] = mv.w; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            own[v][c] = q < nfull || (q == nfull && e0[v] + c >= 4 * nfull);
            ar[v][c] = own[v][c] ? a4[c] : 0.f;          // contributes exact zeros
            mx[v][c] = m4[c];
        }
    }
    for (int k = 0; k < a.npk; ++k) {
        const int64_t base = (int64_t)k * a.nij;
        const float e3k = FULL ? a.e31d[k] : 0.f;
        const float dw  = a.mxlopt ? a.gdepw[k] : 0.f;
        float e3[kVPT][4], tt[kVPT][4], mk[kVPT][4];
#pragma unroll
        for (int v = 0; v < kVPT; ++v) {
            const f4u t4 = ld_stream4u(a.temp + base + e0[v]);
            tt[v][0] = t4.x; tt[v][1] = t4.y; tt[v][2] = t4.z; tt[v][3] = t4.w;
            if (FULL) {
#pragma unroll
                for (int c = 0; c < 4; ++c) e3[v][c] = e3k;
            } else {
                const f4u e4 = ld_stream4u(a.e3t + base + e0[v]);
                e3[v][0] = e4.x; e3[v][1] = e4.y; e3[v][2] = e4.z; e3[v][3] = e4.w;
            }
            if (MASK8) {
                const u4u m = __builtin_nontemporal_load(reinterpret_cast<const u4u *>(a.mask8 + base + e0[v]));
                mk[v][0] = m.x; mk[v][1] = m.y; mk[v][2] = m.z; mk[v][3] = m.w;
            } else {
                const f4u m = ld_stream4u(a.mask4 + base + e0[v]);
                mk[v][0] = m.x; mk[v][1] = m.y; mk[v][2] = m.z; mk[v][3] = m.w;
            }
        }
        double s_surf = 0.0, s_vol = 0.0, s_sum = 0.0;
#pragma unroll
        for (int v = 0; v < kVPT; ++v)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float e = e3[v][c];
                if (a.mxlopt == 1)                        // src/cdfheatc.f90:220
                    e = fmaxf(0.f, fminf(e, __fsub_rn(mx[v][c], dw)));
                else if (a.mxlopt == -1)                  // :222
                    e = fminf(e, fmaxf(0.f, __fsub_rn(__fadd_rn(dw, e), mx[v][c])));
                const float mkc = own[v][c] ? mk[v][c] : 0.f;
                const float ttc = own[v][c] ? tt[v][c] : 0.f;
                const float pv = __fmul_rn(ar[v][c], e);                       // (e1t*e2t)*e3t
                s_surf += (double)__fmul_rn(ar[v][c], mkc);                    // :225
                s_vol  += (double)__fmul_rn(pv, mkc);                          // :226
                s_sum  += (double)__fmul_rn(__fmul_rn(pv, ttc), mkc);          // :229
            }
        s_surf = wave_sum(s_surf); s_vol = wave_sum(s_vol); s_sum = wave_sum(s_sum);
        if (lane == 0) {
            double *w = wsum + ((int64_t)wave * a.npk + k) * 3;
            w[0] = s_surf; w[1] = s_vol; w[2] = s_sum;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < a.npk * 3; t += kThreads) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += wsum[(int64_t)w * a.npk * 3 + t];
        const int k = t / 3, c = t - 3 * k;
        a.partial[((int64_t)k * a.nblk + blockIdx.x) * 3 + c] = s;
    }
}

int launch_u4(const HeatcArgs &a, hipStream_t st) {
    dim3 grid(a.nblk), block(kThreads);
    const size_t shmem = (size_t)(kThreads / 64) * a.npk * 3 * sizeof(double);
    const bool m8 = a.mask8 != nullptr, full = a.e31d != nullptr;
    if (m8 && full)        hipLaunchKernelGGL((heatc_level_u4_kernel<true, true>), grid, block, shmem, st, a);
    else if (m8 && !full)  hipLaunchKernelGGL((heatc_level_u4_kernel<true, false>), grid, block, shmem, st, a);
    else if (!m8 && full)  hipLaunchKernelGGL((heatc_level_u4_kernel<false, true>), grid, block, shmem, st, a);
    else                   hipLaunchKernelGGL((heatc_level_u4_kernel<false, false>), grid, block, shmem, st, a);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's own order.  src/cdfheatc.f90:225-229 forms  SUM(e1t*e2t*tmask), SUM(e1t*e2t*e3t*tmask),
// SUM(e1t*e2t*e3t*temp*tmask)  in REAL(KIND=4): gfortran -O and flang -O1 compile that SUM into one sequential r4
// loop over the array in memory order, so the value a reference build returns is  s = 0; s = s + x(p), p = 1..n,
// rounded to r4 at every step (9e-5 away from the r8 sum at eORCA025).  A sequential rounding
// chain cannot be regrouped, so it is reproduced as it is: one wave per level; the 64 lanes form the three r4
// products of 256 cells at a time in parallel (coalesced 16-byte loads, the next batch already requested) and park
// them in LDS, then lanes 0..2 -- one per sum -- add their 256 values in order.  1.74e6 dependent v_add_f32 per chain
// at eORCA025: a few milliseconds for all 225 chains, which run side by side on 75 CUs.  `carry` (3*npk r4, or NULL)
// continues the chains of the row block to the south (sharded plans): the result does not depend on the partition.
template <bool MASK8, bool FULL>
__global__ void __launch_bounds__(64) heatc_seq_kernel(HeatcArgs a, const float *__restrict__ carry, float *__restrict__ sums4) {
    __shared__ float lds[3][256];
    const int k = blockIdx.x, lane = threadIdx.x, ch = lane < 3 ? lane : 2;
    const int64_t base = (int64_t)k * a.nij;
    const float e3k = FULL ? a.e31d[k] : 0.f;
    const float dw  = a.mxlopt ? a.gdepw[k] : 0.f;
    float s = carry ? carry[k * 3 + ch] : 0.f;
    struct Batch { float ar[4], e3[4], tt[4], mk[4], mx[4]; };
    auto fetch = [&](Batch &x, int64_t p0) {
        const int64_t p = p0 + 4 * lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) { x.ar[c] = x.e3[c] = x.tt[c] = x.mk[c] = x.mx[c] = 0.f; }
        if (p + 3 < a.nij) {
            const f4u av = *reinterpret_cast<const f4u *>(a.area + p);
            x.ar[0] = av.x; x.ar[1] = av.y; x.ar[2] = av.z; x.ar[3] = av.w;
            const f4u tv = ld_stream4u(a.temp + base + p);
            x.tt[0] = tv.x; x.tt[1] = tv.y; x.tt[2] = tv.z; x.tt[3] = tv.w;
            if (!FULL) { const f4u ev = ld_stream4u(a.e3t + base + p); x.e3[0] = ev.x; x.e3[1] = ev.y; x.e3[2] = ev.z; x.e3[3] = ev.w; }
            if (MASK8) { const u4u m = __builtin_nontemporal_load(reinterpret_cast<const u4u *>(a.mask8 + base + p)); x.mk[0] = m.x; x.mk[1] = m.y; x.mk[2] = m.z; x.mk[3] = m.w; }
            else       { const f4u m = ld_stream4u(a.mask4 + base + p); x.mk[0] = m.x; x.mk[1] = m.y; x.mk[2] = m.z; x.mk[3] = m.w; }
            if (a.mxlopt) { const f4u mv = *reinterpret_cast<const f4u *>(a.mxl + p); x.mx[0] = mv.x; x.mx[1] = mv.y; x.mx[2] = mv.z; x.mx[3] = mv.w; }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (p + c < a.nij) {
                    x.ar[c] = a.area[p + c];
                    x.tt[c] = a.temp[base + p + c];
                    if (!FULL) x.e3[c] = a.e3t[base + p + c];
                    x.mk[c] = MASK8 ? (float)a.mask8[base + p + c] : a.mask4[base + p + c];
                    if (a.mxlopt) x.mx[c] = a.mxl[p + c];
                }
        }
    };
    Batch cur, nxt;
    fetch(cur, 0);
    for (int64_t p0 = 0; p0 < a.nij; p0 += 256) {
        if (p0 + 256 < a.nij) fetch(nxt, p0 + 256);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float e = FULL ? e3k : cur.e3[c];
            if (a.mxlopt == 1)       e = fmaxf(0.f, fminf(e, __fsub_rn(cur.mx[c], dw)));                 // :220
            else if (a.mxlopt == -1) e = fminf(e, fmaxf(0.f, __fsub_rn(__fadd_rn(dw, e), cur.mx[c])));   // :222
            const float pv = __fmul_rn(cur.ar[c], e);
            lds[0][4 * lane + c] = __fmul_rn(cur.ar[c], cur.mk[c]);                                   // :225
            lds[1][4 * lane + c] = __fmul_rn(pv, cur.mk[c]);                                          // :226
            lds[2][4 * lane + c] = __fmul_rn(__fmul_rn(pv, cur.tt[c]), cur.mk[c]);                    // :229
        }
        __syncthreads();
        const int n = (int)min((int64_t)256, a.nij - p0);
        const float *x = lds[ch];
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            const f4a v = *reinterpret_cast<const f4a *>(x + i);
            s = __fadd_rn(s, v.x); s = __fadd_rn(s, v.y); s = __fadd_rn(s, v.z); s = __fadd_rn(s, v.w);
        }
        for (; i < n; ++i) s = __fadd_rn(s, x[i]);
        __syncthreads();
        cur = nxt;
    }
    if (lane < 3) sums4[k * 3 + lane] = s;
}

int launch_seq(const HeatcArgs &a, const float *carry, float *sums4, hipStream_t st) {
    dim3 grid(a.npk), block(64);
    const bool m8 = a.mask8 != nullptr, full = a.e31d != nullptr;
    if (m8 && full)        hipLaunchKernelGGL((heatc_seq_kernel<true, true>), grid, block, 0, st, a, carry, sums4);
    else if (m8 && !full)  hipLaunchKernelGGL((heatc_seq_kernel<true, false>), grid, block, 0, st, a, carry, sums4);
    else if (!m8 && full)  hipLaunchKernelGGL((heatc_seq_kernel<false, true>), grid, block, 0, st, a, carry, sums4);
    else                   hipLaunchKernelGGL((heatc_seq_kernel<false, false>), grid, block, 0, st, a, carry, sums4);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

// sums[k][3] = partials of level k added over the blocks in a fixed order (one workgroup per level)
__global__ void __launch_bounds__(kThreads) heatc_final_kernel(const double *__restrict__ partial, int npk, int nblk,
                                                               double *__restrict__ sums) {
    __shared__ double scratch[16];
    const int k = blockIdx.x;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += kThreads)
        for (int c = 0; c < 3; ++c) s[c] += partial[((int64_t)k * nblk + b) * 3 + c];
    for (int c = 0; c < 3; ++c) {
        const double t = block_sum(s[c], scratch);
        if (threadIdx.x == 0) sums[k * 3 + c] = t;
    }
}

template <int VEC>
int launch_vec(const HeatcArgs &a, hipStream_t st) {
    dim3 grid(a.nblk), block(kThreads);
    const size_t shmem = (size_t)(kThreads / 64) * a.npk * 3 * sizeof(double);
    const bool m8 = a.mask8 != nullptr, full = a.e31d != nullptr;
    if (m8 && full)        hipLaunchKernelGGL((heatc_level_kernel<VEC, true, true>), grid, block, shmem, st, a);
    else if (m8 && !full)  hipLaunchKernelGGL((heatc_level_kernel<VEC, true, false>), grid, block, shmem, st, a);
    else if (!m8 && full)  hipLaunchKernelGGL((heatc_level_kernel<VEC, false, true>), grid, block, shmem, st, a);
    else                   hipLaunchKernelGGL((heatc_level_kernel<VEC, false, false>), grid, block, shmem, st, a);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}


bool aligned_to(const void *p, size_t n) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % n) == 0; }

int dev_copy(void **dst, const void *src, size_t bytes, int space, hipStream_t st) {
    if (!*dst && hipMalloc(dst, bytes + 16) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdfheatc: device allocation of %zu bytes failed", bytes);
    }
    CDFT_HIP(hipMemcpyAsync(*dst, src, bytes, space == CDFT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    return CDFT_OK;
}

}  // namespace

extern "C" {

int cdft_heatc_create(cdft_heatc_plan **plan, int npiglo, int npjglo, int npk) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_heatc_create: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 1)
        return fail(CDFT_ERR_ARG, "cdft_heatc_create: bad sizes %d x %d x %d", npiglo, npjglo, npk);
    CDFT_TRY(require_device());
    cdft_heatc_plan *p = new cdft_heatc_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk;
    const int64_t nij = (int64_t)npiglo * npjglo;
    p->nblk = (int)((nij + kThreads * 4 - 1) / (kThreads * 4));   // blocks when VEC = 1 (the most)
    if (hipMalloc((void **)&p->area, nij * sizeof(float) + 16) != hipSuccess ||
        hipMalloc((void **)&p->gdepw, npk * sizeof(float) + 16) != hipSuccess ||
        hipMalloc((void **)&p->partial, (size_t)npk * p->nblk * 3 * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void **)&p->hsums, (size_t)npk * 3 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&p->hsums_dev, p->hsums, 0) != hipSuccess) {
        (void)hipGetLastError();
        cdft_heatc_destroy(p);
        return fail(CDFT_ERR_ALLOC, "cdft_heatc_create: device allocation failed");
    }
    *plan = p;
    return CDFT_OK;
}

int cdft_heatc_create_sharded(cdft_heatc_plan **plan, int npiglo, int npjglo, int npk, int ndev, const int *devices) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_heatc_create_sharded: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 1)
        return fail(CDFT_ERR_ARG, "cdft_heatc_create_sharded: bad sizes %d x %d x %d", npiglo, npjglo, npk);
    cdft_heatc_plan *p = new cdft_heatc_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk;
    int rc = shard::make(&p->sh, npiglo, npjglo, npk, 0, ndev, devices, 0, 7);
    if (rc == CDFT_OK) rc = shard::heatc_parts(*p->sh);
    if (rc != CDFT_OK) { cdft_heatc_destroy(p); return rc; }
    *plan = p;
    return CDFT_OK;
}

int cdft_heatc_set_mesh(cdft_heatc_plan *p, const float *e1t, const float *e2t, const float *e3t,
                        const float *e31d, const void *tmask, int mask_is_r4, int memspace, void *stream) {
    if (!p || !e1t || !e2t || !tmask || (!e3t && !e31d) || (e3t && e31d))
        return fail(CDFT_ERR_ARG, "cdft_heatc_set_mesh: need e1t, e2t, tmask and exactly one of e3t / e31d");
    if (p->sh) {
        if (memspace == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
        CDFT_TRY(shard::heatc_set_mesh(*p->sh, e1t, e2t, e3t, e31d, tmask, mask_is_r4));
        p->mesh_set = true;
        return CDFT_OK;
    }
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj, n3 = nij * p->npk;
    const void *d1, *d2;
    CDFT_TRY(to_device(e1t, nij * sizeof(float), memspace, p->st_a, st, &d1));
    CDFT_TRY(to_device(e2t, nij * sizeof(float), memspace, p->st_b, st, &d2));
    hipLaunchKernelGGL(heatc_area_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st,
                       (const float *)d1, (const float *)d2, p->area, nij);
    CDFT_LAUNCH_CHECK();
    if (e3t) {
        CDFT_TRY(dev_copy((void **)&p->e3t, e3t, n3 * sizeof(float), memspace, st));
        if (p->e31d) { (void)hipFree(p->e31d); p->e31d = nullptr; }
    } else {
        CDFT_TRY(dev_copy((void **)&p->e31d, e31d, p->npk * sizeof(float), memspace, st));
        if (p->e3t) { (void)hipFree(p->e3t); p->e3t = nullptr; }
    }
    if (mask_is_r4) {
        CDFT_TRY(dev_copy((void **)&p->mask4, tmask, n3 * sizeof(float), memspace, st));
        if (p->mask8) { (void)hipFree(p->mask8); p->mask8 = nullptr; }
    } else {
        CDFT_TRY(dev_copy((void **)&p->mask8, tmask, n3, memspace, st));
        if (p->mask4) { (void)hipFree(p->mask4); p->mask4 = nullptr; }
    }
    CDFT_HIP(hipStreamSynchronize(st));
    p->st_a.release(); p->st_b.release();
    p->mesh_set = true;
    return CDFT_OK;
}

int cdft_heatc_compute(cdft_heatc_plan *p, const float *temp, int temp_space, const float *rmxldep,
                       const float *gdepw, int mxloption, double *sums, int out_space, void *stream) {
    if (!p || !temp || !sums) return fail(CDFT_ERR_ARG, "cdft_heatc_compute: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_heatc_compute: cdft_heatc_set_mesh has not succeeded");
    if (mxloption < -1 || mxloption > 1 || (mxloption != 0 && (!rmxldep || !gdepw)))
        return fail(CDFT_ERR_ARG, "cdft_heatc_compute: mxloption %d needs rmxldep and gdepw", mxloption);
    if (p->sh) {
        if (temp_space == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
        return shard::heatc_compute(*p->sh, temp, rmxldep, gdepw, mxloption, sums, out_space, as_stream(stream), p->sum_mode);
    }
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj, n3 = nij * p->npk;
    const void *dt, *dm = nullptr;
    CDFT_TRY(to_device(temp, n3 * sizeof(float), temp_space, p->st_temp, st, &dt));
    if (mxloption) {
        CDFT_TRY(to_device(rmxldep, nij * sizeof(float), temp_space, p->st_mxl, st, &dm));
        CDFT_HIP(hipMemcpyAsync(p->gdepw, gdepw, p->npk * sizeof(float), hipMemcpyHostToDevice, st));
    }
    double *dout = out_space == CDFT_MEM_HOST ? p->hsums_dev : sums;      // host results: written by the kernel into pinned memory
    HeatcArgs a;
    a.area = p->area; a.e3t = p->e3t; a.e31d = p->e31d; a.temp = (const float *)dt; a.mxl = (const float *)dm;
    a.gdepw = p->gdepw; a.mask8 = p->mask8; a.mask4 = p->mask4; a.nij = nij; a.npk = p->npk;
    a.mxlopt = mxloption; a.partial = p->partial;
    if (p->sum_mode == CDFT_HEATC_SUM_R4SEQ) {
        // the reference's sequential r4 sums, bit for bit (heatc_seq_kernel); the r4 results are widened exactly
        const size_t n = (size_t)p->npk * 3;
        CDFT_TRY(p->st_s4.reserve(n * sizeof(float)));
        const float *dcarry = nullptr;
        if (!p->carry.empty()) {
            CDFT_TRY(p->st_c4.reserve(n * sizeof(float)));
            CDFT_HIP(hipMemcpyAsync(p->st_c4.ptr, p->carry.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
            dcarry = (const float *)p->st_c4.ptr;
        }
        CDFT_TRY(launch_seq(a, dcarry, (float *)p->st_s4.ptr, st));
        p->last4.resize(n);
        CDFT_HIP(hipMemcpyAsync(p->last4.data(), p->st_s4.ptr, n * sizeof(float), hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
        p->carry.clear();
        std::vector<double> s8(n);
        for (size_t i = 0; i < n; ++i) s8[i] = (double)p->last4[i];
        if (out_space == CDFT_MEM_HOST) {
            memcpy(sums, s8.data(), n * sizeof(double));
        } else {
            CDFT_HIP(hipMemcpyAsync(sums, s8.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
            CDFT_HIP(hipStreamSynchronize(st));
        }
        return CDFT_OK;
    }
    int vec = 1;
    if (nij % 4 == 0 && aligned_to(dt, 16) && aligned_to(dm, 16)) vec = 4;
    else if (nij >= 4 && !getenv("CDFT_HEATC_NOU4")) vec = -4;           // 16-byte loads at r4 alignment (A/B hook: the old paths)
    else if (nij % 2 == 0 && aligned_to(dt, 8) && aligned_to(dm, 8)) vec = 2;
    a.nblk = vec == -4 ? (int)(((nij + 3) / 4 + 1 + kThreads * 4 - 1) / (kThreads * 4))
                       : (int)((nij / vec + kThreads * 4 - 1) / (kThreads * 4));   // kVPT = 4 vectors per lane
    if (vec == -4)     CDFT_TRY(launch_u4(a, st));
    else if (vec == 4) CDFT_TRY(launch_vec<4>(a, st));
    else if (vec == 2) CDFT_TRY(launch_vec<2>(a, st));
    else               CDFT_TRY(launch_vec<1>(a, st));
    hipLaunchKernelGGL(heatc_final_kernel, dim3(p->npk), dim3(kThreads), 0, st, p->partial, p->npk, a.nblk, dout);
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipStreamSynchronize(st));
        memcpy(sums, p->hsums, (size_t)p->npk * 3 * sizeof(double));
    }
    return CDFT_OK;
}

int cdft_heatc_set_sum_mode(cdft_heatc_plan *p, int mode) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_heatc_set_sum_mode: NULL plan");
    if (mode != CDFT_HEATC_SUM_R8 && mode != CDFT_HEATC_SUM_R4SEQ) return fail(CDFT_ERR_ARG, "cdft_heatc_set_sum_mode: mode %d", mode);
    p->sum_mode = mode;
    if (p->sh)
        for (void *q : p->sh->part) CDFT_TRY(cdft_heatc_set_sum_mode((cdft_heatc_plan *)q, mode));
    return CDFT_OK;
}

int cdft_heatc_chain(cdft_heatc_plan *p, const float *carry_in, float *last_out) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_heatc_chain: NULL plan");
    if (p->sh) return fail(CDFT_ERR_STATE, "cdft_heatc_chain: a sharded plan chains its own blocks");
    const size_t n = (size_t)p->npk * 3;
    if (last_out) {
        if (p->last4.size() != n) return fail(CDFT_ERR_STATE, "cdft_heatc_chain: no CDFT_HEATC_SUM_R4SEQ result to hand on");
        memcpy(last_out, p->last4.data(), n * sizeof(float));
    }
    if (carry_in) p->carry.assign(carry_in, carry_in + n);
    return CDFT_OK;
}

int cdft_heatc_destroy(cdft_heatc_plan *p) {
    if (!p) return CDFT_OK;
    if (p->sh) {
        (void)shard::for_blocks(*p->sh, [&](int b) { return cdft_heatc_destroy((cdft_heatc_plan *)p->sh->part[b]); });
        shard::destroy(p->sh);
        delete p;
        return CDFT_OK;
    }
    void *ptrs[] = {p->area, p->e3t, p->e31d, p->mask8, p->mask4, p->gdepw, p->partial};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (p->hsums) (void)hipHostFree(p->hsums);
    p->st_a.release(); p->st_b.release(); p->st_temp.release(); p->st_mxl.release(); p->st_out.release();
    p->st_c4.release(); p->st_s4.release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"
