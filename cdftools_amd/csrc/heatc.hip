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
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
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
    void    *h4 = nullptr, *h4_dev = nullptr;   // r4seq: the same for the r4 sums and the scan's control block (3*npk floats + ScanCtl)
    int nblk = 0;
    bool mesh_set = false;
    int sum_mode = 0;          // CDFT_HEATC_SUM_R8 | CDFT_HEATC_SUM_R4SEQ
    std::vector<float> carry;  // r4seq: the chains' start values (3*npk), empty = zeros; consumed by the next compute
    std::vector<float> last4;  // r4seq: the r4 sums of the last compute (what the next row block starts from)
    DevBuf st_a, st_b, st_temp, st_mxl, st_out, st_c4, st_s4, st_scan;
};

namespace {

constexpr int kThreads = 256;

struct HeatcArgs {
    const float *area, *e3t, *e31d, *temp, *mxl, *gdepw;
    const uint8_t *mask8;
    const float *mask4;
    int64_t nij;
    int npk, mxlopt, nblk, band, vpt;
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

constexpr int kVPT = 4;   // vectors per lane (the smaller of the two shapes): each wave reads VPT contiguous KiB-sized runs per field and level

// One workgroup owns kThreads*kVPT vectors of the horizontal plane for ALL levels: the area
// e1t*e2t (and the mixed-layer depth) stays in registers, every level streams temp / e3t / tmask
// past it.  Per level each wave reduces its three sums and parks them in LDS; at the end the waves
// are combined in wave order into partial[k][block][3].
template <int VEC, bool MASK8, bool FULL, int VPT>
__global__ void __launch_bounds__(kThreads) heatc_level_kernel(HeatcArgs a) {
    extern __shared__ double wsum[];                   // [nwaves][npk][3]
    const int64_t blk = xcd_band(a.band);              // (common.hpp)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = kThreads / 64;
    const int64_t nvec = a.nij / VEC;                  // nij % VEC == 0 by construction
    typedef typename Vec<VEC>::f FV;
    typedef typename Vec<VEC>::u UV;
    int64_t q[VPT];
    bool ok[VPT];
    float ar[VPT][VEC], mx[VPT][VEC];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        q[v] = ((blk * nwaves + wave) * VPT + v) * 64 + lane;
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
        float e3[VPT][VEC], tt[VPT][VEC], mk[VPT][VEC];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
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
        for (int v = 0; v < VPT; ++v)
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
        if (blk < a.nblk) a.partial[((int64_t)k * a.nblk + blk) * 3 + c] = s;
    }
}

// The same for planes whose size is not a multiple of 4 cells (eORCA025: 1442 x 1207) or whose arrays are not
// 16-byte aligned: 16-byte loads at r4 alignment (a level then starts anywhere), the plane covered by vectors at cells
// 0, 4, 8, ... plus, when nij is not a multiple of 4, one vector anchored at nij-4 whose leading cells repeat the
// previous vector's and count as zeros -- every load is a full vector inside the level, without a branch.
typedef unsigned char u4u __attribute__((ext_vector_type(4), aligned(1)));

template <bool MASK8, bool FULL, int VPT>
__global__ void __launch_bounds__(kThreads) heatc_level_u4_kernel(HeatcArgs a) {
    extern __shared__ double wsum[];                   // [nwaves][npk][3]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = kThreads / 64;
    const int64_t nfull = a.nij >> 2;                  // vectors 0..nfull-1 start at 4v
    const int64_t blk = xcd_band(a.band);
    int64_t e0[VPT];
    bool own[VPT][4];
    float ar[VPT][4], mx[VPT][4];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int64_t q = ((blk * nwaves + wave) * VPT + v) * 64 + lane;
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
        float e3[VPT][4], tt[VPT][4], mk[VPT][4];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
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
        for (int v = 0; v < VPT; ++v)
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
        if (blk < a.nblk) a.partial[((int64_t)k * a.nblk + blk) * 3 + c] = s;
    }
}

template <int VPT>
int launch_u4_vpt(const HeatcArgs &a, hipStream_t st) {
    dim3 grid(xcd_band_grid(a.nblk, a.band)), block(kThreads);
    const size_t shmem = (size_t)(kThreads / 64) * a.npk * 3 * sizeof(double);
    const bool m8 = a.mask8 != nullptr, full = a.e31d != nullptr;
    if (m8 && full)        hipLaunchKernelGGL((heatc_level_u4_kernel<true, true, VPT>), grid, block, shmem, st, a);
    else if (m8 && !full)  hipLaunchKernelGGL((heatc_level_u4_kernel<true, false, VPT>), grid, block, shmem, st, a);
    else if (!m8 && full)  hipLaunchKernelGGL((heatc_level_u4_kernel<false, true, VPT>), grid, block, shmem, st, a);
    else                   hipLaunchKernelGGL((heatc_level_u4_kernel<false, false, VPT>), grid, block, shmem, st, a);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}
int launch_u4(const HeatcArgs &a, hipStream_t st) { return a.vpt == 8 ? launch_u4_vpt<8>(a, st) : launch_u4_vpt<kVPT>(a, st); }

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

// ---------------------------------------------------------------------------------------------------------------
// The same chains at memory speed.  A sequential r4 sum cannot be regrouped -- but while the running sum s stays
// inside one binade [2^e, 2^(e+1)) it IS an integer sum: s = S * 2^(e-23) with S a 24-bit integer, and
//     RN(s + x) = (S + k(x)) * 2^(e-23),   k(x) = x / 2^(e-23) rounded to the nearest integer,
// as long as the result stays in the binade (the spacing of the floats around s + x is then 2^(e-23) whatever x is;
// negative s: the mirror image).  k does not depend on S unless x / 2^(e-23) falls exactly half way between two
// integers (then RN picks the even S + k: a "tie").  So a chunk of cells with no tie and no binade change moves S
// by the plain sum of its k's: order-free, computed by every lane of every wave at once.
//
// Which binade?  An ESTIMATE of the chain's value is enough to name it, and the walk below checks the name against the
// exact value before it uses a sum formed under it -- so the estimate costs correctness nothing, only a chunk resolved
// the slow way when it is wrong.  heatc_sample_kernel reads one 64-cell run of every block of 4096 cells (1/64 of the
// fields; the first 32768 cells of a level in full, where the chain grows from nothing and a crossing comes every
// few cells) and scales it to the block: r8 block sums that, added to the last exact value the level has published,
// estimate the chain at every block boundary.  The binade of the estimate where a chunk starts and, when different,
// where it ends -- or the neighbouring binade when the estimate is within a quarter of its growth since the published
// value of a power of two -- are the two HYPOTHESES (sign, e) for a chunk of 512 cells.  (Rounds 1-2 took the estimates
// from a full r8 pass of heatc_level_kernel: every field read twice.)
// Work is dealt in tickets of 64 to 256 chunks of one level from an atomic counter
// (ticket = step * npk + level: the workgroups in flight spread over all levels and a few consecutive steps of each;
// about 32 steps per level).  A workgroup forms,
// for each chunk, chain and hypothesis,  K = sum k,  pos = sum max(k,0)  and flags (a cell that is not 0; a tie / NaN
// / Inf / |k| >= 2^20; a negative x); then waits for the step before it to publish the chains' exact values.  One
// wave per chain takes the chunks 64 at a time, one per lane: with S_before from a prefix sum of the K's,
//     S_before + neg >= 2^23 (+1 when an x was negative)   and   S_before + pos < 2^24
// prove that no prefix of the chunk left the binade, and S += K is exactly what 512 sequential v_add_f32 would have
// produced; the first chunk that fails the proof, or whose hypotheses do not name the chain's binade, stops the walk.
// That chunk -- the sum crosses a power of two (some 25 times per chain), a tie (about as rare on real fields), the
// first cells of the chain (s = 0, or smaller than x) -- is read again and resolved cell by cell, still in parallel:
// per-cell prefix sums find the first cell that breaks the rule, the cells before it are added as integers, that one
// cell with a v_add_f32, and the rest again under the new binade (a chunk that keeps breaking -- every cell a tie --
// is finished by a plain sequential loop).  A chunk of zeros (land) is the identity under any state.  Since the step
// a workgroup waits for was dealt before its own, it is held by a workgroup that already runs: no deadlock whatever
// else shares the GPU; a wait that lasts over a second raises an error flag instead of hanging.  Bit-identical to
// heatc_seq_kernel (tests/test_heatc.py compares the two on adversarial fields and at size).
// Since round 6 that paragraph describes the form without WALKERS (CDFT_HEATC_SCAN_WALKERS=0, and launches of more
// levels than a quarter of the chip's workgroup slots): by default the workgroup that formed a ticket's sums does not
// wait and does not walk -- it parks the sums in HBM for the level's walker workgroup and draws the next ticket; see the
// kernel.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kScanWaves = 4, kScanNV = 2, kScanChunk = kScanNV * 256, kScanSub = kScanWaves * kScanChunk;
#ifndef CDFT_SCAN_AHEAD
#define CDFT_SCAN_AHEAD 3
#endif
#ifndef CDFT_SCAN_WGS
#define CDFT_SCAN_WGS 3
#endif
constexpr int kScanAhead = CDFT_SCAN_AHEAD;                       // sub-blocks of loads in flight ahead of the one being summed: 2 or 3
constexpr int kScanMaxChunks = 256;                               // chunks per ticket: 64, 128 or 256 (32768 .. 131072 cells)

// the sum over the 64 lanes, in every lane: six DPP adds in the vector pipeline (quad swaps, half-row and row mirrors give
// every lane its row of 16; row_bcast:15 and :31 carry the rows forward to lane 63), not six ds_bpermute round trips
__device__ __forceinline__ int wave_isum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, true);      // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, true);      // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);     // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);     // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
// the same sum, valid in lane 63 only (what a lane that stores it needs: no v_readlane / v_mov round trip)
__device__ __forceinline__ int wave_isum_top(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
    return v;
}
// two r4 operations per instruction (v_pk_mul_f32 / v_pk_add_f32: each half the plain IEEE operation, no contraction)
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v pk_mul(f2v a, f2v b) { return a * b; }
__device__ __forceinline__ f2v pk_add(f2v a, f2v b) { return a + b; }
__device__ __forceinline__ f2v pk_sub(f2v a, f2v b) { return a - b; }
// the minimum over the 64 lanes, in every lane: the same six DPP steps as wave_isum (INT_MAX where a step brings nothing)
__device__ __forceinline__ int wave_imin(int v) {
    const int big = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0xb1, 0xf, 0xf, false));     // quad_perm:[1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x4e, 0xf, 0xf, false));     // quad_perm:[2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x141, 0xf, 0xf, false));    // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x140, 0xf, 0xf, false));    // row_mirror
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false));    // row_bcast:15 into rows 1 and 3
    v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false));    // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum over the lanes: row_shr:1,2,4,8 inside the rows of 16 (a lane without a source adds 0), then the
// rows' totals carried forward by row_bcast:15 and :31 -- six vector instructions instead of six ds_bpermute round trips
// (the walk does one of these per 64 chunks, the cell-by-cell resolution two per round)
__device__ __forceinline__ int wave_iscan_incl(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);             // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);             // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);             // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);             // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);             // row_bcast:15: rows 1 and 3 take the total of the row before
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);             // row_bcast:31: rows 2 and 3 take the total of rows 0-1
    return v;
}

// What a level has published: word[ch] = (number of tickets done) << 32 | bits of chain ch's value there -- tag and
// value in ONE 64-bit atomic word, so a reader never sees a value that does not belong to the step it read.
struct ScanLevel { unsigned long long word[3]; unsigned long long pad; };
__device__ __forceinline__ unsigned long long scan_word(int step, float s) { return ((unsigned long long)(uint32_t)step << 32) | __float_as_uint(s); }
constexpr int kScanMaxLanes = 8;
struct ScanCtl   { int ticket; int error; int nfallback, nunguessed, nslow, nstolen; int lane_next[kScanMaxLanes];   // (nfallback ...: statistics)
                   // round 6, 100 MHz ticks summed over the walker waves (wk_*: waiting for a parked ticket, loading its sums, walking it)
                   // and over the streaming workgroups (st_*: waiting for a ring slot, parking, the whole ticket); t_first / t_last: the kernel's span
                   unsigned long long wk_wait, wk_load, wk_walk, st_ring, st_park, st_all, t_first, t_last; };

// the binade of a chain value: sign and exponent, or -1 when the integer picture does not apply (0, denormal, Inf, NaN,
// or an exponent whose ulp does not scale exactly)
__device__ __forceinline__ int scan_key(float s) {
    const uint32_t b = __float_as_uint(s);
    const int ex = (int)((b >> 23) & 0xffu);
    return (ex >= 27 && ex <= 247) ? (int)(b >> 23) : -1;
}
__device__ __forceinline__ int   scan_mant(float s)         { return (int)((__float_as_uint(s) & 0x7fffffu) | 0x800000u); }
__device__ __forceinline__ float scan_value(int key, int S) { return __uint_as_float(((uint32_t)key << 23) | ((uint32_t)S & 0x7fffffu)); }
__device__ __forceinline__ float scan_scale(int key) {            // +-2^(23-e): x * scale = x / ulp(s), with the sign of s folded in
    return __uint_as_float(((uint32_t)(key & 0x100) << 23) | ((uint32_t)(23 + 254 - (key & 0xff)) << 23));
}

// partial[k][b][ch], b over blocks of kSampleBlock cells: r8 sum of the chain's terms over the block -- of all its cells when the
// block lies in the level's first `dense_cells` cells, else of its first 64 cells scaled by (cells of the block) / 64.
// One wave per block.
constexpr int kSampleBlock = 4096;
// One wave per block: the blocks that are read in full (the grid covers the first `nb` blocks of every level).
template <bool MASK8, bool FULL, bool MXL>
__global__ void __launch_bounds__(256) heatc_sample_kernel(HeatcArgs a, int64_t dense_cells, int nb) {
    const int lane = threadIdx.x & 63, k = blockIdx.y;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nb) return;
    const int64_t c0 = b * kSampleBlock, c1 = min(a.nij, c0 + kSampleBlock), base = (int64_t)k * a.nij;
    const bool dense = c0 < dense_cells;
    const int64_t cend = dense ? c1 : min(c1, c0 + 64);
    const float e3k = FULL ? a.e31d[k] : 0.f;
    const float dw  = MXL ? a.gdepw[k] : 0.f;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 4
    for (int64_t p = c0 + lane; p < cend; p += 64) {
        const float ar = a.area[p], tt = ld_stream(a.temp + base + p);
        const float mk = MASK8 ? (float)ld_stream(a.mask8 + base + p) : ld_stream(a.mask4 + base + p);
        float e = FULL ? e3k : ld_stream(a.e3t + base + p);
        if (MXL) {
            if (a.mxlopt == 1) e = fmaxf(0.f, fminf(e, __fsub_rn(a.mxl[p], dw)));
            else               e = fminf(e, fmaxf(0.f, __fsub_rn(__fadd_rn(dw, e), a.mxl[p])));
        }
        const float pv = __fmul_rn(ar, e);
        s0 += (double)__fmul_rn(ar, mk);
        s1 += (double)__fmul_rn(pv, mk);
        s2 += (double)__fmul_rn(__fmul_rn(pv, tt), mk);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
        const double f = dense ? 1.0 : (double)(c1 - c0) / (double)(cend - c0);
        double *o = a.partial + ((int64_t)k * a.nblk + b) * 3;
        // (a NaN or Inf in the sample would poison every estimate after it: such a block counts as 0 -- its chunks are
        // resolved cell by cell whatever the hypotheses say)
        o[0] = isfinite(s0) ? s0 * f : 0.0; o[1] = isfinite(s1) ? s1 * f : 0.0; o[2] = isfinite(s2) ? s2 * f : 0.0;
    }
}

// The blocks read in full (b < b0: one workgroup each, 16 cells per thread) and the sampled ones (b >= b0): four lanes per block, each 16 of its first 64 cells as four 16-byte loads per field (a load
// instruction takes 64 contiguous bytes of a block), a wave
// 16 blocks -- one wave per block (above) was 286 000 waves of one load each at eORCA12: 92 us, bound by their number.
// The order of a block's 64 terms differs from the kernel above (4 lanes x 16 cells, then across the lanes): an estimate
// either way, fixed for a given grid.
template <bool MASK8, bool FULL, bool MXL>
__global__ void __launch_bounds__(256) heatc_sample16_kernel(HeatcArgs a, int b0) {
    __shared__ double scratch[16];
    const int lane = threadIdx.x & 63, k = blockIdx.y, sub = lane & 3;
    // workgroups 0 .. b0-1: one each for the blocks read in full (16 cells per thread); the others: 64 sampled blocks each
    const bool dense = (int)blockIdx.x < b0;
    const int64_t b = dense ? blockIdx.x : b0 + ((int64_t)(blockIdx.x - b0) * 4 + (threadIdx.x >> 6)) * 16 + (lane >> 2);
    const bool live = b < a.nblk;
    const int64_t bc = live ? b : a.nblk - 1;
    const int64_t c0 = bc * kSampleBlock, c1 = min(a.nij, c0 + kSampleBlock), cend = dense ? c1 : min(c1, c0 + 64), base = (int64_t)k * a.nij;
    const float e3k = FULL ? a.e31d[k] : 0.f;
    const float dw  = MXL ? a.gdepw[k] : 0.f;
    f4u ar[4], tt[4], e3[4], mx[4];
    uint32_t mb[4];
    f4u mf[4];
    int64_t q[4];
    typedef uint32_t u32u __attribute__((aligned(1)));
#pragma unroll
    for (int v = 0; v < 4; ++v) {                      // every load at an address inside the level (clamped), masked afterwards
        q[v] = min(c0 + (dense ? (int64_t)(v * 256 + threadIdx.x) * 4 : v * 16 + sub * 4), a.nij - 4);    // (contiguous bytes per load: 4 KB of a whole block, 64 of a sampled one)
        if (a.nij < 4) q[v] = 0;
        ar[v] = *reinterpret_cast<const f4u *>(a.area + q[v]);
        tt[v] = ld_stream4u(a.temp + base + q[v]);
        if (!FULL) e3[v] = ld_stream4u(a.e3t + base + q[v]);
        if (MXL) mx[v] = *reinterpret_cast<const f4u *>(a.mxl + q[v]);
        if (MASK8) mb[v] = __builtin_nontemporal_load(reinterpret_cast<const u32u *>(a.mask8 + base + q[v]));
        else       mf[v] = ld_stream4u(a.mask4 + base + q[v]);
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int64_t want = c0 + (dense ? (int64_t)(v * 256 + threadIdx.x) * 4 : v * 16 + sub * 4);    // the first cell this vector stands for (its address may have been clamped below it)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t p = q[v] + c;
            const bool mine = p >= want && p < cend;
            const float arc = ar[v][c], ttc = tt[v][c];
            const float mk = MASK8 ? (float)((mb[v] >> (8 * c)) & 0xffu) : mf[v][c];
            float e = FULL ? e3k : e3[v][c];
            if (MXL) {
                if (a.mxlopt == 1) e = fmaxf(0.f, fminf(e, __fsub_rn(mx[v][c], dw)));
                else               e = fminf(e, fmaxf(0.f, __fsub_rn(__fadd_rn(dw, e), mx[v][c])));
            }
            const float pv = __fmul_rn(arc, e);
            const double t0 = (double)__fmul_rn(arc, mk), t1 = (double)__fmul_rn(pv, mk), t2 = (double)__fmul_rn(__fmul_rn(pv, ttc), mk);
            s0 += mine ? t0 : 0.0; s1 += mine ? t1 : 0.0; s2 += mine ? t2 : 0.0;
        }
    }
    if (dense) {                                       // (uniform over the workgroup)
        s0 = block_sum(s0, scratch); s1 = block_sum(s1, scratch); s2 = block_sum(s2, scratch);
    } else {                                           // the four lanes of a block (a quad)
        s0 = dpp_add<0xb1, 0xf>(s0); s0 = dpp_add<0x4e, 0xf>(s0);
        s1 = dpp_add<0xb1, 0xf>(s1); s1 = dpp_add<0x4e, 0xf>(s1);
        s2 = dpp_add<0xb1, 0xf>(s2); s2 = dpp_add<0x4e, 0xf>(s2);
    }
    if (live && (dense ? threadIdx.x == 0 : sub == 0)) {
        const double f = (double)(c1 - c0) / (double)(cend - c0);
        double *o = a.partial + ((int64_t)k * a.nblk + b) * 3;
        // (a NaN or Inf in the sample would poison every estimate after it: such a block counts as 0 -- its chunks are
        // resolved cell by cell whatever the hypotheses say)
        o[0] = isfinite(s0) ? s0 * f : 0.0; o[1] = isfinite(s1) ? s1 * f : 0.0; o[2] = isfinite(s2) ? s2 * f : 0.0;
    }
}

// est[k][b][ch], b = 0..nblk: the r8 value of chain ch of level k where block b of the level kernel starts (b = nblk:
// where the level ends), from its partials and the carry.  One workgroup of kEstThreads per level: a thread adds up a run
// of blocks, the runs are scanned across the workgroup, the thread walks its run again (one wave per level walked runs of
// 60 blocks at eORCA12: 48 us).
constexpr int kEstThreads = 1024;
__global__ void __launch_bounds__(kEstThreads) heatc_scan_est_kernel(const double *__restrict__ partial, int nblk, const float *__restrict__ carry,
                                                                     double *__restrict__ est, ScanLevel *lev, ScanCtl *ctl,
                                                                     int *__restrict__ parked, int nsteps) {
    __shared__ double wtot[3][kEstThreads / 64];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *ek = est + (int64_t)k * (nblk + 1) * 3;
    const double *pk = partial + (int64_t)k * nblk * 3;
    const int seg = (nblk + kEstThreads - 1) / kEstThreads, b0 = min(nblk, tid * seg), b1 = min(nblk, b0 + seg);   // this thread's blocks
    double sum[3] = {0.0, 0.0, 0.0}, excl[3];
    for (int b = b0; b < b1; ++b)
        for (int c = 0; c < 3; ++c) sum[c] += pk[(int64_t)b * 3 + c];
    for (int c = 0; c < 3; ++c) {
        double incl = sum[c];                                     // the sums of the lanes before this one
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        excl[c] = incl - sum[c];
        if (lane == 63) wtot[c][wave] = incl;
    }
    __syncthreads();
    for (int c = 0; c < 3; ++c) {
        double run = (double)(carry ? carry[k * 3 + c] : 0.f);
        for (int w = 0; w < wave; ++w) run += wtot[c][w];         // the waves before this one, in order
        run += excl[c];
        if (tid == 0) ek[c] = run;
        for (int b = b0; b < b1; ++b) { run += pk[(int64_t)b * 3 + c]; ek[(int64_t)(b + 1) * 3 + c] = run; }
    }
    if (tid < 3) lev[k].word[tid] = scan_word(0, carry ? carry[k * 3 + tid] : 0.f);
    if (parked) for (int i = tid; i < nsteps; i += kEstThreads) parked[(int64_t)k * nsteps + i] = 0;     // no ticket of the level is parked yet
    if (k == 0 && tid == 0) {
        ctl->ticket = 0; ctl->error = 0; ctl->nfallback = 0; ctl->nunguessed = 0; ctl->nslow = 0; ctl->nstolen = 0;
        ctl->wk_wait = ctl->wk_load = ctl->wk_walk = ctl->st_ring = ctl->st_park = ctl->st_all = ctl->t_last = 0ull; ctl->t_first = ~0ull;
        for (int x = 0; x < kScanMaxLanes; ++x) ctl->lane_next[x] = 0;
    }
}

template <bool MASK8, bool FULL, bool MXL>
// (four workgroups per CU -- 128 VGPRs, no spill in the byte-mask form -- were measured in round 4: no faster, 2.27 against 2.20 ms)
__global__ void __launch_bounds__(64 * kScanWaves, CDFT_SCAN_WGS) heatc_scan_kernel(HeatcArgs a, ScanLevel *__restrict__ lev, ScanCtl *__restrict__ ctl,
                                                                     const double *__restrict__ est, int block_cells, int nchunk, int nsteps,
                                                                     float *__restrict__ sums4, long long scan_wait_ticks, int nlanes, int align, int dbg,
                                                                     int *__restrict__ parked, int *__restrict__ park, int ring) {
    __shared__ int sm_K[2][kScanMaxChunks][3], sm_pos[2][kScanMaxChunks][3], sm_flag[2][kScanMaxChunks][3], sm_key[2][kScanMaxChunks][3];
    __shared__ int tick;
    __shared__ __attribute__((aligned(16))) float xs[3][kScanChunk];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (w in a scalar register: what depends on it alone stays scalar)
    const int64_t ticket_cells = (int64_t)nchunk * kScanChunk;

    // Round 6: WALKERS.  With `ring` > 0 the first npk workgroups stream nothing: workgroup k is the chains of level k -- wave ch
    // takes chain ch through the level's tickets in order, from the sums the streaming workgroups have PARKED in HBM
    // (park: `ring` slots per level, a ticket in slot step % ring; parked[k * nsteps + step] = 1 once its sums are there).  A
    // streaming workgroup never waits for a chain any more (round 5: a third of the workgroup slots of an eORCA025 run were held
    // by tickets waiting for their start value while 21 cell-by-cell resolutions per level went by, profiles/r04/
    // heatc_scan_timeline_eorca025.txt); it waits only when its ring slot is still unread, i.e. when a level's chains are
    // `ring` tickets behind.  No deadlock: a walker waits for tickets that were dealt (held by workgroups that run and wait, if
    // at all, for a walker to pass tickets parked long before); the walkers, first in the grid, are resident before any of them.
    // Measured (profiles/r06/README.md): 0.50 -> 0.38-0.40 ms per call at eORCA025, 2.10-2.17 -> 2.12-2.26 at eORCA12 (there the
    // streaming itself, 39 vector instructions a cell beside 9.1 bytes, is the whole time either way).  What it took, each a
    // measured step: the walker's loads of a ticket's sums all issued before the first is used (the compiler waits for an
    // atomic load where its value is needed: 32 round trips a ticket, 25 us); no agent-scope ACQUIRE in the polling loop (an
    // L2 invalidation each: it took the e1t*e2t plane from the streaming workgroups, 2.1 -> 3.1 ms); no RELEASE on the flag
    // (a write-back of the L2 per ticket: 2.1 -> 2.4 ms; vmcnt(0) in every wave before the barrier instead).  Tried and not
    // kept: a chunk's sums stored to HBM as they are formed (a wave's loads and stores share one in-order counter: 2.75 ms);
    // the next ticket drawn while this one streams (a ticket held for a whole ticket's time before it is begun: 0.39 -> 0.44 ms
    // at eORCA025); the first loads requested before the hypotheses are formed (no difference).
    const bool walker = ring > 0 && (int)blockIdx.x < a.npk;
    if (walker && w >= 3) return;                                 // (a walker has no workgroup barrier)
    const int park_ints = 24 * nchunk;                            // a ticket's slot: [chain][4 64-bit words][chunk] -- (K, pos), (flag, key) of each hypothesis

    // the XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0)
    const int my_lane = nlanes > 1 ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) % nlanes : 0;

    struct Raw { float ar[4], e3[4], tt[4], mk[4], mx[4]; uint32_t mkb; };      // (mkb: the four mask bytes as loaded, MASK8)
    // The ticket at hand (set_ticket): what the lambdas below read.
    // A level is walked in VIRTUAL cells: `pad` cells that do not exist (terms of -0.0, the one value that changes no sum,
    // not even a sum of -0.0) stand before its first one, so that every chunk of 512 starts on a 128-byte line of temp,
    // e3t and the byte mask alike whatever k * nij is -- a line split between two chunks is fetched by both (two waves,
    // two moments): 0.35 B per cell of 9.1 at eORCA12, where a level starts 48 bytes into a line.
    int k = 0, step = 0, pad = 0;
    int64_t base = 0, t0 = 0, nijv = 0;
    float e3k = 0.f, dw = 0.f;
    auto set_ticket = [&](int kk, int ss) {
        k = kk; step = ss;
        base = (int64_t)k * a.nij; t0 = (int64_t)step * ticket_cells;
        pad = align ? (int)((base + (int64_t)((reinterpret_cast<uintptr_t>(a.temp) >> 2) & 127u)) & 127) : 0;
        nijv = a.nij + pad;
        e3k = FULL ? a.e31d[k] : 0.f;
        dw  = MXL ? a.gdepw[k] : 0.f;
    };
    auto fetch = [&](Raw (&raw)[kScanNV], int64_t c0) {       // 512 cells from c0 on: lane l takes cells 4l..4l+3 of each 256
        if (c0 >= pad && c0 + kScanChunk <= nijv) {           // (uniform) every cell exists: four 16-byte loads per field and lane, nothing else
            // wave-uniform bases (scalar registers) + one 32-bit lane offset: the loads need no 64-bit vector address arithmetic
            const int64_t u0 = c0 - pad;
            const float *ap = a.area + u0, *tp = a.temp + base + u0, *ep = FULL ? nullptr : a.e3t + base + u0;
            const float *mp4 = MASK8 ? nullptr : a.mask4 + base + u0, *xp = MXL ? a.mxl + u0 : nullptr;
            const uint8_t *mp8 = MASK8 ? a.mask8 + base + u0 : nullptr;
#pragma unroll
            for (int v = 0; v < kScanNV; ++v) {
                Raw &x = raw[v];
                const unsigned p = (unsigned)(v * 256 + 4 * lane);
                const f4u av = *reinterpret_cast<const f4u *>(ap + p);
                x.ar[0] = av.x; x.ar[1] = av.y; x.ar[2] = av.z; x.ar[3] = av.w;
                const f4u tv = ld_stream4u(tp + p);
                x.tt[0] = tv.x; x.tt[1] = tv.y; x.tt[2] = tv.z; x.tt[3] = tv.w;
                if (!FULL) { const f4u ev = ld_stream4u(ep + p); x.e3[0] = ev.x; x.e3[1] = ev.y; x.e3[2] = ev.z; x.e3[3] = ev.w; }
                else       { x.e3[0] = x.e3[1] = x.e3[2] = x.e3[3] = 0.f; }
                if (MASK8) { typedef uint32_t u32u __attribute__((aligned(1))); x.mkb = __builtin_nontemporal_load(reinterpret_cast<const u32u *>(mp8 + p)); }
                else       { const f4u m = ld_stream4u(mp4 + p); x.mk[0] = m.x; x.mk[1] = m.y; x.mk[2] = m.z; x.mk[3] = m.w; }
                if (MXL) { const f4u mv = *reinterpret_cast<const f4u *>(xp + p); x.mx[0] = mv.x; x.mx[1] = mv.y; x.mx[2] = mv.z; x.mx[3] = mv.w; }
                else     { x.mx[0] = x.mx[1] = x.mx[2] = x.mx[3] = 0.f; }
            }
            return;
        }
#pragma unroll
        for (int v = 0; v < kScanNV; ++v) {                   // the first and the last chunk of a level: cell by cell, zeros where there is none
            Raw &x = raw[v];
            const int64_t p = c0 + v * 256 + 4 * lane - pad;
            x.mkb = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x.ar[c] = x.e3[c] = x.tt[c] = x.mk[c] = x.mx[c] = 0.f;
                if (p + c >= 0 && p + c < a.nij) {
                    x.ar[c] = a.area[p + c];
                    x.tt[c] = a.temp[base + p + c];
                    if (!FULL) x.e3[c] = a.e3t[base + p + c];
                    if (MASK8) x.mkb |= (uint32_t)a.mask8[base + p + c] << (8 * c);
                    else       x.mk[c] = a.mask4[base + p + c];
                    if (MXL) x.mx[c] = a.mxl[p + c];
                }
            }
        }
    };
    // the three r4 products of every cell in `raw`, exactly as the reference forms them (:225-229); zeros past the end.
    // Two cells per instruction (v_pk_mul_f32: the same IEEE r4 product in each half): x[chain][vector][pair]
    auto products = [&](const Raw (&raw)[kScanNV], int64_t c0, f2v (&x)[3][kScanNV][2]) {
        const bool whole = c0 >= pad && c0 + kScanChunk <= nijv;     // (uniform) every cell of the chunk exists
#pragma unroll
        for (int v = 0; v < kScanNV; ++v) {
            const Raw &r = raw[v];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float e[2], mk[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = 2 * p + h;
                    e[h] = FULL ? e3k : r.e3[c];
                    if (MXL) {
                        if (a.mxlopt == 1) e[h] = fmaxf(0.f, fminf(e[h], __fsub_rn(r.mx[c], dw)));                   // :220
                        else               e[h] = fminf(e[h], fmaxf(0.f, __fsub_rn(__fadd_rn(dw, e[h]), r.mx[c])));  // :222
                    }
                    mk[h] = MASK8 ? (float)((r.mkb >> (8 * c)) & 0xffu) : r.mk[c];
                }
                const f2v ar = {r.ar[2 * p], r.ar[2 * p + 1]}, tt = {r.tt[2 * p], r.tt[2 * p + 1]};
                const f2v e2 = {e[0], e[1]}, mk2 = {mk[0], mk[1]};
                const f2v pv = pk_mul(ar, e2);
                x[0][v][p] = pk_mul(ar, mk2);
                x[1][v][p] = pk_mul(pv, mk2);
                x[2][v][p] = pk_mul(pk_mul(pv, tt), mk2);
            }
        }
        if (!whole) {                                         // (uniform: only the first and the last chunk of a level take this branch)
#pragma unroll
            for (int v = 0; v < kScanNV; ++v)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t id = c0 + v * 256 + 4 * lane + c;
                    if (id < pad || id >= nijv)
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) { if (c & 1) x[ch][v][c >> 1].y = -0.f; else x[ch][v][c >> 1].x = -0.f; }
                }
        }
    };
    // sums of chunk q under the hypotheses of each chain
    auto summarize = [&](const f2v (&x)[3][kScanNV][2], int q) {
        // Is there a cell with a term that is not 0?  One answer PER CHAIN: a chain that is still at 0 (no binade) has to look at
        // every chunk that says yes -- with -mxloption the volume and heat terms are 0 below the mixed layer while the surface
        // terms are not, and one answer for the three sent every chunk of 68 levels x 2 chains to the cells (12.5 ms at eORCA025).
        uint32_t any[3] = {0u, 0u, 0u};
#pragma unroll
        for (int v = 0; v < kScanNV; ++v)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) any[ch] |= __float_as_uint(x[ch][v][p].x) | __float_as_uint(x[ch][v][p].y);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const bool anz = __builtin_amdgcn_ballot_w64((any[ch] & 0x7fffffffu) != 0u) != 0;
            for (int h = 0; h < 2; ++h) {
                const int key = sm_key[h][q][ch];
                if (h == 1 && key < 0) break;                 // (uniform)
                int Kl = 0, Al = 0;                           // this lane's sum of k and of |k| (chain 2; the others: Al = Kl)
                bool bad = key < 0;
                uint32_t sgn = 0u;
                if (key >= 0 && anz) {
                    const float scale = scan_scale(key);
                    const f2v sc = {scale, scale};
                    // A lane's eight k's are added as floats, two at a time (v_pk_add_f32).  The guard is the lane's sum of
                    // |k|: while it stays below 2^23 every partial sum is an integer below 2^24, hence exact, and the chunk's
                    // 64 lane sums fit 32 bits with room for the walk's prefix sums; a sum that reaches 2^23 or is not a
                    // number (a NaN / Inf term, or an x far above the chain's ulp) sends the chunk to the cells.  The surface
                    // and the volume have no negative term on a real mesh: their sum of |k| IS their sum of k, and a sign
                    // bit anywhere (a negative weight of -M, a negative start value) sends the chunk to the cells as well.
                    f2v Kf = {0.f, 0.f};
                    float Af = 0.f, far = 0.f;                // far: the largest |q - rint(q)|, 0.5 exactly = a tie
#pragma unroll
                    for (int v = 0; v < kScanNV; ++v)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const f2v qq = pk_mul(x[ch][v][p], sc);                       // exact (a power of two), or Inf / NaN
                            const f2v kf = {rintf(qq.x), rintf(qq.y)};
                            Kf = pk_add(Kf, kf);
                            const f2v d = pk_sub(qq, kf);                                 // exact
                            far = fmaxf(far, fmaxf(fabsf(d.x), fabsf(d.y)));              // (v_max3_f32)
                            sgn |= __float_as_uint(qq.x) | __float_as_uint(qq.y);         // (v_or3_b32)
                            if (ch == 2) { Af = __fadd_rn(Af, fabsf(kf.x)); Af = __fadd_rn(Af, fabsf(kf.y)); }
                        }
                    const float Ks = __fadd_rn(Kf.x, Kf.y);
                    if (ch != 2) Af = Ks;
                    bad = far == 0.5f || !(Af < 8388608.f) || (ch != 2 && (sgn >> 31) != 0u);
                    Kl = bad ? 0 : (int)Ks;
                    Al = bad ? 0 : (int)Af;
                }
                // the chunk's sums: in lane 63 of the accumulating registers, which writes them (no trip through a scalar)
                const int Kw = wave_isum_top(Kl);
                const int Aw = ch == 2 ? wave_isum_top(Al) : Kw;
                const bool abad = __builtin_amdgcn_ballot_w64(bad) != 0, ang = __builtin_amdgcn_ballot_w64((sgn >> 31) != 0u) != 0;
                if (lane == 63) {
                    sm_K[h][q][ch] = Kw; sm_pos[h][q][ch] = ch == 2 ? (Aw + Kw) / 2 : Kw;      // pos = sum of max(k,0) = (sum|k| + sum k) / 2
                    sm_flag[h][q][ch] = (anz ? 1 : 0) | (abad ? 2 : 0) | (ang ? 4 : 0);
                }
            }
        }
    };
    // chain ch through the ticket whose sums are in LDS, from the value s where the ticket starts: the value where it ends
    auto walk_chain = [&](const int ch, float s) __attribute__((always_inline)) -> float {
            int from = 0;
            for (;;) {
                // the chunks in order, 64 at a time, one per lane
                const int ks = scan_key(s);
                // (-0) + (+0) = +0: a chain handed a start value of -0.0 is the one state a zero term can change; it looks
                // at every chunk until a term that is not -0.0 has made it +0.0 or a number (no chain gets back to -0.0)
                const bool minus0 = __float_as_uint(s) == 0x80000000u;
                int S0 = scan_mant(s), stop = -1, hstop = 0;
                for (int g = from & ~63; g < nchunk && stop < 0; g += 64) {
                    const int q = g + lane;
                    const int f0 = sm_flag[0][q][ch];
                    const bool live = q >= from && q < nchunk && ((f0 & 1) || (minus0 && t0 + (int64_t)q * kScanChunk < nijv));   // zeros: s + 0 = s
                    const int h = (ks >= 0 && ks == sm_key[0][q][ch]) ? 0 : (ks >= 0 && ks == sm_key[1][q][ch]) ? 1 : -1;
                    const int hh = h < 0 ? 0 : h;
                    const int f = sm_flag[hh][q][ch];
                    const int K = (live && h >= 0) ? sm_K[hh][q][ch] : 0, pos = (live && h >= 0) ? sm_pos[hh][q][ch] : 0;
                    const int incl = wave_iscan_incl(K, lane), Sb = S0 + incl - K;
                    const bool stopq = live && (h < 0 || (f & 2) || Sb + pos > (1 << 24) - 1 || Sb + (K - pos) < (1 << 23) + ((f & 4) ? 1 : 0));
                    const uint64_t m = __builtin_amdgcn_ballot_w64(stopq);
                    const int sl = m ? (int)__builtin_ctzll(m) : 64;             // (uniform)
                    S0 += sl == 0 ? 0 : __shfl(incl, sl - 1, 64);
                    if (sl < 64) { stop = g + sl; hstop = __shfl(h, sl, 64); }
                }
                if (ks >= 0) s = scan_value(ks, S0);
                if (stop < 0 || (dbg & 4)) break;
                if (lane == 0 && ks >= 0 && hstop < 0) atomicAdd(&ctl->nunguessed, 1);
                // the chunk that stopped the walk: read again and resolved cell by cell
                const int q = stop;
                const int64_t c0 = t0 + (int64_t)q * kScanChunk;
                const int nvalid = (int)max((int64_t)0, min((int64_t)kScanChunk, nijv - c0));
                Raw raw[kScanNV];
                f2v x3[3][kScanNV][2];
                fetch(raw, c0);
                products(raw, c0, x3);
                float xv[kScanNV][4];
                auto take = [&](const f2v (&xc)[kScanNV][2]) {         // (a branch per chain, each with constant indices: nothing in scratch)
#pragma unroll
                    for (int v = 0; v < kScanNV; ++v)
#pragma unroll
                        for (int c = 0; c < 4; ++c) xv[v][c] = (c & 1) ? xc[v][c >> 1].y : xc[v][c >> 1].x;
                };
                if (ch == 0) take(x3[0]); else if (ch == 1) take(x3[1]); else take(x3[2]);
                if (lane == 0) atomicAdd(&ctl->nfallback, 1);
                int next = 0, rounds = 0;                         // cells [0, next) of the chunk are in s
                while (next < nvalid && rounds < 6) {
                    ++rounds;
                    const int ks = scan_key(s);
                    int first;                                    // the first cell >= next that cannot be added as an integer
                    int incl[kScanNV][4];
                    if (ks < 0) {
                        // no binade (s = 0, ...): the next cell that is not 0 (s = -0.0: that is not -0.0) is added with v_add_f32
                        const bool m0 = __float_as_uint(s) == 0x80000000u;
                        int mine = kScanChunk;
#pragma unroll
                        for (int v = kScanNV - 1; v >= 0; --v)
#pragma unroll
                            for (int c = 3; c >= 0; --c) {
                                const int id = v * 256 + 4 * lane + c;
                                if (id >= next && (m0 ? __float_as_uint(xv[v][c]) != 0x80000000u : xv[v][c] != 0.f)) mine = id;
                            }
                        first = wave_imin(mine);
                    } else {
                        const float scale = scan_scale(ks);
                        const int S0 = scan_mant(s);
                        int run = 0, mine = kScanChunk;
#pragma unroll
                        for (int v = 0; v < kScanNV; ++v) {
                            int kk[4], loc = 0;
                            bool bad[4], ngq[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int id = v * 256 + 4 * lane + c;
                                const float qq = __fmul_rn(xv[v][c], scale), kf = rintf(qq);
                                kk[c] = id >= next ? (int)kf : 0;
                                bad[c] = id >= next && (fabsf(__fsub_rn(qq, kf)) == 0.5f || !(fabsf(qq) < 1048576.f));
                                ngq[c] = qq < 0.f;
                                if (bad[c]) kk[c] = 0;
                                loc += kk[c];
                            }
                            const int lincl = wave_iscan_incl(loc, lane);
                            int before = run + lincl - loc;       // sum of the k's of every cell before this lane's four, from `next` on
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int id = v * 256 + 4 * lane + c, after = S0 + before + kk[c];
                                const bool ok = !bad[c] && after <= (1 << 24) - 1 && after >= (1 << 23) + (ngq[c] ? 1 : 0);
                                if (id >= next && !ok && id < mine) mine = id;
                                before += kk[c];
                                incl[v][c] = before;
                            }
                            run += __shfl(lincl, 63, 64);
                        }
                        first = wave_imin(mine);
                        // the cells [next, first) as integers: the prefix sum just before `first`
                        const int upto = min(first, nvalid);
                        if (upto > next) {
                            const int id = upto - 1, v = id >> 8, l = (id & 255) >> 2, c = id & 3;
                            int val = 0;
#pragma unroll
                            for (int vv = 0; vv < kScanNV; ++vv)
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc) if (vv == v && cc == c) val = incl[vv][cc];
                            s = scan_value(ks, S0 + __shfl(val, l, 64));
                        }
                    }
                    if (first < nvalid) {                         // that one cell: a plain r4 add
                        const int v = first >> 8, l = (first & 255) >> 2, c = first & 3;
                        float val = 0.f;
#pragma unroll
                        for (int vv = 0; vv < kScanNV; ++vv)
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) if (vv == v && cc == c) val = xv[vv][cc];
                        s = __fadd_rn(s, __shfl(val, l, 64));
                        next = first + 1;
                    } else {
                        next = nvalid;
                    }
                }
                if (next < nvalid) {                              // a chunk that keeps breaking (every cell a tie ...): the plain loop
#pragma unroll
                    for (int v = 0; v < kScanNV; ++v)
                        *reinterpret_cast<f4a *>(&xs[ch][v * 256 + 4 * lane]) = f4a{xv[v][0], xv[v][1], xv[v][2], xv[v][3]};
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane == 0) {
                        atomicAdd(&ctl->nslow, 1);
                        const float *xp = xs[ch];
                        int i = next;
                        for (; i < nvalid && (i & 3); ++i) s = __fadd_rn(s, xp[i]);
                        for (; i + 16 <= nvalid; i += 16) {       // 16 values on their way from LDS while 16 are added
                            const f4a q0 = *reinterpret_cast<const f4a *>(xp + i),     q1 = *reinterpret_cast<const f4a *>(xp + i + 4),
                                      q2 = *reinterpret_cast<const f4a *>(xp + i + 8), q3 = *reinterpret_cast<const f4a *>(xp + i + 12);
                            s = __fadd_rn(s, q0.x); s = __fadd_rn(s, q0.y); s = __fadd_rn(s, q0.z); s = __fadd_rn(s, q0.w);
                            s = __fadd_rn(s, q1.x); s = __fadd_rn(s, q1.y); s = __fadd_rn(s, q1.z); s = __fadd_rn(s, q1.w);
                            s = __fadd_rn(s, q2.x); s = __fadd_rn(s, q2.y); s = __fadd_rn(s, q2.z); s = __fadd_rn(s, q2.w);
                            s = __fadd_rn(s, q3.x); s = __fadd_rn(s, q3.y); s = __fadd_rn(s, q3.z); s = __fadd_rn(s, q3.w);
                        }
                        for (; i < nvalid; ++i) s = __fadd_rn(s, xp[i]);
                    }
                    s = __shfl(s, 0, 64);
                    __builtin_amdgcn_wave_barrier();
                }
                from = q + 1;
            }
        return s;
    };

    if (walker) {
        // ---- the chains of level blockIdx.x: wave ch takes chain ch through the level's tickets, as they are parked
        const int ch = w;
        __builtin_amdgcn_s_setprio(2);
        float s = __uint_as_float((uint32_t)__hip_atomic_load(&lev[blockIdx.x].word[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // the carry
        unsigned long long tw = 0, tl = 0, tk = 0;
        for (int ws = 0; ws < nsteps; ++ws) {
            set_ticket((int)blockIdx.x, ws);
            const long long c0w = (dbg & 8) ? wall_clock64() : 0;
            // the ticket's sums: parked by the workgroup that streamed it
            int there = 0;
            if (lane == 0) {
                long long t_start = 0;
                int spins = 0;
                // (relaxed, like every load of the sums below: an agent-scope ACQUIRE is an invalidation of this XCD's L2 -- in a polling
                // loop of 225 waves it took the e1t*e2t plane away from the streaming workgroups again and again, 2.1 -> 3.1 ms at
                // eORCA12.  The loads below are issued after the flag has come back, and are agent-scope loads themselves.)
                while (!(there = __hip_atomic_load(&parked[(int64_t)k * nsteps + step], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __builtin_amdgcn_s_sleep(8);
                    if ((++spins & 1023) == 0 || scan_wait_ticks == 0) {
                        const long long now = wall_clock64();
                        if (!t_start) t_start = now;
                        // (a walker waits for the streaming itself, the whole kernel long if it must: ten times the limit of a hand-over)
                        // (CDFT_HEATC_SCAN_WAIT_US=0, the tests' hook, still makes the first wait that is not over at once give up)
                        if (now - t_start >= (scan_wait_ticks ? 10 * scan_wait_ticks + 100000000LL : 0LL) ||
                            __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            __hip_atomic_store(&ctl->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
            }
            there = __shfl(there, 0, 64);
            if (!there) return;                                // (the error flag is up: the host takes the plain chain)
            const long long c1w = (dbg & 8) ? wall_clock64() : 0;
            // every load of the ticket's sums is issued before the first is used (the compiler waits for an atomic load where its
            // value is needed: a load and an LDS write in turns were 32 memory round trips a ticket, 25 us of a walker's step)
            const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(park + ((int64_t)k * ring + step % ring) * park_ints) +
                                             (int64_t)ch * nchunk * 4;
            unsigned long long pw[kScanMaxChunks / 64][4];
            const bool second = (there >> (1 + ch)) & 1;         // (uniform) the ticket has second hypotheses for this chain
#pragma unroll
            for (int i = 0; i < kScanMaxChunks / 64; ++i) {
                const unsigned long long *o = slot + min(lane + 64 * i, nchunk - 1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pw[i][j] = (j < 2 || second) ? __hip_atomic_load(o + j * nchunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                 : (j == 3 ? 0xffffffff00000000ull : 0ull);      // (flag 0, key -1: no such hypothesis)
            }
#pragma unroll
            for (int i = 0; i < kScanMaxChunks / 64; ++i) {
                const int q = lane + 64 * i;
                if (q < nchunk) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        sm_K[h][q][ch]    = (int)(uint32_t)pw[i][2 * h];
                        sm_pos[h][q][ch]  = (int)(uint32_t)(pw[i][2 * h] >> 32);
                        sm_flag[h][q][ch] = (int)(uint32_t)pw[i][2 * h + 1];
                        sm_key[h][q][ch]  = (int)(uint32_t)(pw[i][2 * h + 1] >> 32);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const long long c2w = (dbg & 8) ? wall_clock64() : 0;
            s = walk_chain(ch, s);
            if (lane == 0) {
                __hip_atomic_store(&lev[k].word[ch], scan_word(step + 1, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the streaming workgroups: hypotheses, ring)
                if (step == nsteps - 1) sums4[k * 3 + ch] = s;
            }
            if (dbg & 8) { const long long c3w = wall_clock64(); tw += c1w - c0w; tl += c2w - c1w; tk += c3w - c2w; }
        }
        if ((dbg & 8) && lane == 0) {
            atomicAdd(&ctl->wk_wait, tw); atomicAdd(&ctl->wk_load, tl); atomicAdd(&ctl->wk_walk, tk);
            atomicMax(&ctl->t_last, (unsigned long long)wall_clock64());
        }
        return;
    }

    __shared__ int own_empty;                                     // (thread 0's) this XCD's lane has come back empty once
    if (threadIdx.x == 0) own_empty = 0;
    int pending = -1, pending_val = 0;                            // the ticket parked last, not yet said to be there (index into `parked`), and its flag
    __shared__ int sm_has1;
    for (;;) {
        // Dealing.  Lane x of `nlanes` holds the steps x, x + nlanes, ... of every level, in that order; a workgroup draws from
        // the lane of the XCD it runs on, and from the next lanes once its own is empty.  The 75 levels of a step multiply
        // the SAME 131072 values of e1t*e2t: drawn by one XCD they find them in its L2 after the first (dealt from one
        // counter the levels of a step spread over the eight L2s and the plane was fetched eight times, 0.4 B per cell).
        // No deadlock: every lane deals in increasing order, so the smallest ticket not yet finished is either held by a
        // workgroup that runs (all it waits for is finished) or the next one of its lane, whose workgroups all hold
        // finished-or-running smaller tickets and come back for it.
        if (threadIdx.x == 0) {
            int t = -1;
            for (int o = 0; o < nlanes && t < 0; ++o) {
                const int x = (my_lane + o) % nlanes;
                const int steps_x = (nsteps - x + nlanes - 1) / nlanes;            // steps x, x + nlanes, ... < nsteps
                // (the workgroup's own lane is drawn from without a look first -- one round trip instead of two -- until it has
                // once come back empty)
                if (steps_x <= 0 || ((o > 0 || own_empty) &&
                                     __hip_atomic_load(&ctl->lane_next[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= steps_x * a.npk)) continue;
                const int i = __hip_atomic_fetch_add(&ctl->lane_next[x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i < steps_x * a.npk) {
                    t = (x + nlanes * (i / a.npk)) * a.npk + i % a.npk;
                    if (o) atomicAdd(&ctl->nstolen, 1);
                } else if (o == 0) {
                    own_empty = 1;
                }
            }
            if (t >= 0) atomicAdd(&ctl->ticket, 1);
            tick = t;
        }
        // (walkers) the stores that parked the ticket before have been acknowledged -- vmcnt(0) in every wave: agent-scope stores,
        // written through this XCD's L2; the barrier itself only waits for LDS -- before the barrier, that ticket is said to be
        // there after it: the stores' way to memory goes by under the drawing of this ticket.  (A RELEASE on the flag would wait
        // for the storing wave's own stores only, and write back the L2 on top: 2.1 -> 2.4 ms at eORCA12.)
        if (pending >= 0) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
        __syncthreads();
        if (pending >= 0) {
            if (threadIdx.x == 0) {
                __hip_atomic_store(&parked[pending], pending_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 1 | chains with second words << 1
                if (dbg & 8) atomicAdd(&ctl->st_all, 1ull);       // (tickets parked)
            }
            pending = -1;
        }
        const int ticket = tick;
        if (ticket < 0) break;                                    // (uniform)
        if (threadIdx.x == 0) sm_has1 = 0;                        // (every wave is past its last reading of it: the barrier above; next written after two more)
        set_ticket(ticket % a.npk, ticket / a.npk);
        if ((dbg & 8) && threadIdx.x == 0) atomicMin(&ctl->t_first, (unsigned long long)wall_clock64());
        if (w < 3) {
            // The hypotheses of chunk `lane`, chain w: the binades of the chain's estimated value at both ends of the
            // level-kernel block the chunk lies in.  Estimated = the last EXACT value the level has published (a few
            // tickets back) + the r8 sum of the blocks since: the r4 chain drifts away from the r8 sums (by 1e-4 at
            // eORCA025, by per cents at eORCA12 where x is of the order of ulp(s)), but not over a handful of tickets.
            const unsigned long long pub = __hip_atomic_load(&lev[k].word[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int sp = min(step, (int)(pub >> 32));
            const float sv = __uint_as_float((uint32_t)pub);
            const double *el = est + (int64_t)k * (a.nblk + 1) * 3 + w;
            const int64_t ab = min((int64_t)a.nblk, max((int64_t)0, (int64_t)sp * ticket_cells - pad) / block_cells);
            for (int q = lane; q < nchunk; q += 64) {
                const int64_t c0 = t0 + (int64_t)q * kScanChunk;
                int k0 = -1, k1 = -1;
                if (c0 < nijv) {
                    const int64_t c0r = max((int64_t)0, c0 - pad), cb = c0r / block_cells;
                    const double b0 = el[cb * 3] - el[ab * 3], b1 = el[(cb + 1) * 3] - el[ab * 3];      // growth since the published value
                    const double per = (b1 - b0) / (double)block_cells;
                    const double g0 = b0 + per * (double)(c0r - cb * block_cells), g1 = g0 + per * (double)kScanChunk;
                    const double d0 = (double)sv + g0, d1 = (double)sv + g1;
                    k0 = scan_key((float)d0); k1 = scan_key((float)d1);
                    if (k0 < 0) { k0 = k1; k1 = -1; }
                    if (k1 == k0) k1 = -1;
                    if (k1 < 0 && k0 >= 0) {                      // the estimate may be off by a share of what it added to the exact value
                        const double u = 0.25 * fabs(g0);
                        const int ku = scan_key((float)(d0 + u)), kd = scan_key((float)(d0 - u));
                        k1 = (ku >= 0 && ku != k0) ? ku : (kd >= 0 && kd != k0) ? kd : -1;
                    }
                }
                sm_key[0][q][w] = k0; sm_key[1][q][w] = k1;
            }
        }
        __syncthreads();
        {   // Every chunk of the ticket; the loads run TWO sub-blocks ahead (two register sets, the loop unrolled by two so that
            // neither is ever copied): counters showed the waves waiting for memory 61 % of their time with one sub-block ahead
            // -- a chunk's vector work (~2 us with three waves on the SIMD) is about the latency of its loads under load.
            // Round 4: THREE sub-blocks ahead (AHEAD register sets, the loop unrolled by as many): the skeleton alone -- loads and
            // products, no sums -- ran at the pace of a wave's load round trips, not of HBM.
            constexpr int AHEAD = (MASK8 && !MXL) ? kScanAhead : 2;       // (the r4-mask and mixed-layer forms would spill with three sets)
            Raw rawA[kScanNV], rawB[kScanNV], rawC[kScanNV];
            f2v x[3][kScanNV][2];
            auto chunk = [&](Raw (&raw)[kScanNV], int q) {        // sums chunk q from `raw`, then asks for chunk q + AHEAD * kScanWaves into it
                if (q >= nchunk) return;                          // (AHEAD = 3 does not divide the chunks of a wave)
                const int64_t c0 = t0 + (int64_t)q * kScanChunk;
                if (c0 >= nijv) {                                 // past the end of the level: nothing there
                    if (lane == 63) for (int ch = 0; ch < 3; ++ch) sm_flag[0][q][ch] = sm_flag[1][q][ch] = 0;
                    return;
                }
                products(raw, c0, x);
                if (q + AHEAD * kScanWaves < nchunk && c0 + (int64_t)AHEAD * kScanSub < nijv) fetch(raw, c0 + (int64_t)AHEAD * kScanSub);
                if (!(dbg & 2)) summarize(x, q);      // (dbg: timing experiments of tools/, results then meaningless)
                else if (lane == 63) for (int ch = 0; ch < 3; ++ch) sm_flag[0][q][ch] = sm_flag[1][q][ch] = 0;     // (chunks of zeros: the walk passes them)
            };
            auto first = [&](Raw (&raw)[kScanNV], int q) {
                if (q < nchunk && t0 + (int64_t)q * kScanChunk < nijv) fetch(raw, t0 + (int64_t)q * kScanChunk);
            };
            first(rawA, w);
            first(rawB, w + kScanWaves);
            if (AHEAD == 3) first(rawC, w + 2 * kScanWaves);
#pragma clang loop unroll(disable)
            for (int q = w; q < nchunk; q += AHEAD * kScanWaves) {
                chunk(rawA, q);
                chunk(rawB, q + kScanWaves);
                if (AHEAD == 3) chunk(rawC, q + 2 * kScanWaves);
            }
        }
        __syncthreads();                                          // the sums of every chunk are in LDS
        if (ring > 0) {
            // ---- park the ticket's sums for the level's walker and go on streaming
            const long long ts1 = ((dbg & 8) && threadIdx.x == 0) ? wall_clock64() : 0;
            if (threadIdx.x == 0 && step >= ring) {
                // the slot still holds ticket step - ring until the three chains have passed it (their published tags say so)
                long long t_start = 0;
                int spins = 0;
                for (;;) {
                    int done = 0x7fffffff;
                    for (int c = 0; c < 3; ++c)
                        done = min(done, (int)(__hip_atomic_load(&lev[k].word[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32));
                    if (done > step - ring) break;
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 1023) == 0 || scan_wait_ticks == 0) {
                        const long long now = wall_clock64();
                        if (!t_start) t_start = now;
                        if (now - t_start >= scan_wait_ticks || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            __hip_atomic_store(&ctl->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
            }
            if (step >= ring) __syncthreads();
            const long long ts2 = ((dbg & 8) && threadIdx.x == 0) ? wall_clock64() : 0;
            // [chain][word 0..3][chunk], 64-bit words: (K, pos) and (flag, key) of the first hypothesis, then of the second -- the chunk
            // innermost, so that a store instruction of a wave writes 512 contiguous bytes (with the four words of a chunk side by
            // side each instruction wrote 8 bytes of every 32: counters had 0.75 B per cell written for 0.19 parked)
            unsigned long long *slot = reinterpret_cast<unsigned long long *>(park + ((int64_t)k * ring + step % ring) * park_ints);
            // (agent-scope stores and loads: the slot's lines are written and read by workgroups on different XCDs, ring after ring)
            auto two = [](int lo, int hi) { return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)hi << 32); };
            // which chains have a chunk with a second hypothesis at all (few tickets do: the ones near a power of two): the others'
            // second words are neither stored nor, by the walker, loaded -- half the parked bytes
            {
                int mine = 0;
                for (int idx = threadIdx.x; idx < 3 * nchunk; idx += 64 * kScanWaves) {
                    const int c = idx / nchunk, q = idx - c * nchunk;
                    if (sm_key[1][q][c] >= 0) mine |= 1 << c;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (__builtin_amdgcn_ballot_w64((mine >> c) & 1) != 0 && lane == 0) atomicOr(&sm_has1, 1 << c);
            }
            __syncthreads();
            const int has1 = sm_has1;
            for (int idx = threadIdx.x; idx < 3 * nchunk; idx += 64 * kScanWaves) {
                const int c = idx / nchunk, q = idx - c * nchunk;
                unsigned long long *o = slot + (int64_t)c * 4 * nchunk + q;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && !((has1 >> c) & 1)) continue;
                    __hip_atomic_store(o + (2 * h) * nchunk, two(sm_K[h][q][c], sm_pos[h][q][c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(o + (2 * h + 1) * nchunk, two(sm_flag[h][q][c], sm_key[h][q][c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            pending_val = 1 | (has1 << 1);
            pending = k * nsteps + step;                          // (said to be there at the next drawing, above)
            if ((dbg & 8) && threadIdx.x == 0) { atomicAdd(&ctl->st_ring, (unsigned long long)(ts2 - ts1)); atomicAdd(&ctl->st_park, (unsigned long long)(wall_clock64() - ts2)); }
        } else
        // ---- wave ch takes chain ch through the ticket on its own: no workgroup barrier, no other chain waited for
        if (w < 3) {
            const int ch = w;
            // the exact value where this ticket starts: published by the step before (tag == step)
            unsigned long long pub = 0;
            if (lane == 0) {
                // (the step waited for was dealt before this one: it is held by a workgroup that runs.  Two seconds of the
                // 100 MHz wall clock without it -- a preempted queue, a fault -- raise the error flag; every other wait sees the
                // flag and gives up too, the kernel ends and the host takes the plain chain instead)
                long long t_start = 0;
                int spins = 0;
                while (!(dbg & 1) && (int)((pub = __hip_atomic_load(&lev[k].word[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != step) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023) == 0 || scan_wait_ticks == 0) {
                        const long long now = wall_clock64();
                        if (!t_start) t_start = now;
                        if (now - t_start >= scan_wait_ticks || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            __hip_atomic_store(&ctl->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
            }
            const float s = walk_chain(ch, __uint_as_float((uint32_t)__shfl((int)(uint32_t)pub, 0, 64)));
            // publish: the value where the next step of this level starts
            if (lane == 0) {
                __hip_atomic_store(&lev[k].word[ch], scan_word(step + 1, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (step == nsteps - 1) sums4[k * 3 + ch] = s;
            }
        }
        if (ring == 0) __syncthreads();                           // (the LDS arrays are written again by the next ticket; with walkers the barrier of the next drawing stands between)
    }
}

// the plain chain of v_add_f32 (CDFT_HEATC_SEQ=chain: what heatc_scan_kernel is checked against)
int launch_seq_chain(const HeatcArgs &a, const float *carry, float *sums4, hipStream_t st) {
    const bool m8 = a.mask8 != nullptr, full = a.e31d != nullptr;
    dim3 grid(a.npk), block(64);
    if (m8 && full)        hipLaunchKernelGGL((heatc_seq_kernel<true, true>), grid, block, 0, st, a, carry, sums4);
    else if (m8 && !full)  hipLaunchKernelGGL((heatc_seq_kernel<true, false>), grid, block, 0, st, a, carry, sums4);
    else if (!m8 && full)  hipLaunchKernelGGL((heatc_seq_kernel<false, true>), grid, block, 0, st, a, carry, sums4);
    else                   hipLaunchKernelGGL((heatc_seq_kernel<false, false>), grid, block, 0, st, a, carry, sums4);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

// How many XCDs the workgroups of a launch land on (HW_REG_XCC_ID): 8 on an MI355X in SPX mode.  Asked once per device; the
// scan deals its tickets XCD by XCD only when the ids seen are exactly 0 .. n-1 (else from one counter, as before).
__global__ void heatc_xcc_probe_kernel(unsigned *seen) {
    if (threadIdx.x == 0) atomicOr(seen, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u));
}
int scan_lanes(hipStream_t st) {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    if (const char *e = getenv("CDFT_HEATC_SCAN_LANES")) return std::min(std::max(atoi(e), 1), kScanMaxLanes);   // (A/B and test hook: 1 = one counter)
    if (cached[dev]) return cached[dev];
    int n = 1;
    {
        unsigned *d = nullptr, h = 0;
        if (hipMalloc((void **)&d, sizeof(unsigned)) == hipSuccess) {
            if (hipMemsetAsync(d, 0, sizeof(unsigned), st) == hipSuccess) {
                hipLaunchKernelGGL(heatc_xcc_probe_kernel, dim3(4096), dim3(64), 0, st, d);
                if (hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
                    for (int c = 2; c <= kScanMaxLanes; ++c)
                        if (h == (1u << c) - 1u) n = c;
            }
            (void)hipFree(d);
        }
        (void)hipGetLastError();
    }
    return cached[dev] = n;
}

// the r4 sums (and the scan's control block) to where the host reads them: page-locked, device-mapped memory -- one small
// kernel behind the scan instead of two copies into pageable memory (4 + 18 + 4 + 32 us of a 0.55 ms call at eORCA025)
__global__ void heatc_seq_out_kernel(const float *__restrict__ s4, int n, const ScanCtl *__restrict__ ctl, float *__restrict__ h4,
                                     ScanCtl *__restrict__ hctl) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) h4[i] = s4[i];
    if (threadIdx.x == 0) {
        ScanCtl c = {};
        if (ctl) { c.ticket = ctl->ticket; c.error = ctl->error; c.nfallback = ctl->nfallback; c.nunguessed = ctl->nunguessed;
                   c.nslow = ctl->nslow; c.nstolen = ctl->nstolen;
                   c.wk_wait = ctl->wk_wait; c.wk_load = ctl->wk_load; c.wk_walk = ctl->wk_walk; c.st_ring = ctl->st_ring; c.st_park = ctl->st_park;
                   c.st_all = ctl->st_all; c.t_first = ctl->t_first; c.t_last = ctl->t_last; }
        *hctl = c;
    }
}

inline bool dbg_stats_on() { const char *e = getenv("CDFT_HEATC_SCAN_DEBUG"); return e && (atoi(e) & 8); }
// the sampled estimates (heatc_sample_kernel: blocks of kSampleBlock cells -- a multiple of a chunk, a divisor of a ticket), then the scan
int launch_seq_scan(HeatcArgs a, const float *carry, float *sums4, DevBuf &work, hipStream_t st) {
    const bool m8 = a.mask8 != nullptr, full = a.e31d != nullptr, mx = a.mxlopt != 0;
    const int block_cells = kSampleBlock;
    a.nblk = (int)((a.nij + kSampleBlock - 1) / kSampleBlock);
    // two seconds of the 100 MHz clock (CDFT_HEATC_SCAN_WAIT_US: test hook -- 0 makes the first wait that is not over at once give up)
    long long wait_ticks = 200000000LL;
    if (const char *e = getenv("CDFT_HEATC_SCAN_WAIT_US")) wait_ticks = std::max(0LL, atoll(e)) * 100LL;
    // about 32 tickets per level: enough of them in flight for every CU, few enough steps in a level's chain of waits
    int nchunk = 64;
    while (nchunk < kScanMaxChunks && a.nij > (int64_t)nchunk * kScanChunk * 32) nchunk *= 2;
    if (const char *e = getenv("CDFT_HEATC_SCAN_CHUNKS")) { const int v = atoi(e); if (v >= 8 && v <= kScanMaxChunks && v % 8 == 0) nchunk = v; }   // (tuning hook: a multiple of 8)
    const int64_t ticket_cells = (int64_t)nchunk * kScanChunk;
    const int nsteps = (int)((a.nij + 127 + ticket_cells - 1) / ticket_cells);   // (a level is walked with up to 127 cells of padding in front)
    const int nlanes = scan_lanes(st);
    const int align = getenv("CDFT_HEATC_SCAN_NOALIGN") ? 0 : 1;          // (A/B hook)
    // timing experiments only (the sums are then wrong): 1 = no chain waits, 2 = no chunk sums, 4 = no cell-by-cell resolution
    const int dbg = getenv("CDFT_HEATC_SCAN_DEBUG") ? atoi(getenv("CDFT_HEATC_SCAN_DEBUG")) : 0;
    if (dbg) {                                                    // never silently: the r4 sums of this call are not the reference's
        static bool told = false;
        if (!told) fprintf(stderr, "libcdftools_hip: CDFT_HEATC_SCAN_DEBUG=%d is a timing experiment -- parts of the reference-order scan are "
                                   "switched off and the sums cdft_heatc_compute returns are WRONG\n", dbg);
        told = true;
    }
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int per_cu = CDFT_SCAN_WGS;                           // (the kernel's launch bounds)
    // Walkers (round 6, see the kernel): on when the levels' walker workgroups leave most of the chip to the streaming ones;
    // a ring of up to 48 parked tickets per level (59 MB at eORCA12).  CDFT_HEATC_SCAN_WALKERS=0: the round-5 kernel, every
    // workgroup walking its own ticket (A/B and test hook); CDFT_HEATC_SCAN_RING=n: the ring (tests: 1 and 2)
    int ring = 4 * a.npk <= per_cu * ncu ? std::min(nsteps, 48) : 0;
    if (const char *e = getenv("CDFT_HEATC_SCAN_WALKERS")) { if (e[0] == '0') ring = 0; }
    if (ring > 0) { if (const char *e = getenv("CDFT_HEATC_SCAN_RING")) ring = std::max(1, std::min(nsteps, atoi(e))); }
    const size_t head = (sizeof(ScanLevel) * (size_t)a.npk + sizeof(ScanCtl) + 15) / 16 * 16;
    const size_t est_bytes = sizeof(double) * (size_t)a.npk * (a.nblk + 1) * 3;
    const size_t parked_bytes = ring > 0 ? (sizeof(int) * (size_t)a.npk * nsteps + 15) / 16 * 16 : 0;
    const size_t park_bytes = ring > 0 ? sizeof(int) * (size_t)a.npk * ring * 24 * nchunk : 0;
    CDFT_TRY(work.reserve(head + est_bytes + parked_bytes + park_bytes));
    ScanLevel *lev = (ScanLevel *)work.ptr;
    ScanCtl *ctl = (ScanCtl *)(lev + a.npk);
    double *est = (double *)((char *)work.ptr + head);
    int *parked = ring > 0 ? (int *)((char *)work.ptr + head + est_bytes) : nullptr;
    int *park = ring > 0 ? (int *)((char *)work.ptr + head + est_bytes + parked_bytes) : nullptr;
    {
        int64_t dense_cells = 32768;                              // read in full: where a chain grows from nothing
        if (const char *e = getenv("CDFT_HEATC_SCAN_DENSE")) dense_cells = std::max(0LL, atoll(e));           // (tuning hook)
        // the blocks read in full by one wave each, the others sampled 16 to a wave (tiny planes: one wave each throughout)
        const int nb_dense = a.nij >= 64 ? (int)std::min<int64_t>(a.nblk, (dense_cells + kSampleBlock - 1) / kSampleBlock) : a.nblk;
        const int nb_sparse = a.nblk - nb_dense;
        const bool tiny = a.nij < 64;                              // (one wave per block, scalar loads)
        const dim3 sgrid((unsigned)((a.nblk + 3) / 4), (unsigned)a.npk), sgrid16((unsigned)(nb_dense + (nb_sparse + 63) / 64), (unsigned)a.npk), sblock(256);
#define CDFT_SAMPLE_LAUNCH(M8, FU, MX) do { \
            if (tiny) hipLaunchKernelGGL((heatc_sample_kernel<M8, FU, MX>), sgrid, sblock, 0, st, a, dense_cells, a.nblk); \
            else      hipLaunchKernelGGL((heatc_sample16_kernel<M8, FU, MX>), sgrid16, sblock, 0, st, a, nb_dense); } while (0)
        if (m8 && full)        { if (mx) CDFT_SAMPLE_LAUNCH(true, true, true);   else CDFT_SAMPLE_LAUNCH(true, true, false); }
        else if (m8 && !full)  { if (mx) CDFT_SAMPLE_LAUNCH(true, false, true);  else CDFT_SAMPLE_LAUNCH(true, false, false); }
        else if (!m8 && full)  { if (mx) CDFT_SAMPLE_LAUNCH(false, true, true);  else CDFT_SAMPLE_LAUNCH(false, true, false); }
        else                   { if (mx) CDFT_SAMPLE_LAUNCH(false, false, true); else CDFT_SAMPLE_LAUNCH(false, false, false); }
#undef CDFT_SAMPLE_LAUNCH
        CDFT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(heatc_scan_est_kernel, dim3(a.npk), dim3(kEstThreads), 0, st, a.partial, a.nblk, carry, est, lev, ctl, parked, nsteps);
    const int64_t tickets = (int64_t)nsteps * a.npk;
    // (walkers: the first npk workgroups; with them the grid is what the chip holds at once, so that they all are resident)
    const int64_t streamers = std::max<int64_t>(1, std::min<int64_t>(tickets, per_cu * (int64_t)ncu - (ring > 0 ? a.npk : 0)));
    dim3 grid((unsigned)(streamers + (ring > 0 ? a.npk : 0))), block(64 * kScanWaves);
#define CDFT_SCAN_LAUNCH(M8, FU, MX) hipLaunchKernelGGL((heatc_scan_kernel<M8, FU, MX>), grid, block, 0, st, a, lev, ctl, est, block_cells, nchunk, nsteps, sums4, wait_ticks, nlanes, align, dbg, parked, park, ring)
    if (m8 && full)        { if (mx) CDFT_SCAN_LAUNCH(true, true, true);   else CDFT_SCAN_LAUNCH(true, true, false); }
    else if (m8 && !full)  { if (mx) CDFT_SCAN_LAUNCH(true, false, true);  else CDFT_SCAN_LAUNCH(true, false, false); }
    else if (!m8 && full)  { if (mx) CDFT_SCAN_LAUNCH(false, true, true);  else CDFT_SCAN_LAUNCH(false, true, false); }
    else                   { if (mx) CDFT_SCAN_LAUNCH(false, false, true); else CDFT_SCAN_LAUNCH(false, false, false); }
#undef CDFT_SCAN_LAUNCH
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

template <int VEC, int VPT>
int launch_vec_vpt(const HeatcArgs &a, hipStream_t st) {
    dim3 grid(xcd_band_grid(a.nblk, a.band)), block(kThreads);
    const size_t shmem = (size_t)(kThreads / 64) * a.npk * 3 * sizeof(double);
    const bool m8 = a.mask8 != nullptr, full = a.e31d != nullptr;
    if (m8 && full)        hipLaunchKernelGGL((heatc_level_kernel<VEC, true, true, VPT>), grid, block, shmem, st, a);
    else if (m8 && !full)  hipLaunchKernelGGL((heatc_level_kernel<VEC, true, false, VPT>), grid, block, shmem, st, a);
    else if (!m8 && full)  hipLaunchKernelGGL((heatc_level_kernel<VEC, false, true, VPT>), grid, block, shmem, st, a);
    else                   hipLaunchKernelGGL((heatc_level_kernel<VEC, false, false, VPT>), grid, block, shmem, st, a);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}
template <int VEC>
int launch_vec(const HeatcArgs &a, hipStream_t st) { return (a.vpt == 8 && VEC == 4) ? launch_vec_vpt<4, 8>(a, st) : launch_vec_vpt<VEC, kVPT>(a, st); }


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
    int rc = shard::make(&p->sh, npiglo, npjglo, npk, 0, ndev, devices, 0, 9);
    if (rc == CDFT_OK) rc = shard::heatc_parts(*p->sh);
    if (rc != CDFT_OK) { cdft_heatc_destroy(p); return rc; }
    *plan = p;
    return CDFT_OK;
}

int cdft_heatc_set_mesh(cdft_heatc_plan *p, const float *e1t, const float *e2t, const float *e3t,
                        const float *e31d, const void *tmask, int mask_is_r4, int memspace, void *stream) {
    TraceRange trace_("cdft_heatc_set_mesh");
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
    TraceRange trace_("cdft_heatc_compute");
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
    a.band = xcd_bands_on();
    const char *seq_env = getenv("CDFT_HEATC_SEQ");
    const bool r4seq = p->sum_mode == CDFT_HEATC_SUM_R4SEQ, chain = r4seq && seq_env && !strcmp(seq_env, "chain");
    int vec = 1;
    if (nij % 4 == 0 && aligned_to(dt, 16) && aligned_to(dm, 16)) vec = 4;
    else if (nij >= 4 && !getenv("CDFT_HEATC_NOU4")) vec = -4;           // 16-byte loads at r4 alignment (A/B hook: the old paths)
    else if (nij % 2 == 0 && aligned_to(dt, 8) && aligned_to(dm, 8)) vec = 2;
    // vectors per lane: 8 where that leaves four workgroups per CU (1.68 -> 1.65 ms at eORCA12; at eORCA025 there would be 212
    // workgroups for 256 CUs), else 4; CDFT_HEATC_VPT = 4 | 8: A/B hook
    auto blocks_with = [&](int v) { return vec == -4 ? (int)(((nij + 3) / 4 + 1 + kThreads * v - 1) / (kThreads * v))
                                                     : (int)((nij / vec + kThreads * v - 1) / (kThreads * v)); };
    a.vpt = ((vec == 4 || vec == -4) && blocks_with(8) >= 1024) ? 8 : kVPT;
    if (const char *e = getenv("CDFT_HEATC_VPT")) { const int v = atoi(e); if ((v == 8 && (vec == 4 || vec == -4)) || v == 4) a.vpt = v; }
    a.nblk = blocks_with(a.vpt);
    if (!r4seq) {                                         // the r8 sums per block
        if (vec == -4)     CDFT_TRY(launch_u4(a, st));
        else if (vec == 4) CDFT_TRY(launch_vec<4>(a, st));
        else if (vec == 2) CDFT_TRY(launch_vec<2>(a, st));
        else               CDFT_TRY(launch_vec<1>(a, st));
    }
    if (r4seq) {
        // the reference's sequential r4 sums, bit for bit (heatc_scan_kernel / heatc_seq_kernel); the r4 results are widened exactly
        const size_t n = (size_t)p->npk * 3;
        CDFT_TRY(p->st_s4.reserve(n * sizeof(float)));
        const float *dcarry = nullptr;
        if (!p->carry.empty()) {
            CDFT_TRY(p->st_c4.reserve(n * sizeof(float)));
            CDFT_HIP(hipMemcpyAsync(p->st_c4.ptr, p->carry.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
            dcarry = (const float *)p->st_c4.ptr;
        }
        if (chain) CDFT_TRY(launch_seq_chain(a, dcarry, (float *)p->st_s4.ptr, st));
        else       CDFT_TRY(launch_seq_scan(a, dcarry, (float *)p->st_s4.ptr, p->st_scan, st));
        p->last4.resize(n);
        const size_t h4_bytes = (n * sizeof(float) + 15) / 16 * 16;
        if (!p->h4) {                                                 // (once per plan)
            if (hipHostMalloc(&p->h4, h4_bytes + sizeof(ScanCtl), hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer(&p->h4_dev, p->h4, 0) != hipSuccess) {
                (void)hipGetLastError();
                if (p->h4) { (void)hipHostFree(p->h4); p->h4 = nullptr; }
                return fail(CDFT_ERR_ALLOC, "cdft_heatc_compute: page-locked allocation failed");
            }
        }
        auto fetch_out = [&]() -> int {
            const ScanCtl *dctl = chain ? nullptr : (const ScanCtl *)((char *)p->st_scan.ptr + sizeof(ScanLevel) * (size_t)p->npk);
            hipLaunchKernelGGL(heatc_seq_out_kernel, dim3(1), dim3(256), 0, st, (const float *)p->st_s4.ptr, (int)n, dctl, (float *)p->h4_dev,
                               (ScanCtl *)((char *)p->h4_dev + h4_bytes));
            CDFT_LAUNCH_CHECK();
            CDFT_HIP(hipStreamSynchronize(st));
            memcpy(p->last4.data(), p->h4, n * sizeof(float));
            return CDFT_OK;
        };
        CDFT_TRY(fetch_out());
        ScanCtl ctl = *(const ScanCtl *)((const char *)p->h4 + h4_bytes);
        if (!chain && getenv("CDFT_HEATC_SCAN_STATS"))
            fprintf(stderr, "heatc_scan_kernel: %d tickets, %d chunks resolved cell by cell (%d of them finished by the sequential loop), "
                    "%d times in a binade neither estimate named; %d tickets drawn from another XCD's lane\n", ctl.ticket, ctl.nfallback, ctl.nslow,
                    ctl.nunguessed, ctl.nstolen);
        if (!chain && getenv("CDFT_HEATC_SCAN_STATS") && (dbg_stats_on()))
            fprintf(stderr, "heatc_scan_kernel timing (us; CDFT_HEATC_SCAN_DEBUG=8): span %.1f; walker waves: waiting %.1f, loading %.1f, walking %.1f (sums over the waves); "
                    "streaming workgroups: ring waits %.1f, parking %.1f (sums over the workgroups), %.0f tickets parked\n", (double)(ctl.t_last - ctl.t_first) / 100.0,
                    ctl.wk_wait / 100.0, ctl.wk_load / 100.0, ctl.wk_walk / 100.0, ctl.st_ring / 100.0, ctl.st_park / 100.0, (double)ctl.st_all);
        if (ctl.error) {
            // a wait that did not end (heatc_scan_kernel): its results are discarded and the plain chain of v_add_f32 --
            // no workgroup waits for another there -- gives the same bits, 25 times slower
            fprintf(stderr, "cdfheatc: heatc_scan_kernel gave up waiting for a step of its chains; the sums are formed by the plain chain instead\n");
            CDFT_TRY(launch_seq_chain(a, dcarry, (float *)p->st_s4.ptr, st));
            CDFT_TRY(fetch_out());
        }
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
    if (p->h4) (void)hipHostFree(p->h4);
    p->st_a.release(); p->st_b.release(); p->st_temp.release(); p->st_mxl.release(); p->st_out.release();
    p->st_c4.release(); p->st_s4.release(); p->st_scan.release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"
