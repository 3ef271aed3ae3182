// transport.hip -- cdfmhst and cdfpsi (C ABI sections "cdfmhst", "cdfpsi").
//
// Both tools integrate a metric-weighted velocity (times tracer) vertically, cell by cell, and then
// reduce or scan the resulting 2-D r8 field:
//   cdfmhst src/cdfmhst.f90:308-367  dtrph += (zvt*e1v*e3v*1.d0)*pprau0*pprcp ; dtrps += zvs*e1v*e3v*1.d0
//                                    zonal SUM(dtrp*zmask) per row and basin
//   cdfpsi  src/cdfpsi.f90:271-298   dtrpu += zu*e2u*e3u*1.d0 ; dtrpv += zv*e1v*e3v*1.d0
//           :377-423                 running differences along j (U) or i (V), reference point, SNGL
// The vertical sums run sequentially in k per cell with the reference's r4 product order, so the 2-D
// transports are BIT-IDENTICAL to the reference; cdfpsi's scans are sequential too (bit-identical);
// only cdfmhst's zonal sums use a fixed tree order (tolerance 1e-13 * sum|terms|).
//
// HBM-bound: the 3-D pass reads 8 B (cdfpsi: u + e3u) or 12 B (cdfmhst: vt + vs + e3v) per cell
// once; the 2-D work is < 1 % of that.
#include <cstdlib>
#include "common.hpp"
#include "reduce.hpp"
#include "shard.hpp"

using namespace cdft;

namespace {

constexpr int kThreads = 256;

template <int VEC> struct FV;
template <> struct FV<4> { typedef float4 t; };
template <> struct FV<2> { typedef float2 t; };
template <> struct FV<1> { typedef float t; };
template <int VEC> __device__ __forceinline__ void ld(const float *p, int64_t e, float (&o)[VEC]);
// (16-byte loads at r4 alignment: a level of a plane that is not a multiple of 4 cells starts anywhere)
template <> __device__ __forceinline__ void ld<4>(const float *p, int64_t e, float (&o)[4]) {
    const f4u v = *reinterpret_cast<const f4u *>(p + e); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void ld<2>(const float *p, int64_t e, float (&o)[2]) {
    const float2 v = *reinterpret_cast<const float2 *>(p + e); o[0] = v.x; o[1] = v.y;
}
template <> __device__ __forceinline__ void ld<1>(const float *p, int64_t e, float (&o)[1]) { o[0] = p[e]; }
// read-once fields: nontemporal (common.hpp)
typedef float f2a __attribute__((ext_vector_type(2)));
template <int VEC> __device__ __forceinline__ void lds_(const float *p, int64_t e, float (&o)[VEC]);
template <> __device__ __forceinline__ void lds_<4>(const float *p, int64_t e, float (&o)[4]) {
    const f4u v = ld_stream4u(p + e); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void lds_<2>(const float *p, int64_t e, float (&o)[2]) {
    const f2a v = ld_stream(reinterpret_cast<const f2a *>(p + e)); o[0] = v.x; o[1] = v.y;
}
template <> __device__ __forceinline__ void lds_<1>(const float *p, int64_t e, float (&o)[1]) { o[0] = ld_stream(p + e); }

enum VMode { kPsi = 0, kMhstVt = 1, kMhstSep = 2, kVhst = 3 };

struct VintArgs {
    const float *f0, *f1, *f2;   // psi: velocity ; mhst -vt: vt, vs ; mhst sep: v, t, s
    const float *m2d;            // e2u / e1v (npi,npj)
    const float *e3;             // (npi,npj,npk) or null
    const float *e31d;           // (npk) or null
    double *acc0, *acc1;         // (npi,npj) r8 accumulators (acc1: salt)
    int64_t nij;
    int npi, npj, k0, k1, init, band;
};

// One lane owns VPT vectors of VEC consecutive cells and walks the levels k0..k1-1 in order
// (sequential r8 sum per cell).  The VPT vectors of a wave are 64 vectors apart, so every wave reads
// VPT KiB-sized contiguous runs per field and level (better DRAM row locality than one run per wave).
template <int VEC, int MODE, bool FULL, int VPT>
__global__ void __launch_bounds__(kThreads) vint_kernel(VintArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (xcd_band(a.band) * kThreads + threadIdx.x) >> 6;      // (common.hpp: XCD bands)
    int64_t p[VPT];
    bool ok[VPT], lastrow[VPT], own[VPT][VEC];
    float m[VPT][VEC];
    double s0[VPT][VEC], s1[VPT][VEC];
    // VEC = 4: vectors at cells 0, 4, 8, ... and, when the plane is not a multiple of 4 cells, one more anchored at
    // nij-4 whose leading cells belong to the previous vector (computed twice, stored once): no partial vector
    const int64_t nfull = a.nij / VEC, nvtot = nfull + ((VEC == 4 && (a.nij & 3)) ? 1 : 0);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int64_t q = (wave * VPT + v) * 64 + lane;
        ok[v] = q < nvtot;
        p[v] = ok[v] ? (VEC == 4 ? min(q * 4, a.nij - 4) : q * VEC) : 0;      // (not ok: a harmless in-range address)
        ld<VEC>(a.m2d, p[v], m[v]);
        lastrow[v] = (MODE == kMhstSep) && (p[v] / a.npi == a.npj - 1);    // npi % VEC == 0 in this mode
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            own[v][c] = ok[v] && (q < nfull || p[v] + c >= nfull * VEC);
            s0[v][c] = a.init ? 0.0 : a.acc0[p[v] + c];
            s1[v][c] = (MODE != kPsi && !a.init) ? a.acc1[p[v] + c] : 0.0;
        }
    }
    if (!ok[0]) return;
    // One level's operands of this lane.  Level k+1 is requested before level k is accumulated, so a
    // wave always has a full level of loads in flight behind its arithmetic (software pipeline).
    struct Level {
        float x0[VPT][VEC], x1[VPT][VEC], e3[VPT][VEC];
        float tn[MODE == kMhstSep ? VPT : 1][VEC], qn[MODE == kMhstSep ? VPT : 1][VEC];   // T, S one row north
        float vv[MODE == kMhstSep ? VPT : 1][VEC];
    };
    auto fetch = [&](Level &L, int k) {
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int64_t e = (int64_t)k * a.nij + p[v];
            if (FULL) {
                const float z = a.e31d[k];
#pragma unroll
                for (int c = 0; c < VEC; ++c) L.e3[v][c] = z;
            } else {
                lds_<VEC>(a.e3, e, L.e3[v]);
            }
            if (MODE == kMhstSep) {                       // src/cdfmhst.f90:312-322: T and S are read again by row j-1
                lds_<VEC>(a.f0, e, L.vv[v]);
                ld<VEC>(a.f1, e, L.x0[v]);
                ld<VEC>(a.f2, e, L.x1[v]);
                const int64_t en = lastrow[v] ? e : e + a.npi;
                ld<VEC>(a.f1, en, L.tn[v]);
                ld<VEC>(a.f2, en, L.qn[v]);
            } else {
                lds_<VEC>(a.f0, e, L.x0[v]);
                if (MODE == kMhstVt || MODE == kVhst) lds_<VEC>(a.f1, e, L.x1[v]);
            }
        }
    };
    auto accumulate = [&](const Level &L) {
#pragma unroll
        for (int v = 0; v < VPT; ++v)
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float x0 = L.x0[v][c], x1 = L.x1[v][c];
                if (MODE == kMhstSep) {
                    x0 = lastrow[v] ? 0.f : __fmul_rn(__fmul_rn(0.5f, __fadd_rn(L.x0[v][c], L.tn[v][c])), L.vv[v][c]);
                    x1 = lastrow[v] ? 0.f : __fmul_rn(__fmul_rn(0.5f, __fadd_rn(L.x1[v][c], L.qn[v][c])), L.vv[v][c]);
                }
                if (MODE == kVhst) {
                    // src/cdfvhst.f90:167-178: e3v = e3v*e1v first, then zvt*e3v*pp_rau0*pp_rcp*1.d0, all r4
                    const float em = __fmul_rn(L.e3[v][c], m[v][c]);
                    s0[v][c] = __dadd_rn(s0[v][c], (double)__fmul_rn(__fmul_rn(__fmul_rn(x0, em), 1000.f), 4000.f));
                    s1[v][c] = __dadd_rn(s1[v][c], (double)__fmul_rn(x1, em));
                    continue;
                }
                // (field*metric)*e3 in r4, then widened: src/cdfmhst.f90:332-333, src/cdfpsi.f90:277,296
                const double d0 = (double)__fmul_rn(__fmul_rn(x0, m[v][c]), L.e3[v][c]);
                if (MODE == kPsi) {
                    s0[v][c] = __dadd_rn(s0[v][c], d0);
                } else {
                    s0[v][c] = __dadd_rn(s0[v][c], __dmul_rn(__dmul_rn(d0, 1000.0), 4000.0));   // :336 pprau0, pprcp
                    s1[v][c] = __dadd_rn(s1[v][c], (double)__fmul_rn(__fmul_rn(x1, m[v][c]), L.e3[v][c]));   // :337
                }
            }
    };
    Level cur, nxt;
    if (a.k0 < a.k1) fetch(cur, a.k0);
    for (int k = a.k0; k < a.k1; ++k) {
        if (k + 1 < a.k1) fetch(nxt, k + 1);
        accumulate(cur);
        cur = nxt;
    }
#pragma unroll
    for (int v = 0; v < VPT; ++v)
#pragma unroll
        for (int c = 0; c < VEC; ++c)
            if (own[v][c]) {
                a.acc0[p[v] + c] = s0[v][c];
                if (MODE != kPsi) a.acc1[p[v] + c] = s1[v][c];
            }
}


template <int MODE>
int launch_vint(const VintArgs &a_in, hipStream_t st) {
    VintArgs a = a_in;
    auto al = [](const void *p, size_t n) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % n) == 0; };
    auto all = [&](size_t n) {
        return al(a.f0, n) && al(a.f1, n) && al(a.f2, n) && al(a.m2d, n) && al(a.e3, n);
    };
    int vec = 1;
    const bool rowok4 = MODE != kMhstSep || a.npi % 4 == 0, rowok2 = MODE != kMhstSep || a.npi % 2 == 0;
    if (a.nij >= 4 && rowok4 && !getenv("CDFT_VINT_ALIGNED_ONLY")) vec = 4;      // r4-aligned 16-byte loads, any plane size
    else if (a.nij % 4 == 0 && rowok4 && all(16)) vec = 4;
    else if (a.nij % 2 == 0 && rowok2 && all(8)) vec = 2;
    const bool full = a.e31d != nullptr;
    // vectors per lane.  The more bytes a wave has in flight the better these streams run, down to two waves per SIMD:
    // cdfpsi's form (one r8 sum per cell, two fields) 1.92 / 1.83 / 1.71 / 1.63 ms at eORCA12 with 1 / 2 / 4 / 8 vectors
    // (36 / 70 / 138 / 256 VGPRs); cdfmhst -vt (two sums, three fields) 2.45 -> 2.34 ms with 6 instead of 4 (256 VGPRs);
    // the other forms keep 4.  CDFT_VINT_VPT = 1 | 2 | 4 | 6 | 8: tuning and test hook (6 and 8 only where they fit).
    int vpt_env = 0;                                     // (read at every launch: tools/ab_env.py switches it inside one process)
    { const char *e = getenv("CDFT_VINT_VPT"); vpt_env = e ? atoi(e) : 0; if (vpt_env != 1 && vpt_env != 2 && vpt_env != 4 && vpt_env != 6 && vpt_env != 8) vpt_env = 0; }
    const int64_t nv4 = (a.nij + vec - 1) / vec;
    auto blocks_with = [&](int v) { return (nv4 + (int64_t)v * kThreads - 1) / ((int64_t)v * kThreads); };
    int vpt = vpt_env ? vpt_env : 4;
    if (!vpt_env && vec == 4) {                          // the larger shapes only where they leave four workgroups per CU (eORCA025: 6 is 20 % slower for cdfmhst)
        if (MODE == kPsi && blocks_with(8) >= 1024) vpt = 8;
        if (MODE == kMhstVt && blocks_with(6) >= 1024) vpt = 6;
    }
    if (vpt == 8 && !(MODE == kPsi && vec == 4)) vpt = 4;
    if (vpt == 6 && !((MODE == kMhstVt || MODE == kPsi) && vec == 4)) vpt = 4;
    const int64_t nvecs = (a.nij + vec - 1) / vec, nthr = (nvecs + vpt - 1) / vpt;
    a.band = xcd_bands_on();
    const dim3 grid(xcd_band_grid((nthr + kThreads - 1) / kThreads, a.band)), block(kThreads);
#define CDFT_GO(V, F) do { if (vpt == 1) hipLaunchKernelGGL((vint_kernel<V, MODE, F, 1>), grid, block, 0, st, a); \
                           else if (vpt == 8 && MODE == kPsi && V == 4) hipLaunchKernelGGL((vint_kernel<4, kPsi, F, 8>), grid, block, 0, st, a); \
                           else if (vpt == 6 && MODE == kMhstVt && V == 4) hipLaunchKernelGGL((vint_kernel<4, kMhstVt, F, 6>), grid, block, 0, st, a); \
                           else if (vpt == 6 && MODE == kPsi && V == 4) hipLaunchKernelGGL((vint_kernel<4, kPsi, F, 6>), grid, block, 0, st, a); \
                           else if (vpt == 2) hipLaunchKernelGGL((vint_kernel<V, MODE, F, 2>), grid, block, 0, st, a); \
                           else hipLaunchKernelGGL((vint_kernel<V, MODE, F, 4>), grid, block, 0, st, a); } while (0)
    if (vec == 4) { if (full) CDFT_GO(4, true); else CDFT_GO(4, false); }
    else if (vec == 2) { if (full) CDFT_GO(2, true); else CDFT_GO(2, false); }
    else { if (full) CDFT_GO(1, true); else CDFT_GO(1, false); }
#undef CDFT_GO
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

// cdfmhst zonal sums (src/cdfmhst.f90:340-367): one workgroup per row, every basin in one pass.
// masks (npi,npj,nb) r4; basin 0 (global) runs over ji = 2..npi-1, the others over the whole row.
template <int NB>
__global__ void __launch_bounds__(kThreads)
mhst_zonal_kernel(const double *__restrict__ dtrph, const double *__restrict__ dtrps,
                  const float *__restrict__ masks, int npi, int npj, int nb, double *__restrict__ heat,
                  double *__restrict__ salt) {
    __shared__ double scratch[16];
    const int jj = blockIdx.x;
    const int64_t nij = (int64_t)npi * npj, row = (int64_t)jj * npi;
    double sh[NB], ss[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { sh[b] = 0.0; ss[b] = 0.0; }
    for (int i = threadIdx.x; i < npi; i += kThreads) {
        const double h = dtrph[row + i], s = dtrps[row + i];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < nb && !(b == 0 && (i == 0 || i == npi - 1))) {
                const double zm = (double)masks[(int64_t)b * nij + row + i];
                sh[b] += h * zm;
                ss[b] += s * zm;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b < nb) {                                   // nb is uniform: no divergent barrier
            const double th = block_sum(sh[b], scratch), ts = block_sum(ss[b], scratch);
            if (threadIdx.x == 0) {
                heat[(int64_t)jj * nb + b] = th;
                salt[(int64_t)jj * nb + b] = ts;
            }
        }
    }
}

// Basin masks that hold only 0 and 1 (the usual case) are packed to one byte per cell at set_mesh: the zonal kernel
// then reads 17 bytes per cell instead of 16 + 4*nb, and multiplies by the same 0.d0 / 1.d0.  flag[0] is raised by any
// other value: the r4 masks are used as they are.
__global__ void mhst_mask_bits_kernel(const float *__restrict__ masks, int64_t nij, int nb, uint8_t *__restrict__ bits,
                                      int *__restrict__ flag) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    unsigned v = 0;
    for (int b = 0; b < nb; ++b) {
        const float m = masks[(int64_t)b * nij + p];
        if (m == 1.f) v |= 1u << b;
        else if (!(m == 0.f) || signbit(m)) flag[0] = 1;         // -0. would flip the sign of a product: not packed either
    }
    bits[p] = (uint8_t)v;
}

template <int NB>
__global__ void __launch_bounds__(kThreads)
mhst_zonal_bits_kernel(const double *__restrict__ dtrph, const double *__restrict__ dtrps,
                       const uint8_t *__restrict__ bits, int npi, int npj, int nb, double *__restrict__ heat,
                       double *__restrict__ salt) {
    __shared__ double scratch[16];
    const int jj = blockIdx.x;
    const int64_t row = (int64_t)jj * npi;
    double sh[NB], ss[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { sh[b] = 0.0; ss[b] = 0.0; }
    for (int i = threadIdx.x; i < npi; i += kThreads) {
        const double h = dtrph[row + i], s = dtrps[row + i];
        const unsigned v = bits[row + i];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < nb && !(b == 0 && (i == 0 || i == npi - 1))) {
                const double zm = ((v >> b) & 1u) ? 1.0 : 0.0;
                sh[b] += h * zm;
                ss[b] += s * zm;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b < nb) {
            const double th = block_sum(sh[b], scratch), ts = block_sum(ss[b], scratch);
            if (threadIdx.x == 0) {
                heat[(int64_t)jj * nb + b] = th;
                salt[(int64_t)jj * nb + b] = ts;
            }
        }
    }
}

// cdfmhst -Zdim from a VT file: the running totals after every level (src/cdfmhst.f90:332-367 inside the level loop).  One
// workgroup (256 threads, 512 for rows beyond 1536 cells) = one row jj for ALL levels: a thread keeps the two running r8 sums of
// its cells (i = tid, tid + T, ...: at most 9 -- 17 per thread of 256, or 5 per thread of 1024 with their 128 registers, went to
// scratch memory at eORCA12: 32 ms) in
// registers, adds level k exactly as vint_kernel does, and the row's zonal sums of the totals so far -- every basin, heat
// and salt -- leave through wave_sum8 and one barrier per level.  (Until round 4: a vertical-integral launch and a zonal
// launch per level with the running sums going through memory, 61 bytes per cell and level: 2.3 ms at eORCA025.)
template <int NC, bool FULL, int T>
__global__ void __launch_bounds__(T)
mhst_zdim_kernel(const float *__restrict__ vt, const float *__restrict__ vs, const float *__restrict__ e1v, const float *__restrict__ e3v,
                 const float *__restrict__ e31d, const uint8_t *__restrict__ bits, int npi, int npj, int npk, int nb,
                 double *__restrict__ heat, double *__restrict__ salt) {
    __shared__ double scr[T / 64][8 * kSum8Stride];
    __shared__ double part[2][T / 64][16];
    const int jj = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nij = (int64_t)npi * npj, row = (int64_t)jj * npi;
    float m[NC];
    unsigned bb[NC];
    double s0[NC], s1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = tid + c * T;
        const bool in = i < npi;
        m[c] = in ? e1v[row + i] : 0.f;
        bb[c] = in ? bits[row + i] : 0u;
        if (in && (i == 0 || i == npi - 1)) bb[c] &= ~1u;       // basin 0 (global) runs over ji = 2 .. npi-1 (:340-367)
        s0[c] = 0.0; s1[c] = 0.0;
    }
    for (int k = 0; k < npk; ++k) {
        const int64_t base = (int64_t)k * nij + row;
        const float e3k = FULL ? e31d[k] : 0.f;
        float x0[NC], x1[NC], e3[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(bb[c]));   // (the 0.d0 / 1.d0 factors are formed from the bytes at every
                                                                    //  level: hoisted out of the loop they are 16 registers per cell)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = min(tid + c * T, npi - 1);       // (clamped: every load unconditional; the clamped cells have m = 0, bits = 0)
            x0[c] = ld_stream(vt + base + i); x1[c] = ld_stream(vs + base + i);
            e3[c] = FULL ? e3k : ld_stream(e3v + base + i);
        }
        double hv[8], sv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { hv[q] = 0.0; sv[q] = 0.0; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool in = tid + c * T < npi;
            // (field*metric)*e3 in r4, then widened: src/cdfmhst.f90:332-337 -- as vint_kernel<kMhstVt>
            const double d0 = (double)__fmul_rn(__fmul_rn(x0[c], m[c]), e3[c]);
            if (in) {
                s0[c] = __dadd_rn(s0[c], __dmul_rn(__dmul_rn(d0, 1000.0), 4000.0));
                s1[c] = __dadd_rn(s1[c], (double)__fmul_rn(__fmul_rn(x1[c], m[c]), e3[c]));
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {                         // as mhst_zonal_bits_kernel: h * (0.d0 | 1.d0), a thread's cells in order
                const double zm = ((bb[c] >> b) & 1u) ? 1.0 : 0.0;
                hv[b] += s0[c] * zm;
                sv[b] += s1[c] * zm;
            }
        }
        const double th = wave_sum8(hv, scr[wave], lane), ts = wave_sum8(sv, scr[wave], lane);
        if ((lane & 7) == 0) { part[k & 1][wave][lane >> 3] = th; part[k & 1][wave][8 + (lane >> 3)] = ts; }
        __syncthreads();                                          // (one per level: part is double-buffered)
        if (tid < 16) {
            double t = 0.0;
            for (int w = 0; w < T / 64; ++w) t += part[k & 1][w][tid];   // waves in order
            const int b = tid & 7;
            if (b < nb) (tid < 8 ? heat : salt)[((int64_t)k * npj + jj) * nb + b] = t;
        }
    }
}

// cdfpsi U path, src/cdfpsi.f90:395-398: dpsiu(:,1) = 0 ; dpsiu(:,jj) = dpsiu(:,jj-1) - dtrpu(:,jj)
// Row-block sharding: a block that does not hold global row 1 continues the running difference of
// the block south of it: it starts from that block's last row (carry) and its first row is an
// ordinary row.  Chained this way the result does not depend on the number of blocks (bit for bit).
__global__ void __launch_bounds__(64) psi_scan_j_kernel(const double *__restrict__ dtrp, double *__restrict__ dpsi, int npi, int npj,
                                  const double *__restrict__ carry, int first_block) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npi) return;
    double acc = carry ? carry[i] : 0.0;
    if (first_block) {
        dpsi[i] = acc;                                    // global row 1: dpsiu(:,1) = 0, its transport is not used
    } else {
        acc = __dadd_rn(acc, -dtrp[i]);
        dpsi[i] = acc;
    }
    // Only the additions are sequential: rows are fetched kB at a time, and the next batch is in
    // flight while the current one is accumulated and stored (double buffer in registers).  Only
    // npi/64 waves exist, so each must keep ~100 KB in flight to draw a useful share of the HBM
    // bandwidth: one wave per SIMD with (nearly) the whole 512-VGPR file, 2 x 96 rows of r8.
    constexpr int kB = 96;
    double xa[kB], xb[kB];
    auto fetch = [&](double (&x)[kB], int j0) {
#pragma unroll
        for (int r = 0; r < kB; ++r) x[r] = (j0 + r < npj) ? dtrp[(int64_t)(j0 + r) * npi + i] : 0.0;
    };
    auto drain = [&](const double (&x)[kB], int j0) {
#pragma unroll
        for (int r = 0; r < kB; ++r)
            if (j0 + r < npj) {
                acc = __dadd_rn(acc, -x[r]);
                dpsi[(int64_t)(j0 + r) * npi + i] = acc;
            }
    };
    fetch(xa, 1);
    for (int j0 = 1; j0 < npj; j0 += 2 * kB) {
        fetch(xb, j0 + kB);
        drain(xa, j0);
        fetch(xa, j0 + 2 * kB);
        drain(xb, j0 + kB);
    }
}

// The same scan with a workgroup per 32 columns (the kernel that runs): the column chain of additions is sequential,
// the loads and stores around it are not.  16 waves bring 128 rows x 32 columns into LDS (a load instruction covers 2
// rows of 256 bytes; the next 128 rows are already on their way into registers), the first 32 lanes of wave 0 run the
// chain through LDS in place, 8 rows read ahead of the additions, and all 16 waves store the result.  One wave per 64
// columns (above) has npiglo/64 waves on the whole chip: 0.25 ms at eORCA12, latency-bound; this one 0.085 ms
// (16 / 64 columns and 32 .. 512 rows per pass were measured too: 0.09 .. 0.15 ms).
constexpr int kScanWaves = 16, kScanRW = 4, kScanCW = 32, kScanRPI = 64 / kScanCW;
constexpr int kScanRows = kScanWaves * kScanRW * kScanRPI;

__global__ void __launch_bounds__(kScanWaves * 64)
psi_scan_j_wg_kernel(const double *__restrict__ dtrp, double *__restrict__ dpsi, int npi, int npj,
                     const double *__restrict__ carry, int first_block) {
    __shared__ double rows[kScanRows][kScanCW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % kScanCW, sub = lane / kScanCW;
    const int i = blockIdx.x * kScanCW + c;
    const bool col = i < npi;
    double acc = (carry && col) ? carry[i] : 0.0;                 // used by the chain lanes only
    double x[kScanRW];
    auto row_of = [&](int r) { return (wave * kScanRW + r) * kScanRPI + sub; };
    auto fetch = [&](int j0) {
#pragma unroll
        for (int r = 0; r < kScanRW; ++r) {
            const int j = j0 + row_of(r);
            x[r] = (col && j < npj) ? __builtin_nontemporal_load(dtrp + (int64_t)j * npi + i) : 0.0;
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < npj; j0 += kScanRows) {
#pragma unroll
        for (int r = 0; r < kScanRW; ++r) rows[row_of(r)][c] = x[r];
        if (j0 + kScanRows < npj) fetch(j0 + kScanRows);
        __syncthreads();
        if (threadIdx.x < kScanCW) {                              // rows past the end hold 0: the chain passes through them
            double t[8], tn[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = rows[r][c];
            // global row 1 of the first block: dpsiu(:,1) = 0 (or the carry), its transport is not used
            if (first_block && j0 == 0) t[0] = 0.0;
            for (int r0 = 0; r0 < kScanRows; r0 += 8) {
                if (r0 + 8 < kScanRows) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) tn[r] = rows[r0 + 8 + r][c];
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    acc = __dadd_rn(acc, -t[r]);
                    rows[r0 + r][c] = acc;
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) t[r] = tn[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kScanRW; ++r) {
            const int j = j0 + row_of(r);
            if (col && j < npj) dpsi[(int64_t)j * npi + i] = rows[row_of(r)][c];
        }
    }
}

// The scan that runs since round 4: the chain wave waits for nothing but its terms.  Wave 0 (32 lanes, one per column) only
// walks the chain, in place, through a ring of four LDS slots (112 rows x 32 columns each); waves 1..7 only load (pass q
// travels in register set q % 6, requested six passes before it is written to its slot; every load is unconditional, at a
// clamped address, so that the waits count loads instead of waiting for all); waves 8..15 only store.  No workgroup barrier
// after the first: three sets of LDS counters say which pass has been written, walked and stored, and a wave sleeps on the
// one it needs.  Measured (profiles/r04/psi_scan_*.txt): 25.2 us at eORCA025 (27.6 with the round-3 kernel), 78.6 at eORCA12
// (84).  Where it goes at eORCA025: 4.7 us launching 46 workgroups of 1024 threads and 114 KB, 9.7 us the chain itself
// (7.9 ns per row: tools/chain_probe.hip has 2.4 ns for the additions alone and 3.7 when a whole pass is held in registers,
// which 1024-thread workgroups have no room for), 3.7 us the counters, 7 us the first loads and the last stores.  What was
// tried on the way: one barrier per pass (every interval costs the slowest of the three roles: 26 us); loading and storing in
// the same waves (with both in flight every wait is for everything: one counter, no order between the two kinds); the chain
// wave storing its own results, a 256-byte store per row (a wave has 64 memory operations in flight at most: 35 ns per row).
constexpr int kRingCW = 32, kRingSub = 64 / kRingCW, kRingFeed = 7, kRingDrain = 8, kRingRW = 8, kRingDW = 7;   // 7 x 8 = 8 x 7 rows per half wave
constexpr int kRingRows = kRingFeed * kRingRW * kRingSub, kRingSlots = 4, kRingAhead = 6, kRingCnt = 8;
static_assert(kRingRows == kRingDrain * kRingDW * kRingSub && kRingRows % 8 == 0, "loaders and storers cover the same rows");

// (workgroup-scope acquire / release on LDS: the wave's own LDS operations are complete before the count moves, and the
// count is read before the data it guards)
__device__ __forceinline__ void ring_wait(int *ctr, int need) {
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void ring_bump(int *ctr) { (void)__hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

__global__ void __launch_bounds__((1 + kRingFeed + kRingDrain) * 64)
psi_scan_j_ring_kernel(const double *__restrict__ dtrp, double *__restrict__ dpsi, int npi, int npj,
                       const double *__restrict__ carry, int first_block) {
    extern __shared__ double ring[];                              // [kRingSlots][kRingRows][kRingCW], then the counters
    // filled[q % 8]: loader waves that have written pass q (and the passes q-8, q-16, ... before it); walked: passes the chain
    // has left; stored[q % 8]: storer waves that have stored pass q (...)
    int *filled = reinterpret_cast<int *>(ring + (size_t)kRingSlots * kRingRows * kRingCW);
    int *stored = filled + kRingCnt, *walked = stored + kRingCnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % kRingCW, sub = lane / kRingCW;
    const int i = blockIdx.x * kRingCW + c;
    const bool col = i < npi;
    const int ic = min(i, npi - 1);
    const int npass = (npj + kRingRows - 1) / kRingRows;
    if (threadIdx.x < 2 * kRingCnt + 1) filled[threadIdx.x] = 0;
    __syncthreads();                                              // (the only one)
    auto slot = [&](int p) { return ring + (size_t)(p % kRingSlots) * kRingRows * kRingCW; };
    if (wave == 0) {
        // ---- the chain ----
        __builtin_amdgcn_s_setprio(3);                            // (it shares its SIMD with three of the other waves)
        double acc = (carry && col && lane < kRingCW) ? carry[i] : 0.0;
        for (int p = 0; p < npass; ++p) {
            ring_wait(filled + p % kRingCnt, (p / kRingCnt + 1) * kRingFeed);       // pass p is in its slot
            if (lane < kRingCW) {
                double *rows = slot(p);
                // Batches of 14 rows (7 + 7 two-row LDS operations behind the reads a batch waits for: the counter holds 15): the
                // terms of a batch arrive in registers a batch ahead, its 14 additions run back to back
                // (2.4 ns each, tools/chain_probe.hip), then its 14 results go to LDS in one run.  With every pair of additions
                // followed by its own ds_write -- the compiler's schedule of the plain loop -- a row cost 8.5 ns.
                constexpr int BR = 14, NB = kRingRows / BR;
                static_assert(kRingRows % BR == 0, "whole batches");
                double t[2][BR];
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int r = 0; r < BR; ++r) t[g][r] = rows[(g * BR + r) * kRingCW + c];
                // global row 1 of the first block: dpsiu(:,1) = 0 (or the carry), its transport is not used
                if (first_block && p == 0) t[0][0] = 0.0;
#pragma unroll
                for (int g = 0; g < NB; ++g) {
#pragma unroll
                    for (int r = 0; r < BR; ++r) {
                        acc = __dadd_rn(acc, -t[g % 2][r]);         // rows past the end hold 0: the chain passes through them
                        t[g % 2][r] = acc;
                    }
                    __builtin_amdgcn_sched_barrier(0);              // (keep the additions together)
#pragma unroll
                    for (int r = 0; r < BR; ++r) rows[(g * BR + r) * kRingCW + c] = t[g % 2][r];
                    if (g + 2 < NB) {
#pragma unroll
                        for (int r = 0; r < BR; ++r) t[g % 2][r] = rows[((g + 2) * BR + r) * kRingCW + c];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (lane == 0) ring_bump(walked);                     // (release: the results are in LDS before the count says so)
        }
        return;
    }
    if (wave <= kRingFeed) {
        // ---- the loaders ----
        const int lw = wave - 1;
        auto row_of = [&](int r) { return (lw * kRingRW + r) * kRingSub + sub; };
        auto fetch = [&](double (&x)[kRingRW], int p) {           // (branch-free: a clamped address, the value dropped in put)
#pragma unroll
            for (int r = 0; r < kRingRW; ++r) {
                const int j = min(p * kRingRows + row_of(r), npj - 1);
                x[r] = __builtin_nontemporal_load(dtrp + (int64_t)j * npi + ic);
            }
        };
        auto put = [&](const double (&x)[kRingRW], int p) {
            if (p >= npass) return;
            if (p >= kRingSlots) ring_wait(stored + (p - kRingSlots) % kRingCnt, ((p - kRingSlots) / kRingCnt + 1) * kRingDrain);   // the slot is free
            double *rows = slot(p);
#pragma unroll
            for (int r = 0; r < kRingRW; ++r) {
                const int j = p * kRingRows + row_of(r);
                rows[row_of(r) * kRingCW + c] = (col && j < npj) ? x[r] : 0.0;
            }
            if (lane == 0) ring_bump(filled + p % kRingCnt);
        };
        double x[kRingAhead][kRingRW];
#pragma unroll
        for (int q = 0; q < kRingAhead; ++q) fetch(x[q], q);
        for (int p = 0; p < npass; p += kRingAhead) {
#pragma unroll
            for (int d = 0; d < kRingAhead; ++d) { put(x[d], p + d); fetch(x[d], p + d + kRingAhead); }
        }
        return;
    }
    // ---- the storers ----
    const int dw = wave - 1 - kRingFeed;
    for (int p = 0; p < npass; ++p) {
        ring_wait(walked, p + 1);                                 // the chain has left pass p
        const double *rows = slot(p);
        double y[kRingDW];
#pragma unroll
        for (int r = 0; r < kRingDW; ++r) y[r] = rows[((dw * kRingDW + r) * kRingSub + sub) * kRingCW + c];
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the slot has been read into registers ...
        if (lane == 0) ring_bump(stored + p % kRingCnt);          // ... and may be written again
#pragma unroll
        for (int r = 0; r < kRingDW; ++r) {
            const int j = p * kRingRows + (dw * kRingDW + r) * kRingSub + sub;
            if (col && j < npj) dpsi[(int64_t)j * npi + i] = y[r];
        }
    }
}

// The same ring for the scan along i of the V path (src/cdfpsi.f90:380-383: dpsiv(npi,:) = 0 ; dpsiv(ji,:) = dpsiv(ji+1,:) -
// dtrpv(ji,:), ji = npi-1 .. 1): a workgroup takes 32 ROWS, a pass is 112 columns from the east; a load or store instruction
// moves 64 columns of one row (512 contiguous bytes), so the LDS slot is [column][row] with 33 doubles between columns: the
// loaders write it along columns, the chain reads it along rows, both free of bank conflicts.  Eight waves load (two
// instructions per row and pass, eight per wave), seven store.  (One wave per 64 rows, the round-2 kernel below, has 19 waves
// on the whole chip at eORCA025: 0.5 ms.)
constexpr int kRingIStr = kRingCW + 1, kRingIFeed = 8, kRingIDrain = 7, kRingILoads = 2 * kRingCW / kRingIFeed;

__global__ void __launch_bounds__((1 + kRingIFeed + kRingIDrain) * 64)
psi_scan_i_ring_kernel(const double *__restrict__ dtrp, double *__restrict__ dpsi, int npi, int npj) {
    extern __shared__ double ring[];                              // [kRingSlots][kRingRows][kRingIStr], then the counters
    constexpr int kSlot = kRingRows * kRingIStr;
    int *filled = reinterpret_cast<int *>(ring + (size_t)kRingSlots * kSlot);
    int *stored = filled + kRingCnt, *walked = stored + kRingCnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = blockIdx.x * kRingCW;
    const int npass = (npi + kRingRows - 1) / kRingRows;
    if (threadIdx.x < 2 * kRingCnt + 1) filled[threadIdx.x] = 0;
    __syncthreads();                                              // (the only one)
    auto slot = [&](int p) { return ring + (size_t)(p % kRingSlots) * kSlot; };
    if (wave == 0) {
        // ---- the chain: lane c walks row j0 + c from the east ----
        __builtin_amdgcn_s_setprio(3);
        const int c = lane;
        double acc = 0.0;
        for (int p = 0; p < npass; ++p) {
            ring_wait(filled + p % kRingCnt, (p / kRingCnt + 1) * kRingIFeed);
            if (lane < kRingCW) {
                double *rows = slot(p);
                constexpr int BR = 14, NB = kRingRows / BR;       // (as in the scan along j)
                double t[2][BR];
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int r = 0; r < BR; ++r) t[g][r] = rows[(g * BR + r) * kRingIStr + c];
                if (p == 0) t[0][0] = 0.0;                        // the eastern column itself is 0: its transport is not used
#pragma unroll
                for (int g = 0; g < NB; ++g) {
#pragma unroll
                    for (int r = 0; r < BR; ++r) {
                        acc = __dadd_rn(acc, -t[g % 2][r]);         // columns past the western end hold 0
                        t[g % 2][r] = acc;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < BR; ++r) rows[(g * BR + r) * kRingIStr + c] = t[g % 2][r];
                    if (g + 2 < NB) {
#pragma unroll
                        for (int r = 0; r < BR; ++r) t[g % 2][r] = rows[((g + 2) * BR + r) * kRingIStr + c];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (lane == 0) ring_bump(walked);
        }
        return;
    }
    // item q = 0 .. 63 of a pass: row q / 2 of the workgroup, columns 64 (q % 2) + lane of the pass (the second half: 48)
    auto col_of = [&](int p, int q) { return npi - 1 - (p * kRingRows + (q & 1) * 64 + lane); };
    if (wave <= kRingIFeed) {
        // ---- the loaders ----
        const int lw = wave - 1;
        auto fetch = [&](double (&x)[kRingILoads], int p) {       // (branch-free: clamped addresses, the value dropped in put)
#pragma unroll
            for (int r = 0; r < kRingILoads; ++r) {
                const int q = lw * kRingILoads + r;
                const int j = min(j0 + (q >> 1), npj - 1), i = min(max(col_of(p, q), 0), npi - 1);
                x[r] = __builtin_nontemporal_load(dtrp + (int64_t)j * npi + i);
            }
        };
        auto put = [&](const double (&x)[kRingILoads], int p) {
            if (p >= npass) return;
            if (p >= kRingSlots) ring_wait(stored + (p - kRingSlots) % kRingCnt, ((p - kRingSlots) / kRingCnt + 1) * kRingIDrain);
            double *rows = slot(p);
#pragma unroll
            for (int r = 0; r < kRingILoads; ++r) {
                const int q = lw * kRingILoads + r, step = (q & 1) * 64 + lane;
                if (step < kRingRows) rows[step * kRingIStr + (q >> 1)] = (j0 + (q >> 1) < npj && col_of(p, q) >= 0) ? x[r] : 0.0;
            }
            if (lane == 0) ring_bump(filled + p % kRingCnt);
        };
        double x[kRingAhead][kRingILoads];
#pragma unroll
        for (int q = 0; q < kRingAhead; ++q) fetch(x[q], q);
        for (int p = 0; p < npass; p += kRingAhead) {
#pragma unroll
            for (int d = 0; d < kRingAhead; ++d) { put(x[d], p + d); fetch(x[d], p + d + kRingAhead); }
        }
        return;
    }
    // ---- the storers ----
    const int dw = wave - 1 - kRingIFeed;
    for (int p = 0; p < npass; ++p) {
        ring_wait(walked, p + 1);
        const double *rows = slot(p);
        constexpr int kMaxItems = (2 * kRingCW + kRingIDrain - 1) / kRingIDrain;
        double y[kMaxItems];
#pragma unroll
        for (int r = 0; r < kMaxItems; ++r) {
            const int q = dw + r * kRingIDrain, step = (q & 1) * 64 + lane;
            y[r] = (q < 2 * kRingCW && step < kRingRows) ? rows[step * kRingIStr + (q >> 1)] : 0.0;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the slot has been read into registers ...
        if (lane == 0) ring_bump(stored + p % kRingCnt);          // ... and may be written again
#pragma unroll
        for (int r = 0; r < kMaxItems; ++r) {
            const int q = dw + r * kRingIDrain, step = (q & 1) * 64 + lane, i = col_of(p, q), j = j0 + (q >> 1);
            if (q < 2 * kRingCW && step < kRingRows && i >= 0 && j < npj) dpsi[(int64_t)j * npi + i] = y[r];
        }
    }
}

// cdfpsi V path, src/cdfpsi.f90:380-383: dpsiv(npi,:) = 0 ; dpsiv(ji,:) = dpsiv(ji+1,:) - dtrpv(ji,:),
// ji = npi-1..1.  The scan runs along i, the contiguous axis: one wave takes 64 rows, stages a
// 64 x 64 tile through LDS with coalesced row segments, each lane scans its own row of the tile
// sequentially (carry in a register), and the tile is written back coalesced.
__global__ void __launch_bounds__(64) psi_scan_i_kernel(const double *__restrict__ dtrp, double *__restrict__ dpsi,
                                                        int npi, int npj) {
    __shared__ double tile[64][65];
    const int lane = threadIdx.x, j0 = blockIdx.x * 64;
    double carry = 0.0;                                   // dpsiv(ji+1, row) entering the chunk
    for (int c1 = npi; c1 > 0; c1 -= 64) {                // columns [c0, c1), right to left
        const int c0 = max(c1 - 64, 0), w = c1 - c0;
        for (int r = 0; r < 64; ++r)
            if (j0 + r < npj && lane < w) tile[r][lane] = dtrp[(int64_t)(j0 + r) * npi + c0 + lane];
        __syncthreads();
        if (j0 + lane < npj) {
            for (int c = w - 1; c >= 0; --c) {
                if (c0 + c == npi - 1) carry = 0.0;       // the eastern column itself is 0
                else carry = __dadd_rn(carry, -tile[lane][c]);
                tile[lane][c] = carry;
            }
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r)
            if (j0 + r < npj && lane < w) dpsi[(int64_t)(j0 + r) * npi + c0 + lane] = tile[r][lane];
        __syncthreads();
    }
}

// src/cdfpsi.f90:412-438: optional mean of the two estimates, reference point, SNGL, optional mask
__global__ void psi_final_kernel(const double *__restrict__ du, const double *__restrict__ dv, int64_t nij,
                                 int64_t pref, int use_ref, double ref_value, const float *__restrict__ fmask,
                                 float *__restrict__ psi4, double *__restrict__ dpsi8) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    auto val = [&](int64_t q) {
        if (du && dv) return __dmul_rn(0.5, __dadd_rn(du[q], dv[q]));
        return du ? du[q] : dv[q];
    };
    const double d = __dadd_rn(val(p), -(use_ref ? ref_value : val(pref)));
    if (dpsi8) dpsi8[p] = d;
    psi4[p] = fmask ? __fmul_rn((float)d, fmask[p]) : (float)d;
}

// cdfpsi -open, src/cdfpsi.f90:335-356: psi started from a line inside the domain.
// line kernels: one thread walks the starting line (npi or npj sequential additions).
__global__ void psi_open_line_u_kernel(const double *__restrict__ dtrpv, double *__restrict__ dpsiu, int npi, int J) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double acc = dtrpv[(int64_t)J * npi];                    // dpsiu(1,J) = dtrpv(1,J)
    dpsiu[(int64_t)J * npi] = acc;
    for (int i = 1; i < npi; ++i) {
        acc = __dadd_rn(acc, dtrpv[(int64_t)J * npi + i]);   // dpsiu(ji,J) = dpsiu(ji-1,J) + dtrpv(ji,J)
        dpsiu[(int64_t)J * npi + i] = acc;
    }
}
// southwards from row J: dpsiu(ji,jj) = dpsiu(ji,jj+1) + dtrpu(ji,jj+1); rows north of J are 0 (:259)
__global__ void psi_open_march_u_kernel(const double *__restrict__ dtrpu, double *__restrict__ dpsiu, int npi, int npj, int J) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npi) return;
    for (int j = npj - 1; j > J; --j) dpsiu[(int64_t)j * npi + i] = 0.0;
    double acc = dpsiu[(int64_t)J * npi + i];
    for (int j = J - 1; j >= 0; --j) {
        acc = __dadd_rn(acc, dtrpu[(int64_t)(j + 1) * npi + i]);
        dpsiu[(int64_t)j * npi + i] = acc;
    }
}
// the other estimate (-mean): column I from the north, dpsiv(I,npj) = dtrpu(I,npj), then + dtrpu(I,jj+1)
__global__ void psi_open_line_v_kernel(const double *__restrict__ dtrpu, double *__restrict__ dpsiv, int npi, int npj, int I) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double acc = dtrpu[(int64_t)(npj - 1) * npi + I];
    dpsiv[(int64_t)(npj - 1) * npi + I] = acc;
    for (int j = npj - 2; j >= 0; --j) {
        acc = __dadd_rn(acc, dtrpu[(int64_t)(j + 1) * npi + I]);
        dpsiv[(int64_t)j * npi + I] = acc;
    }
}
// westwards from column I: dpsiv(ji,jj) = dpsiv(ji+1,jj) - dtrpv(ji+1,jj); columns east of I are 0
__global__ void psi_open_march_v_kernel(const double *__restrict__ dtrpv, double *__restrict__ dpsiv, int npi, int npj, int I) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npj) return;
    double *row = dpsiv + (int64_t)j * npi;
    const double *tr = dtrpv + (int64_t)j * npi;
    for (int i = npi - 1; i > I; --i) row[i] = 0.0;
    double acc = row[I];
    for (int i = I - 1; i >= 0; --i) {
        acc = __dadd_rn(acc, -tr[i + 1]);
        row[i] = acc;
    }
}

int dmalloc(void **p, size_t bytes) {
    if (hipMalloc(p, bytes + 64) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        return fail(CDFT_ERR_ALLOC, "device allocation of %zu bytes failed", bytes);
    }
    return CDFT_OK;
}

int upload(void **dst, const void *src, size_t bytes, int space, hipStream_t st) {
    if (!*dst) CDFT_TRY(dmalloc(dst, bytes));
    CDFT_HIP(hipMemcpyAsync(*dst, src, bytes, space == CDFT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    return CDFT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------ cdfmhst
namespace cdft { namespace shard {
int mhst_parts(ShardSet &sh);
int mhst_set_mesh(ShardSet &sh, const float *e1v, const float *e3v, const float *e31d, const float *masks);
int mhst_compute(ShardSet &sh, const float *vt, const float *vs, const float *zv, const float *zt, const float *zs, int zdim,
                 double *heat, double *salt);
int psi_parts(ShardSet &sh);
int psi_integrate(ShardSet &sh, const float *zu, const float *e2u, const float *e3u, const float *zv, const float *e1v,
                  const float *e3v, const float *e31d);
int psi_scan_u(ShardSet &sh);
int psi_value(ShardSet &sh, int ii, int ij, double *value);
int psi_finish(ShardSet &sh, int use_ref, double ref_value, int iiref, int ijref, const float *fmask, float *psi4, double *dpsi8);
} }

struct cdft_mhst_plan {
    cdft::ShardSet *sh = nullptr;   // set: a shell over one plan per row block (shard.hpp)
    int npi = 0, npj = 0, npk = 0, nb = 0;
    float *e1v = nullptr, *e3v = nullptr, *e31d = nullptr, *masks = nullptr;
    uint8_t *mbits = nullptr;                            // masks packed to a byte per cell when they are all 0 / 1
    bool use_bits = false;
    double *dtrph = nullptr, *dtrps = nullptr, *oh = nullptr, *os = nullptr;
    bool mesh_set = false;
    double *hout = nullptr, *hout_dev = nullptr;         // page-locked, device-mapped (heat, salt) of a host call: the zonal kernel
    size_t hout_bytes = 0;                               // writes the result where the host reads it (as cdfheatc)
    DevBuf s0, s1, s2;
};

extern "C" {

int cdft_mhst_create(cdft_mhst_plan **plan, int npiglo, int npjglo, int npk, int nbasins) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_mhst_create: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 1 || nbasins < 1 || nbasins > 8)
        return fail(CDFT_ERR_ARG, "cdft_mhst_create: bad sizes %d x %d x %d, %d basins", npiglo, npjglo, npk, nbasins);
    CDFT_TRY(require_device());
    cdft_mhst_plan *p = new cdft_mhst_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk; p->nb = nbasins;
    const size_t nij = (size_t)npiglo * npjglo;
    int rc = dmalloc((void **)&p->dtrph, nij * 8);
    if (!rc) rc = dmalloc((void **)&p->dtrps, nij * 8);
    if (!rc) rc = dmalloc((void **)&p->oh, (size_t)nbasins * npjglo * npk * 8);
    if (!rc) rc = dmalloc((void **)&p->os, (size_t)nbasins * npjglo * npk * 8);
    if (rc) { cdft_mhst_destroy(p); return rc; }
    *plan = p;
    return CDFT_OK;
}

int cdft_mhst_create_sharded(cdft_mhst_plan **plan, int npiglo, int npjglo, int npk, int nbasins, int ndev, const int *devices) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_mhst_create_sharded: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 1 || nbasins < 1 || nbasins > 8)
        return fail(CDFT_ERR_ARG, "cdft_mhst_create_sharded: bad sizes %d x %d x %d, %d basins", npiglo, npjglo, npk, nbasins);
    cdft_mhst_plan *p = new cdft_mhst_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk; p->nb = nbasins;
    int rc = shard::make(&p->sh, npiglo, npjglo, npk, nbasins, ndev, devices, 1, 9);     // one halo row to the north (:318)
    if (rc == CDFT_OK) rc = shard::mhst_parts(*p->sh);
    if (rc != CDFT_OK) { cdft_mhst_destroy(p); return rc; }
    *plan = p;
    return CDFT_OK;
}

int cdft_mhst_set_mesh(cdft_mhst_plan *p, const float *e1v, const float *e3v, const float *e31d,
                       const float *masks, int memspace, void *stream) {
    if (!p || !e1v || !masks || (!e3v && !e31d) || (e3v && e31d))
        return fail(CDFT_ERR_ARG, "cdft_mhst_set_mesh: need e1v, masks and exactly one of e3v / e31d");
    if (p->sh) {
        if (memspace == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
        CDFT_TRY(shard::mhst_set_mesh(*p->sh, e1v, e3v, e31d, masks));
        p->mesh_set = true;
        return CDFT_OK;
    }
    hipStream_t st = as_stream(stream);
    const size_t nij = (size_t)p->npi * p->npj;
    CDFT_TRY(upload((void **)&p->e1v, e1v, nij * 4, memspace, st));
    CDFT_TRY(upload((void **)&p->masks, masks, nij * p->nb * 4, memspace, st));
    if (e3v) {
        CDFT_TRY(upload((void **)&p->e3v, e3v, nij * p->npk * 4, memspace, st));
        if (p->e31d) { (void)hipFree(p->e31d); p->e31d = nullptr; }
    } else {
        CDFT_TRY(upload((void **)&p->e31d, e31d, (size_t)p->npk * 4, memspace, st));
        if (p->e3v) { (void)hipFree(p->e3v); p->e3v = nullptr; }
    }
    if (!p->mbits && hipMalloc((void **)&p->mbits, nij + 64) != hipSuccess) {     // the flag sits behind the bytes, 16-aligned
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdft_mhst_set_mesh: device allocation failed");
    }
    int *flag = (int *)(p->mbits + ((nij + 15) / 16) * 16);
    int not01 = 0;
    CDFT_HIP(hipMemsetAsync(flag, 0, 4, st));
    hipLaunchKernelGGL(mhst_mask_bits_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, p->masks, (int64_t)nij,
                       p->nb, p->mbits, flag);
    CDFT_LAUNCH_CHECK();
    CDFT_HIP(hipMemcpyAsync(&not01, flag, 4, hipMemcpyDeviceToHost, st));
    CDFT_HIP(hipStreamSynchronize(st));
    p->use_bits = not01 == 0;
    p->mesh_set = true;
    return CDFT_OK;
}

int cdft_mhst_compute(cdft_mhst_plan *p, const float *vt, const float *vs, const float *zv, const float *zt,
                      const float *zs, int in_space, int zdim, double *heat, double *salt, int out_space,
                      void *stream) {
    TraceRange trace_("cdft_mhst_compute");
    if (!p || !heat || !salt) return fail(CDFT_ERR_ARG, "cdft_mhst_compute: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_mhst_compute: cdft_mhst_set_mesh has not succeeded");
    const bool sep = (vt == nullptr);
    if ((sep && (!zv || !zt || !zs)) || (!sep && !vs))
        return fail(CDFT_ERR_ARG, "cdft_mhst_compute: give (vt, vs) or (zv, zt, zs)");
    if (p->sh) {
        if (in_space == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
        return shard::mhst_compute(*p->sh, vt, vs, zv, zt, zs, zdim, heat, salt);
    }
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj;
    const size_t n3b = (size_t)nij * p->npk * 4;
    const void *d0, *d1, *d2 = nullptr;
    CDFT_TRY(to_device(sep ? zv : vt, n3b, in_space, p->s0, st, &d0));
    CDFT_TRY(to_device(sep ? zt : vs, n3b, in_space, p->s1, st, &d1));
    if (sep) CDFT_TRY(to_device(zs, n3b, in_space, p->s2, st, &d2));
    VintArgs a;
    a.f0 = (const float *)d0; a.f1 = (const float *)d1; a.f2 = (const float *)d2; a.m2d = p->e1v;
    a.e3 = p->e3v; a.e31d = p->e31d; a.acc0 = p->dtrph; a.acc1 = p->dtrps; a.nij = nij; a.npi = p->npi;
    a.npj = p->npj;
    const int nlev_out = zdim ? p->npk : 1;
    const size_t out_bytes = (size_t)p->nb * p->npj * nlev_out * 8;
    double *oh = (out_space == CDFT_MEM_DEVICE) ? heat : p->oh, *os = (out_space == CDFT_MEM_DEVICE) ? salt : p->os;
    bool mapped = false;
    if (out_space == CDFT_MEM_HOST) {
        if (p->hout_bytes < 2 * out_bytes) {
            if (p->hout) { (void)hipHostFree(p->hout); p->hout = nullptr; p->hout_bytes = 0; }
            if (hipHostMalloc((void **)&p->hout, 2 * out_bytes, hipHostMallocMapped) == hipSuccess &&
                hipHostGetDevicePointer((void **)&p->hout_dev, p->hout, 0) == hipSuccess) p->hout_bytes = 2 * out_bytes;
            else { (void)hipGetLastError(); if (p->hout) { (void)hipHostFree(p->hout); p->hout = nullptr; } }
        }
        if (p->hout_bytes >= 2 * out_bytes) { oh = p->hout_dev; os = p->hout_dev + out_bytes / 8; mapped = true; }
    }
    auto zonal = [&](double *h, double *s) -> int {
        const char *e = getenv("CDFT_MHST_R4MASKS");             // test hook: the unpacked masks
        const bool bits = p->use_bits && !(e && e[0] == '1');
        if (bits && p->nb == 1)
            hipLaunchKernelGGL((mhst_zonal_bits_kernel<1>), dim3(p->npj), dim3(kThreads), 0, st, p->dtrph, p->dtrps,
                               p->mbits, p->npi, p->npj, p->nb, h, s);
        else if (bits)
            hipLaunchKernelGGL((mhst_zonal_bits_kernel<8>), dim3(p->npj), dim3(kThreads), 0, st, p->dtrph, p->dtrps,
                               p->mbits, p->npi, p->npj, p->nb, h, s);
        else if (p->nb == 1)
            hipLaunchKernelGGL((mhst_zonal_kernel<1>), dim3(p->npj), dim3(kThreads), 0, st, p->dtrph, p->dtrps,
                               p->masks, p->npi, p->npj, p->nb, h, s);
        else
            hipLaunchKernelGGL((mhst_zonal_kernel<8>), dim3(p->npj), dim3(kThreads), 0, st, p->dtrph, p->dtrps,
                               p->masks, p->npi, p->npj, p->nb, h, s);
        CDFT_LAUNCH_CHECK();
        return CDFT_OK;
    };
    if (!zdim) {
        a.k0 = 0; a.k1 = p->npk; a.init = 1;
        CDFT_TRY(sep ? launch_vint<kMhstSep>(a, st) : launch_vint<kMhstVt>(a, st));
        CDFT_TRY(zonal(oh, os));
    } else if (!sep && p->use_bits && p->npi <= 9 * 512 && !getenv("CDFT_MHST_ZDIM_LEVELS")) {
        // -Zdim from a VT file with 0/1 masks: one kernel, a row per workgroup, the running sums in registers
        const float *fvt = (const float *)d0, *fvs = (const float *)d1;
        const bool full = p->e31d != nullptr;
        const int threads = p->npi <= 6 * kThreads ? kThreads : 512, nc = (p->npi + threads - 1) / threads;
#define CDFT_ZDIM2(NCC, TT) do { \
            if (full) hipLaunchKernelGGL((mhst_zdim_kernel<NCC, true, TT>), dim3(p->npj), dim3(TT), 0, st, fvt, fvs, p->e1v, p->e3v, p->e31d, \
                                         p->mbits, p->npi, p->npj, p->npk, p->nb, oh, os); \
            else      hipLaunchKernelGGL((mhst_zdim_kernel<NCC, false, TT>), dim3(p->npj), dim3(TT), 0, st, fvt, fvs, p->e1v, p->e3v, p->e31d, \
                                         p->mbits, p->npi, p->npj, p->npk, p->nb, oh, os); } while (0)
        if (threads == kThreads) { if (nc <= 2) CDFT_ZDIM2(2, 256); else CDFT_ZDIM2(6, 256); }
        else                     { if (nc <= 6) CDFT_ZDIM2(6, 512); else CDFT_ZDIM2(9, 512); }
#undef CDFT_ZDIM2
        CDFT_LAUNCH_CHECK();
    } else {                                            // -Zdim: running totals after every level, a launch pair per level
        for (int k = 0; k < p->npk; ++k) {
            a.k0 = k; a.k1 = k + 1; a.init = (k == 0);
            CDFT_TRY(sep ? launch_vint<kMhstSep>(a, st) : launch_vint<kMhstVt>(a, st));
            CDFT_TRY(zonal(oh + (size_t)k * p->npj * p->nb, os + (size_t)k * p->npj * p->nb));
        }
    }
    if (out_space == CDFT_MEM_HOST) {
        if (mapped) {
            CDFT_HIP(hipStreamSynchronize(st));
            memcpy(heat, p->hout, out_bytes);
            memcpy(salt, p->hout + out_bytes / 8, out_bytes);
        } else {
            CDFT_HIP(hipMemcpyAsync(heat, oh, out_bytes, hipMemcpyDeviceToHost, st));
            CDFT_HIP(hipMemcpyAsync(salt, os, out_bytes, hipMemcpyDeviceToHost, st));
            CDFT_HIP(hipStreamSynchronize(st));
        }
    }
    return CDFT_OK;
}

int cdft_mhst_destroy(cdft_mhst_plan *p) {
    if (!p) return CDFT_OK;
    if (p->sh) {
        (void)shard::for_blocks(*p->sh, [&](int b) { return cdft_mhst_destroy((cdft_mhst_plan *)p->sh->part[b]); });
        shard::destroy(p->sh);
        delete p;
        return CDFT_OK;
    }
    void *ptrs[] = {p->e1v, p->e3v, p->e31d, p->masks, p->mbits, p->dtrph, p->dtrps, p->oh, p->os};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (p->hout) (void)hipHostFree(p->hout);
    p->s0.release(); p->s1.release(); p->s2.release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"

// ------------------------------------------------------------- cdfvtrp / cdfvhst
namespace { DevBuf g_vi[4], g_vo[2], g_sl[6]; }

// cdfvtrp -bathy (src/cdfvtrp.f90:210-252): the vertically integrated transports brought to the T point and rotated onto
// the direction of the bathymetric slope.  One thread per (i,j) of the plane; every operand is 2-D, so the whole option is
// 40 B per cell read once -- its cost is the launch.  Arithmetic as the reference writes it: the means in r8 (0.5 is
// exact), the slopes, their angle and its cosine / sine in r4, the rotation in r8 with the r4 factors widened.  The three
// transcendental results are formed in r8 and rounded to r4 -- the correctly rounded r4 value except on near-ties -- so the
// result does not depend on the accuracy of a single-precision device routine (the host libraries the reference meets are
// themselves within an ulp of that value: tolerance in tests/test_vtrp_bathy.py).
__global__ void __launch_bounds__(256)
slope_rotate_kernel(const double *__restrict__ dtrpu, const double *__restrict__ dtrpv, const float *__restrict__ hdepw,
                    const float *__restrict__ e1u, const float *__restrict__ e2v, const float *__restrict__ tmask,
                    int npi, int npj, double *__restrict__ dalong, double *__restrict__ dcross) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= npi || j >= npj) return;
    const size_t p = (size_t)j * npi + i;
    // dwku(1,jj) = dwku(npiglo-1,jj): the E-W overlap column takes the mean of columns npiglo-2, npiglo-1 (:218)
    const int iu = (i == 0) ? npi - 2 : i;
    double dwku = 0.0, dwkv = 0.0;
    if (iu >= 1) dwku = 0.5 * (dtrpu[(size_t)j * npi + iu] + dtrpu[(size_t)j * npi + iu - 1]);
    if (j >= 1) dwkv = 0.5 * (dtrpv[p] + dtrpv[p - npi]);
    const float tm = tmask[p];
    float zdhdx = 0.f, zdhdy = 0.f;
    if (i >= 1 && i <= npi - 2) zdhdx = __fmul_rn(__fdiv_rn(__fsub_rn(hdepw[p + 1], hdepw[p - 1]), __fadd_rn(e1u[p], e1u[p - 1])), tm);
    if (j >= 1 && j <= npj - 2) zdhdy = __fmul_rn(__fdiv_rn(__fsub_rn(hdepw[p + npi], hdepw[p - npi]), __fadd_rn(e2v[p], e2v[p - npi])), tm);
    const float zalpha = __fmul_rn((float)atan2((double)zdhdx, (double)zdhdy), tm);
    const double c = (double)(float)cos((double)zalpha), sn = (double)(float)sin((double)zalpha);
    dalong[p] = __dmul_rn(__dadd_rn(__dmul_rn(dwku, c), __dmul_rn(dwkv, sn)), (double)tm);
    dcross[p] = __dmul_rn(__dadd_rn(__dmul_rn(-dwku, sn), __dmul_rn(dwkv, c)), (double)tm);
}

extern "C" {

int cdft_vint_transport(const float *vel, const float *m2d, const float *e3, const float *e31d, int npiglo,
                        int npjglo, int npk, int in_space, double *dtrp, int out_space, void *stream) {
    if (!vel || !m2d || !dtrp || (!e3 && !e31d)) return fail(CDFT_ERR_ARG, "cdft_vint_transport: NULL argument");
    if (npiglo < 1 || npjglo < 1 || npk < 1) return fail(CDFT_ERR_ARG, "cdft_vint_transport: bad sizes");
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)npiglo * npjglo;
    const void *dv, *dm, *de = nullptr, *d1 = nullptr;
    CDFT_TRY(to_device(vel, (size_t)nij * npk * 4, in_space, g_vi[0], st, &dv));
    CDFT_TRY(to_device(m2d, (size_t)nij * 4, in_space, g_vi[1], st, &dm));
    if (e3) CDFT_TRY(to_device(e3, (size_t)nij * npk * 4, in_space, g_vi[2], st, &de));
    else    CDFT_TRY(to_device(e31d, (size_t)npk * 4, in_space, g_vi[3], st, &d1));
    double *out = dtrp;
    if (out_space == CDFT_MEM_HOST) { CDFT_TRY(g_vo[0].reserve((size_t)nij * 8)); out = (double *)g_vo[0].ptr; }
    VintArgs a;
    a.f0 = (const float *)dv; a.f1 = nullptr; a.f2 = nullptr; a.m2d = (const float *)dm; a.e3 = (const float *)de;
    a.e31d = (const float *)d1; a.acc0 = out; a.acc1 = nullptr; a.nij = nij; a.npi = npiglo; a.npj = npjglo;
    a.k0 = 0; a.k1 = npk; a.init = 1;
    CDFT_TRY(launch_vint<kPsi>(a, st));
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(dtrp, out, (size_t)nij * 8, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_vhst_transport(const float *xt, const float *xs, const float *m2d, const float *e3, const float *e31d,
                        int npiglo, int npjglo, int npk, int in_space, double *dheat, double *dsalt, int out_space,
                        void *stream) {
    if (!xt || !xs || !m2d || !dheat || !dsalt || (!e3 && !e31d)) return fail(CDFT_ERR_ARG, "cdft_vhst_transport: NULL argument");
    if (npiglo < 1 || npjglo < 1 || npk < 1) return fail(CDFT_ERR_ARG, "cdft_vhst_transport: bad sizes");
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)npiglo * npjglo;
    static DevBuf sxs;
    const void *dt, *dsx, *dm, *de = nullptr, *d1 = nullptr;
    CDFT_TRY(to_device(xt, (size_t)nij * npk * 4, in_space, g_vi[0], st, &dt));
    CDFT_TRY(to_device(xs, (size_t)nij * npk * 4, in_space, sxs, st, &dsx));
    CDFT_TRY(to_device(m2d, (size_t)nij * 4, in_space, g_vi[1], st, &dm));
    if (e3) CDFT_TRY(to_device(e3, (size_t)nij * npk * 4, in_space, g_vi[2], st, &de));
    else    CDFT_TRY(to_device(e31d, (size_t)npk * 4, in_space, g_vi[3], st, &d1));
    double *oh = dheat, *os = dsalt;
    if (out_space == CDFT_MEM_HOST) {
        CDFT_TRY(g_vo[0].reserve((size_t)nij * 8)); CDFT_TRY(g_vo[1].reserve((size_t)nij * 8));
        oh = (double *)g_vo[0].ptr; os = (double *)g_vo[1].ptr;
    }
    VintArgs a;
    a.f0 = (const float *)dt; a.f1 = (const float *)dsx; a.f2 = nullptr; a.m2d = (const float *)dm;
    a.e3 = (const float *)de; a.e31d = (const float *)d1; a.acc0 = oh; a.acc1 = os; a.nij = nij; a.npi = npiglo;
    a.npj = npjglo; a.k0 = 0; a.k1 = npk; a.init = 1;
    CDFT_TRY(launch_vint<kVhst>(a, st));
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(dheat, oh, (size_t)nij * 8, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipMemcpyAsync(dsalt, os, (size_t)nij * 8, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_vint_slope_rotate(const double *dtrpu, const double *dtrpv, const float *hdepw, const float *e1u,
                           const float *e2v, const float *tmask, int npiglo, int npjglo, int in_space, double *dalong,
                           double *dcross, int out_space, void *stream) {
    if (!dtrpu || !dtrpv || !hdepw || !e1u || !e2v || !tmask || !dalong || !dcross)
        return fail(CDFT_ERR_ARG, "cdft_vint_slope_rotate: NULL argument");
    if (npiglo < 1 || npjglo < 1) return fail(CDFT_ERR_ARG, "cdft_vint_slope_rotate: bad sizes");
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const size_t nij = (size_t)npiglo * npjglo;
    const void *du, *dv, *dh, *d1, *d2, *dm;
    CDFT_TRY(to_device(dtrpu, nij * 8, in_space, g_sl[0], st, &du));
    CDFT_TRY(to_device(dtrpv, nij * 8, in_space, g_sl[1], st, &dv));
    CDFT_TRY(to_device(hdepw, nij * 4, in_space, g_sl[2], st, &dh));
    CDFT_TRY(to_device(e1u, nij * 4, in_space, g_sl[3], st, &d1));
    CDFT_TRY(to_device(e2v, nij * 4, in_space, g_sl[4], st, &d2));
    CDFT_TRY(to_device(tmask, nij * 4, in_space, g_sl[5], st, &dm));
    double *oa = dalong, *oc = dcross;
    if (out_space == CDFT_MEM_HOST) {
        CDFT_TRY(g_vo[0].reserve(nij * 8)); CDFT_TRY(g_vo[1].reserve(nij * 8));
        oa = (double *)g_vo[0].ptr; oc = (double *)g_vo[1].ptr;
    }
    const dim3 block(64, 4), grid((unsigned)((npiglo + 63) / 64), (unsigned)((npjglo + 3) / 4));
    hipLaunchKernelGGL(slope_rotate_kernel, grid, block, 0, st, (const double *)du, (const double *)dv, (const float *)dh,
                       (const float *)d1, (const float *)d2, (const float *)dm, npiglo, npjglo, oa, oc);
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(dalong, oa, nij * 8, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipMemcpyAsync(dcross, oc, nij * 8, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_vint_release(void) {
    for (DevBuf &b : g_vi) b.release();
    for (DevBuf &b : g_vo) b.release();
    for (DevBuf &b : g_sl) b.release();
    return CDFT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ cdfpsi
struct cdft_psi_plan {
    cdft::ShardSet *sh = nullptr;   // set: a shell over one plan per row block (shard.hpp)
    int npi = 0, npj = 0, npk = 0;
    double *dtrpu = nullptr, *dtrpv = nullptr, *dpsiu = nullptr, *dpsiv = nullptr, *d8 = nullptr;
    float *p4 = nullptr;
    bool lu = false, lv = false;                       // what the last cdft_psi_integrate filled
    bool use_u = false, use_v = false;                 // which estimates cdft_psi_finish combines
    DevBuf su, se3u, sv, se3v, sm0, sm1, sfm, s1d, scar;
};

extern "C" {

int cdft_psi_create(cdft_psi_plan **plan, int npiglo, int npjglo, int npk) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_psi_create: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 1)
        return fail(CDFT_ERR_ARG, "cdft_psi_create: bad sizes %d x %d x %d", npiglo, npjglo, npk);
    CDFT_TRY(require_device());
    cdft_psi_plan *p = new cdft_psi_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk;
    const size_t nij = (size_t)npiglo * npjglo;
    int rc = dmalloc((void **)&p->dtrpu, nij * 8);
    if (!rc) rc = dmalloc((void **)&p->dtrpv, nij * 8);
    if (!rc) rc = dmalloc((void **)&p->dpsiu, nij * 8);
    if (!rc) rc = dmalloc((void **)&p->dpsiv, nij * 8);
    if (!rc) rc = dmalloc((void **)&p->d8, nij * 8);
    if (!rc) rc = dmalloc((void **)&p->p4, nij * 4);
    if (rc) { cdft_psi_destroy(p); return rc; }
    *plan = p;
    return CDFT_OK;
}

int cdft_psi_create_sharded(cdft_psi_plan **plan, int npiglo, int npjglo, int npk, int ndev, const int *devices) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_psi_create_sharded: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 1)
        return fail(CDFT_ERR_ARG, "cdft_psi_create_sharded: bad sizes %d x %d x %d", npiglo, npjglo, npk);
    cdft_psi_plan *p = new cdft_psi_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk;
    int rc = shard::make(&p->sh, npiglo, npjglo, npk, 0, ndev, devices, 0, 13);
    if (rc == CDFT_OK) rc = shard::psi_parts(*p->sh);
    if (rc != CDFT_OK) { cdft_psi_destroy(p); return rc; }
    *plan = p;
    return CDFT_OK;
}

int cdft_psi_integrate(cdft_psi_plan *p, const float *zu, const float *e2u, const float *e3u, const float *zv,
                       const float *e1v, const float *e3v, const float *e31d, int in_space, void *stream) {
    TraceRange trace_("cdft_psi_integrate");
    if (!p) return fail(CDFT_ERR_ARG, "cdft_psi_integrate: NULL plan");
    if (p->sh) {
        if ((zu && !e2u) || (zv && !e1v) || (!zu && !zv)) return fail(CDFT_ERR_ARG, "cdft_psi: each velocity needs its horizontal metric");
        if (in_space == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
        return shard::psi_integrate(*p->sh, zu, e2u, e3u, zv, e1v, e3v, e31d);
    }
    const bool lu = zu != nullptr, lv = zv != nullptr;
    if (!lu && !lv) return fail(CDFT_ERR_ARG, "cdft_psi: give the U set, the V set, or both (-mean)");
    if ((lu && (!e2u || (!e3u && !e31d))) || (lv && (!e1v || (!e3v && !e31d))))
        return fail(CDFT_ERR_ARG, "cdft_psi: each velocity needs its horizontal metric and e3 (or e31d)");
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj;
    const size_t n3b = (size_t)nij * p->npk * 4, n2b = (size_t)nij * 4;
    const void *d1d = nullptr;
    if (e31d) CDFT_TRY(to_device(e31d, (size_t)p->npk * 4, in_space, p->s1d, st, &d1d));
    auto integrate = [&](const float *z, const float *m, const float *e3, DevBuf &bz, DevBuf &be, DevBuf &bm,
                         double *acc) -> int {
        const void *dz, *dm, *de = nullptr;
        CDFT_TRY(to_device(z, n3b, in_space, bz, st, &dz));
        CDFT_TRY(to_device(m, n2b, in_space, bm, st, &dm));
        if (e3) CDFT_TRY(to_device(e3, n3b, in_space, be, st, &de));
        VintArgs a;
        a.f0 = (const float *)dz; a.f1 = nullptr; a.f2 = nullptr; a.m2d = (const float *)dm;
        a.e3 = (const float *)de; a.e31d = e3 ? nullptr : (const float *)d1d; a.acc0 = acc; a.acc1 = nullptr;
        a.nij = nij; a.npi = p->npi; a.npj = p->npj; a.k0 = 0; a.k1 = p->npk; a.init = 1;
        return launch_vint<kPsi>(a, st);
    };
    if (lv) {                                            // the i-scan stays inside a row: local to a row block
        CDFT_TRY(integrate(zv, e1v, e3v, p->sv, p->se3v, p->sm1, p->dtrpv));
        const char *ew = getenv("CDFT_PSI_SCAN_WAVE");            // (1: the one-wave-per-64-rows scan, tests compare the two)
        if (ew && ew[0] == '1') {
            hipLaunchKernelGGL(psi_scan_i_kernel, dim3((p->npj + 63) / 64), dim3(64), 0, st, p->dtrpv, p->dpsiv, p->npi, p->npj);
        } else {
            const size_t shmem = sizeof(double) * kRingSlots * kRingRows * kRingIStr + sizeof(int) * (2 * kRingCnt + 2);
            CDFT_TRY(dyn_lds_once((const void *)psi_scan_i_ring_kernel, (int)shmem));
            hipLaunchKernelGGL(psi_scan_i_ring_kernel, dim3((p->npj + kRingCW - 1) / kRingCW), dim3((1 + kRingIFeed + kRingIDrain) * 64), shmem, st,
                               p->dtrpv, p->dpsiv, p->npi, p->npj);
        }
        CDFT_LAUNCH_CHECK();
    }
    if (lu) CDFT_TRY(integrate(zu, e2u, e3u, p->su, p->se3u, p->sm0, p->dtrpu));
    p->lu = lu; p->lv = lv;
    p->use_u = lu; p->use_v = lv;
    return CDFT_OK;
}

int cdft_psi_scan_u(cdft_psi_plan *p, const double *carry, int carry_space, int first_block, double *last_row,
                    int last_space, void *stream) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_psi_scan_u: NULL plan");
    if (p->sh) {
        if (carry || last_row || !first_block)
            return fail(CDFT_ERR_ARG, "cdft_psi_scan_u: a sharded plan chains its own blocks; it takes no carry and returns no row");
        return shard::psi_scan_u(*p->sh);
    }
    if (!p->lu) return CDFT_OK;                          // V-only: nothing runs along j
    hipStream_t st = as_stream(stream);
    const void *dc = nullptr;
    if (carry) CDFT_TRY(to_device(carry, (size_t)p->npi * 8, carry_space, p->scar, st, &dc));
    const char *ew = getenv("CDFT_PSI_SCAN_WAVE");                // the one-wave-per-64-columns scan (tests compare the two)
    if (ew && ew[0] == '1')
        hipLaunchKernelGGL(psi_scan_j_kernel, dim3((p->npi + 63) / 64), dim3(64), 0, st, p->dtrpu, p->dpsiu, p->npi, p->npj,
                           (const double *)dc, first_block ? 1 : 0);
    else if (ew && ew[0] == '2')                                  // the round-3 scan: 128 rows staged, walked and stored in turn
        hipLaunchKernelGGL(psi_scan_j_wg_kernel, dim3((p->npi + kScanCW - 1) / kScanCW), dim3(kScanWaves * 64), 0, st, p->dtrpu, p->dpsiu,
                           p->npi, p->npj, (const double *)dc, first_block ? 1 : 0);
    else {
        const size_t shmem = sizeof(double) * kRingSlots * kRingRows * kRingCW + sizeof(int) * (2 * kRingCnt + 2);
        // (more than 64 KB of dynamic LDS has to be asked for, on the device the plan lives on)
        CDFT_TRY(dyn_lds_once((const void *)psi_scan_j_ring_kernel, (int)shmem));
        hipLaunchKernelGGL(psi_scan_j_ring_kernel, dim3((p->npi + kRingCW - 1) / kRingCW), dim3((1 + kRingFeed + kRingDrain) * 64), shmem, st,
                           p->dtrpu, p->dpsiu, p->npi, p->npj, (const double *)dc, first_block ? 1 : 0);
    }
    CDFT_LAUNCH_CHECK();
    if (last_row) {
        const double *src = p->dpsiu + (int64_t)(p->npj - 1) * p->npi;
        CDFT_HIP(hipMemcpyAsync(last_row, src, (size_t)p->npi * 8,
                                last_space == CDFT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
        if (last_space == CDFT_MEM_HOST) CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_psi_scan_open(cdft_psi_plan *p, int lmean, void *stream) {
    TraceRange trace_("cdft_psi_scan_open");
    if (!p) return fail(CDFT_ERR_ARG, "cdft_psi_scan_open: NULL plan");
    if (p->sh)
        return fail(CDFT_ERR_STATE, "cdft_psi_scan_open: -open integrates along one row of the whole domain; use a plan of cdft_psi_create");
    if (!p->lu || !p->lv) return fail(CDFT_ERR_STATE, "cdft_psi_scan_open: -open needs both transports (cdft_psi_integrate with U and V)");
    if (p->npi < 3 || p->npj < 3) return fail(CDFT_ERR_ARG, "cdft_psi_scan_open: npiglo, npjglo >= 3 required");
    hipStream_t st = as_stream(stream);
    const int J = p->npj - 3, I = p->npi - 3;               // row npjglo-2 / column npiglo-2, 0-based
    hipLaunchKernelGGL(psi_open_line_u_kernel, dim3(1), dim3(64), 0, st, p->dtrpv, p->dpsiu, p->npi, J);
    hipLaunchKernelGGL(psi_open_march_u_kernel, dim3((p->npi + 63) / 64), dim3(64), 0, st, p->dtrpu, p->dpsiu, p->npi, p->npj, J);
    CDFT_LAUNCH_CHECK();
    if (lmean) {
        hipLaunchKernelGGL(psi_open_line_v_kernel, dim3(1), dim3(64), 0, st, p->dtrpu, p->dpsiv, p->npi, p->npj, I);
        hipLaunchKernelGGL(psi_open_march_v_kernel, dim3((p->npj + 63) / 64), dim3(64), 0, st, p->dtrpv, p->dpsiv, p->npi, p->npj, I);
        CDFT_LAUNCH_CHECK();
    }
    p->use_u = true; p->use_v = lmean != 0;
    return CDFT_OK;
}

int cdft_psi_value(cdft_psi_plan *p, int ii, int ij, double *value, void *stream) {
    if (!p || !value) return fail(CDFT_ERR_ARG, "cdft_psi_value: NULL argument");
    if (p->sh) return shard::psi_value(*p->sh, ii, ij, value);
    if (ii < 1 || ii > p->npi || ij < 1 || ij > p->npj)
        return fail(CDFT_ERR_ARG, "cdft_psi_value: point (%d,%d) outside the local block", ii, ij);
    hipStream_t st = as_stream(stream);
    const int64_t q = (int64_t)(ij - 1) * p->npi + (ii - 1);
    double u = 0.0, v = 0.0;
    if (p->use_u) CDFT_HIP(hipMemcpyAsync(&u, p->dpsiu + q, 8, hipMemcpyDeviceToHost, st));
    if (p->use_v) CDFT_HIP(hipMemcpyAsync(&v, p->dpsiv + q, 8, hipMemcpyDeviceToHost, st));
    CDFT_HIP(hipStreamSynchronize(st));
    *value = (p->use_u && p->use_v) ? 0.5 * (u + v) : (p->use_u ? u : v);    // as psi_final_kernel's val()
    return CDFT_OK;
}

int cdft_psi_finish(cdft_psi_plan *p, int use_ref, double ref_value, int iiref, int ijref, const float *fmask,
                    int mask_space, float *psi4, double *dpsi8, int out_space, void *stream) {
    TraceRange trace_("cdft_psi_finish");
    if (!p || !psi4) return fail(CDFT_ERR_ARG, "cdft_psi_finish: NULL argument");
    if (p->sh) {
        if (fmask && mask_space == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
        return shard::psi_finish(*p->sh, use_ref, ref_value, iiref, ijref, fmask, psi4, dpsi8);
    }
    if (!p->lu && !p->lv) return fail(CDFT_ERR_STATE, "cdft_psi_finish: cdft_psi_integrate has not run");
    if (!use_ref && (iiref < 1 || iiref > p->npi || ijref < 1 || ijref > p->npj))
        return fail(CDFT_ERR_ARG, "cdft_psi: reference point (%d,%d) outside the domain", iiref, ijref);
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj;
    const size_t n2b = (size_t)nij * 4;
    const void *dfm = nullptr;
    if (fmask) CDFT_TRY(to_device(fmask, n2b, mask_space, p->sfm, st, &dfm));
    float *o4 = out_space == CDFT_MEM_DEVICE ? psi4 : p->p4;
    double *o8 = dpsi8 ? (out_space == CDFT_MEM_DEVICE ? dpsi8 : p->d8) : nullptr;
    const int64_t pref = use_ref ? 0 : (int64_t)(ijref - 1) * p->npi + (iiref - 1);
    hipLaunchKernelGGL(psi_final_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st,
                       p->use_u ? p->dpsiu : nullptr, p->use_v ? p->dpsiv : nullptr, nij, pref, use_ref ? 1 : 0, ref_value,
                       (const float *)dfm, o4, o8);
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(psi4, o4, n2b, hipMemcpyDeviceToHost, st));
        if (dpsi8) CDFT_HIP(hipMemcpyAsync(dpsi8, o8, (size_t)nij * 8, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_psi_compute(cdft_psi_plan *p, const float *zu, const float *e2u, const float *e3u, const float *zv,
                     const float *e1v, const float *e3v, const float *e31d, int iiref, int ijref,
                     const float *fmask, int in_space, float *psi4, double *dpsi8, int out_space, void *stream) {
    TraceRange trace_("cdft_psi_compute");
    if (!p || !psi4) return fail(CDFT_ERR_ARG, "cdft_psi_compute: NULL argument");
    if (iiref < 1 || iiref > p->npi || ijref < 1 || ijref > p->npj)
        return fail(CDFT_ERR_ARG, "cdft_psi_compute: reference point (%d,%d) outside the domain", iiref, ijref);
    CDFT_TRY(cdft_psi_integrate(p, zu, e2u, e3u, zv, e1v, e3v, e31d, in_space, stream));
    CDFT_TRY(cdft_psi_scan_u(p, nullptr, CDFT_MEM_HOST, 1, nullptr, CDFT_MEM_HOST, stream));
    return cdft_psi_finish(p, 0, 0.0, iiref, ijref, fmask, in_space, psi4, dpsi8, out_space, stream);
}

int cdft_psi_destroy(cdft_psi_plan *p) {
    if (!p) return CDFT_OK;
    if (p->sh) {
        (void)shard::for_blocks(*p->sh, [&](int b) { return cdft_psi_destroy((cdft_psi_plan *)p->sh->part[b]); });
        shard::destroy(p->sh);
        delete p;
        return CDFT_OK;
    }
    void *ptrs[] = {p->dtrpu, p->dtrpv, p->dpsiu, p->dpsiv, p->d8, p->p4};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    DevBuf *bufs[] = {&p->su, &p->se3u, &p->sv, &p->se3v, &p->sm0, &p->sm1, &p->sfm, &p->s1d, &p->scar};
    for (DevBuf *b : bufs) b->release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"
