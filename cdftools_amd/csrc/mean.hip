// mean.hip -- cdfmean / cdfsum weighted sums per level and sub-basin (C ABI section "cdfmean").
// Reference loop src/cdfmean.f90:415-431, per level and basin:
//     zv     = zv * zmask                                                   (:410)
//     dsurf  = SUM(DBLE(          e1 * e2      * zmask * ibmask(b,:,:)))
//     dvol2d = SUM(DBLE(          e1 * e2 * e3 * zmask * ibmask(b,:,:)))
//     dsum2d = SUM(DBLE(zv      * e1 * e2 * e3         * ibmask(b,:,:)))
//     dvar2d = SUM(DBLE(zv * zv * e1 * e2 * e3         * ibmask(b,:,:)))
// Every product is formed in r4, left to right, exactly as written, then widened; the r8 SUMs are
// regrouped in a fixed order (|d| <= 1e-13 * sum|terms|).  Same layout of the work as cdfheatc: one
// workgroup owns a set of cells of the horizontal plane for ALL levels, keeps e1, e2 and the basin
// values in registers and streams zv / e3 / mask past them (12 bytes per cell and level).
#include <cstdlib>
#include <cstring>
#include "common.hpp"
#include <type_traits>
#include "reduce.hpp"

using namespace cdft;

struct cdft_mean_plan {
    int npi = 0, npj = 0, npk = 0, nb = 0;
    float *e1 = nullptr, *e2 = nullptr, *e3 = nullptr, *e31d = nullptr, *mask = nullptr;
    uint8_t *mask8 = nullptr;      // zmask as bytes when it holds only +0 and 1 (9 bytes per cell and level instead of 12)
    bool mask01 = false;
    int16_t *ib = nullptr;          // (nb,npi,npj), basin index fastest
    uint8_t *ibits = nullptr;       // (npi,npj) bit b = ibmask(b) when every value is 0 or 1 (ib01): one byte per cell instead of nb values
    bool ib01 = false;
    double *partial = nullptr;      // [npk][nblk][nb][4]
    int nblk = 0;
    int form = 0;                   // CDFT_MEAN_FORM_*
    bool mesh_set = false;
    double *hsums = nullptr, *hsums_dev = nullptr;   // page-locked, device-mapped: the final kernel writes host results where the host reads them (as cdfheatc)
    size_t hsums_bytes = 0;
    DevBuf st_zv, st_out;
};

namespace {

constexpr int kThreads = 256;
constexpr int kVPT     = 2;     // vectors per lane of the scalar path; the 16-byte path takes 4.  The plan's partial sums are sized for 2
constexpr int kMaxNB   = 8;

struct MeanArgs {
    const float *e1, *e2, *e3, *e31d, *mask, *zv;
    const uint8_t *mask8;          // or null
    const int16_t *ib;
    const uint8_t *ibits;          // or null
    int64_t nij;
    int npk, nb, nblk, form, band, redoff;
    double *partial;
};

// NB: basin capacity of the instantiation (1, 2, 4 or 8): every loop over basins is unrolled so that the
// accumulators stay in registers (a run-time bound put them in scratch memory: 6x slower)
// mask8 = zmask as bytes; flag[0] is raised by a value that is neither +0 nor 1
__global__ void mean_mask8_kernel(const float *__restrict__ m, int64_t n, uint8_t *__restrict__ m8, int *__restrict__ flag) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float v = m[p];
    if (v == 1.f) m8[p] = 1;
    else { m8[p] = 0; if (!(v == 0.f) || signbit(v)) flag[0] = 1; }
}

// bits[p] = the basins cell p belongs to; flag[0] is raised by a basin value that is neither 0 nor 1
__global__ void mean_ibits_kernel(const int16_t *__restrict__ ib, int64_t nij, int nb, uint8_t *__restrict__ bits, int *__restrict__ flag) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    unsigned v = 0;
    for (int b = 0; b < nb; ++b) {
        const int x = ib[p * nb + b];
        if (x == 1) v |= 1u << b;
        else if (x != 0) flag[0] = 1;
    }
    bits[p] = (uint8_t)v;
}

// Two cells per vector instruction where the cells come four at a time (v_pk_mul_f32: each half the plain IEEE r4 product):
// Lanes<4> works on pairs, Lanes<1> on single cells; the r8 sums take the cells in the same order either way.
typedef float f2v __attribute__((ext_vector_type(2)));
template <int VEC> struct Lanes;
template <> struct Lanes<4> {
    typedef f2v T; static constexpr int G = 2, W = 2;
    static __device__ __forceinline__ T mk(float a, float b) { return T{a, b}; }
    static __device__ __forceinline__ void add(double &s, T t) { s += (double)t.x; s += (double)t.y; }
    static __device__ __forceinline__ float at(T t, int h) { return h ? t.y : t.x; }
};
template <> struct Lanes<1> {
    typedef float T; static constexpr int G = 1, W = 1;
    static __device__ __forceinline__ T mk(float a, float) { return a; }
    static __device__ __forceinline__ void add(double &s, T t) { s += (double)t; }
    static __device__ __forceinline__ float at(T t, int) { return t; }
};

// PARK: the wave sums of every level stay in LDS [wave][level][nb*4] and the four waves are combined once, after the
// last level -- no workgroup barrier inside the level loop (the layout of cdfheatc's level kernel).  Without it (when
// npk * nb * 4 sums per wave do not fit): two barriers per level.
// The form of the sums (cdfmean with / without the variance, cdfsum's two) picks one of four straight-line copies of the
// cell loop by a uniform branch per level: no select and no scalar branch per cell.
// IB01: every basin value is 0 or 1 (the usual sub-basin file): a cell's basins are one byte in a register instead of NB r4
// values, and a term reaches a basin's sum as fma(term, 0.d0 | 1.d0, sum) -- one instruction, the same r8 value as adding the
// r4 product term * 1 or term * 0 (a zero of either sign leaves the sum, a NaN / Inf term gives NaN both ways).  Five basins:
// 57 -> ~40 vector instructions per cell, and no NB x 16 registers of basin values.
template <int VEC, bool FULL, int NB, bool PARK, int VPT, bool MASK8, bool IB01>
__global__ void __launch_bounds__(kThreads) mean_level_kernel(MeanArgs a) {
    extern __shared__ double wsum[];                   // PARK: [nwaves][npk][nb*4], else [nwaves][nb*4]; then (NB >= 2) the reduction scratch
    typedef Lanes<VEC> LN;
    typedef typename LN::T T;
    constexpr int G = LN::G, W = LN::W;                // groups per vector, cells per group
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = kThreads / 64;
    double *red = wsum + a.redoff;                     // [nwaves][8][kSum8Stride] (reduce.hpp)
    (void)red;
    const int64_t blk = xcd_band(a.band);              // (common.hpp)
    int64_t q[VPT];
    T e1[VPT][G], e2[VPT][G], ar[VPT][G], fb[IB01 ? 1 : VPT][G][IB01 ? 1 : NB];
    uint32_t bb[VPT];                                  // IB01: the basin bytes of the vector's cells
    bool ok[VPT][VEC];
    bool edge = false;                                 // a lane of this wave holds a cell that is not its own
    // VEC = 4: vectors at cells 0, 4, 8, ... and, when the plane is not a multiple of 4 cells, one more anchored at
    // nij-4 whose leading cells belong to the previous vector and count as zeros here: no partial vector, no branch
    const int64_t nfull = a.nij / VEC, nvtot = nfull + ((VEC == 4 && (a.nij & 3)) ? 1 : 0);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int64_t iv = ((blk * nwaves + wave) * VPT + v) * 64 + lane;                     // vector index
        q[v] = iv < nvtot ? (VEC == 4 ? min(iv * 4, a.nij - 4) : iv * VEC) : 0;                  // its first cell
        float x1[VEC], x2[VEC], xf[VEC][NB];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const int64_t p = q[v] + c;
            ok[v][c] = iv < nvtot && (iv < nfull || p >= nfull * VEC);
            edge = edge || !ok[v][c];
            const int64_t pc = ok[v][c] ? p : 0;
            x1[c] = ok[v][c] ? a.e1[pc] : 0.f;         // cells past the plane contribute exact zeros
            x2[c] = a.e2[pc];
            if (IB01) {
                if (c == 0) bb[v] = 0u;
                bb[v] |= (ok[v][c] ? (uint32_t)a.ibits[pc] : 0u) << (8 * c);
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b) xf[c][b] = (b < a.nb) ? (float)a.ib[pc * a.nb + min(b, a.nb - 1)] : 0.f;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            e1[v][g] = LN::mk(x1[W * g], x1[W * g + W - 1]);
            e2[v][g] = LN::mk(x2[W * g], x2[W * g + W - 1]);
            ar[v][g] = e1[v][g] * e2[v][g];
            if (!IB01) {
#pragma unroll
                for (int b = 0; b < NB; ++b) fb[v][g][b] = LN::mk(xf[W * g][b], xf[W * g + W - 1][b]);
            }
        }
    }
    edge = __any(edge);                                // (wave-uniform)
    for (int k = 0; k < a.npk; ++k) {
        const int64_t base = (int64_t)k * a.nij;
        if constexpr (IB01) {                              // (the 0.d0 / 1.d0 words are formed from the bytes at every level: hoisted out of
#pragma unroll                                             //  the loop they are NB registers per cell again -- 196 VGPRs with five basins)
            for (int v = 0; v < VPT; ++v) asm volatile("" : "+v"(bb[v]));
        }
        T zv[VPT][G], e3[VPT][G], mk[VPT][G];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if constexpr (VEC == 4) {                  // 16-byte loads at r4 alignment, always inside the level
                const f4u z = ld_stream4u(a.zv + base + q[v]);
                zv[v][0] = T{z.x, z.y}; zv[v][1] = T{z.z, z.w};
                if (MASK8) {                           // 4 mask bytes (an unaligned dword): the same 0.f / 1.f.  (A template
                                                       // parameter: as a run-time branch every mask load was followed by a wait for
                                                       // all loads issued so far -- one round trip per vector instead of per level)
                    typedef uint32_t u32u __attribute__((aligned(1)));
                    const uint32_t m = __builtin_nontemporal_load(reinterpret_cast<const u32u *>(a.mask8 + base + q[v]));
                    mk[v][0] = T{(float)(m & 0xffu), (float)((m >> 8) & 0xffu)};
                    mk[v][1] = T{(float)((m >> 16) & 0xffu), (float)(m >> 24)};
                } else {
                    const f4u m = ld_stream4u(a.mask + base + q[v]);
                    mk[v][0] = T{m.x, m.y}; mk[v][1] = T{m.z, m.w};
                }
                if (!FULL) {
                    const f4u e = ld_stream4u(a.e3 + base + q[v]);
                    e3[v][0] = T{e.x, e.y}; e3[v][1] = T{e.z, e.w};
                }
                if (edge) {                            // not this lane's cell: an exact zero
                    if (!ok[v][0]) zv[v][0].x = 0.f;
                    if (!ok[v][1]) zv[v][0].y = 0.f;
                    if (!ok[v][2]) zv[v][1].x = 0.f;
                    if (!ok[v][3]) zv[v][1].y = 0.f;
                }
            } else {
                const int64_t pc = ok[v][0] ? q[v] : 0;
                zv[v][0] = LN::mk(ok[v][0] ? a.zv[base + pc] : 0.f, 0.f);
                mk[v][0] = LN::mk(a.mask[base + pc], 0.f);
                if (!FULL) e3[v][0] = LN::mk(a.e3[base + pc], 0.f);
            }
            if (FULL) {
                const float z = a.e31d[k];
#pragma unroll
                for (int g = 0; g < G; ++g) e3[v][g] = LN::mk(z, z);
            }
        }
        double s[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b) { s[b][0] = s[b][1] = s[b][2] = s[b][3] = 0.0; }
        auto cells = [&](auto sumform, auto want_var) {
            constexpr int SF = decltype(sumform)::value;
            constexpr bool VAR = decltype(want_var)::value;
#pragma unroll
            for (int v = 0; v < VPT; ++v)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const T m = mk[v][g];
                    const T zm = zv[v][g] * m;                                                 // :410
                    const T t_surf = ar[v][g] * m;                                             // e1*e2*zmask
                    const T t_vol  = (ar[v][g] * e3[v][g]) * m;                                // e1*e2*e3*zmask
                    const T zee = (zm * e1[v][g]) * e2[v][g];                                  // zv*e1*e2
                    // cdfmean: zv*e1*e2*e3 (:421); cdfsum: zv*e1*e2 (src/cdfsum.f90:309), forcing field: zv*e1*e2*zmask (:320)
                    const T t_sum  = SF == 0 ? zee * e3[v][g] : (SF == 1 ? zee : zee * m);
                    T t_var = t_sum;
                    if (VAR) t_var = (((zm * zm) * e1[v][g]) * e2[v][g]) * e3[v][g];
                    if constexpr (IB01) {
#pragma unroll
                        for (int h = 0; h < W; ++h) {           // the cells of the group in order
                            const int c = W * g + h;
                            const double d0 = (double)LN::at(t_surf, h), d1 = (double)LN::at(t_vol, h), d2 = (double)LN::at(t_sum, h);
                            const double d3 = VAR ? (double)LN::at(t_var, h) : 0.0;
#pragma unroll
                            for (int b = 0; b < NB; ++b) {
                                // 1.d0 where bit b of the cell's byte is set, else 0.d0: the high word is 0x3ff00000 or 0
                                const int hi = ((int)(bb[v] << (31 - 8 * c - b)) >> 31) & 0x3ff00000;
                                const double m = __hiloint2double(hi, 0);
                                s[b][0] = __fma_rn(d0, m, s[b][0]);
                                s[b][1] = __fma_rn(d1, m, s[b][1]);
                                s[b][2] = __fma_rn(d2, m, s[b][2]);
                                if (VAR) s[b][3] = __fma_rn(d3, m, s[b][3]);
                            }
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < NB; ++b) {         // basins past nb have f = 0: they add exact zeros
                            const T f = fb[v][g][b];
                            LN::add(s[b][0], t_surf * f);
                            LN::add(s[b][1], t_vol * f);
                            LN::add(s[b][2], t_sum * f);
                            if (VAR) LN::add(s[b][3], t_var * f);
                        }
                    }
                }
        };
        typedef std::integral_constant<int, 0> F0;
        typedef std::integral_constant<int, 1> F1;
        typedef std::integral_constant<int, 2> F2;
        if (a.form == CDFT_MEAN_FORM_MEAN)            cells(F0{}, std::true_type{});
        else if (a.form == CDFT_MEAN_FORM_MEAN_NOVAR) cells(F0{}, std::false_type{});
        else if (a.form == 1)                         cells(F1{}, std::false_type{});
        else                                          cells(F2{}, std::false_type{});
        if constexpr (NB >= 2) {
            // Two basins' four sums at a time through LDS: lane l adds 8 lanes' values of sum l / 8, three DPP steps add the 8
            // lanes of a sum -- 33 instructions for 8 wave sums instead of 8 x 20 (five basins: 15 sums per level and wave were
            // as much work as the level's cells).  A fixed order, like the DPP tree it replaces.
            double *scr = red + wave * (8 * kSum8Stride);
#pragma unroll
            for (int b0 = 0; b0 < NB; b0 += 2) {
                double v8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = (b0 + (j >> 2) < NB) ? s[(b0 + (j >> 2)) < NB ? b0 + (j >> 2) : 0][j & 3] : 0.0;
                const double t = wave_sum8(v8, scr, lane);        // (reduce.hpp: the total of sum n in the lanes 8 n .. 8 n + 7)
                const int n = lane >> 3, b = b0 + (n >> 2), c = n & 3;
                if ((lane & 7) == 0 && b < a.nb && b < NB) wsum[((PARK ? wave * a.npk + k : wave) * a.nb + b) * 4 + c] = (c < 3 || a.form == CDFT_MEAN_FORM_MEAN) ? t : 0.0;
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double t = (c < 3 || a.form == CDFT_MEAN_FORM_MEAN) ? wave_sum(s[b][c]) : 0.0;       // (uniform)
                    if (lane == 0 && b < a.nb) wsum[((PARK ? wave * a.npk + k : wave) * a.nb + b) * 4 + c] = t;
                }
        }
        if (!PARK) {
            __syncthreads();
            if ((int)threadIdx.x < a.nb * 4) {
                double t = 0.0;
                for (int w = 0; w < nwaves; ++w) t += wsum[w * a.nb * 4 + threadIdx.x];
                if (blk < a.nblk) a.partial[((int64_t)k * a.nblk + blk) * a.nb * 4 + threadIdx.x] = t;
            }
            __syncthreads();
        }
    }
    if (PARK) {
        __syncthreads();
        const int nq = a.nb * 4;
        for (int t = threadIdx.x; t < a.npk * nq; t += kThreads) {
            double sum = 0.0;
            for (int w = 0; w < nwaves; ++w) sum += wsum[(int64_t)w * a.npk * nq + t];     // waves in order, as above
            const int k = t / nq, qq = t - k * nq;
            if (blk < a.nblk) a.partial[((int64_t)k * a.nblk + blk) * nq + qq] = sum;
        }
    }
}

// sums[k][nb][4] = the partials of level k added over the blocks in block order
__global__ void __launch_bounds__(kThreads) mean_final_kernel(const double *__restrict__ partial, int nblk, int nq,
                                                              double *__restrict__ sums) {
    __shared__ double scratch[16];
    const int k = blockIdx.x;
    for (int c = 0; c < nq; ++c) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblk; b += kThreads) s += partial[((int64_t)k * nblk + b) * nq + c];
        const double t = block_sum(s, scratch);
        if (threadIdx.x == 0) sums[(int64_t)k * nq + c] = t;
        __syncthreads();
    }
}

int keep(void **dst, const void *src, size_t bytes, int space, hipStream_t st) {
    if (!*dst && hipMalloc(dst, bytes + 64) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdfmean: device allocation of %zu bytes failed", bytes);
    }
    CDFT_HIP(hipMemcpyAsync(*dst, src, bytes, space == CDFT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    return CDFT_OK;
}

bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

extern "C" {

int cdft_mean_create(cdft_mean_plan **plan, int npiglo, int npjglo, int npk, int nbasins) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_mean_create: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 1 || nbasins < 1 || nbasins > kMaxNB)
        return fail(CDFT_ERR_ARG, "cdft_mean_create: bad sizes %d x %d x %d, %d basins (1..%d)", npiglo, npjglo, npk,
                    nbasins, kMaxNB);
    CDFT_TRY(require_device());
    cdft_mean_plan *p = new cdft_mean_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk; p->nb = nbasins;
    const int64_t nij = (int64_t)npiglo * npjglo;
    p->nblk = (int)((nij + kThreads * kVPT - 1) / (kThreads * kVPT));            // blocks when VEC = 1 (the most)
    if (hipMalloc((void **)&p->partial, (size_t)npk * p->nblk * nbasins * 4 * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        delete p;
        return fail(CDFT_ERR_ALLOC, "cdft_mean_create: device allocation failed");
    }
    *plan = p;
    return CDFT_OK;
}

int cdft_mean_set_mesh(cdft_mean_plan *p, const float *e1, const float *e2, const float *e3, const float *e31d,
                       const float *zmask, const int16_t *ibmask, int memspace, void *stream) {
    if (!p || !e1 || !e2 || !zmask || !ibmask || (!e3 && !e31d) || (e3 && e31d))
        return fail(CDFT_ERR_ARG, "cdft_mean_set_mesh: need e1, e2, zmask, ibmask and exactly one of e3 / e31d");
    hipStream_t st = as_stream(stream);
    const size_t nij = (size_t)p->npi * p->npj, n3 = nij * p->npk;
    CDFT_TRY(keep((void **)&p->e1, e1, nij * 4, memspace, st));
    CDFT_TRY(keep((void **)&p->e2, e2, nij * 4, memspace, st));
    if (e3) {
        CDFT_TRY(keep((void **)&p->e3, e3, n3 * 4, memspace, st));
        if (p->e31d) { (void)hipFree(p->e31d); p->e31d = nullptr; }
    } else {
        CDFT_TRY(keep((void **)&p->e31d, e31d, (size_t)p->npk * 4, memspace, st));
        if (p->e3) { (void)hipFree(p->e3); p->e3 = nullptr; }
    }
    CDFT_TRY(keep((void **)&p->mask, zmask, n3 * 4, memspace, st));
    CDFT_TRY(keep((void **)&p->ib, ibmask, nij * p->nb * 2, memspace, st));
    if (!p->mask8 && hipMalloc((void **)&p->mask8, n3 + 64) != hipSuccess) {       // the flag sits behind the bytes
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdft_mean_set_mesh: device allocation failed");
    }
    int *flag = (int *)(p->mask8 + ((n3 + 15) / 16) * 16);
    int not01 = 0;
    CDFT_HIP(hipMemsetAsync(flag, 0, 4, st));
    hipLaunchKernelGGL(mean_mask8_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, p->mask, (int64_t)n3, p->mask8, flag);
    CDFT_LAUNCH_CHECK();
    CDFT_HIP(hipMemcpyAsync(&not01, flag, 4, hipMemcpyDeviceToHost, st));
    CDFT_HIP(hipStreamSynchronize(st));
    p->mask01 = not01 == 0;
    // the basin values as one byte per cell when they are all 0 or 1
    if (!p->ibits && hipMalloc((void **)&p->ibits, nij + 64) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdft_mean_set_mesh: device allocation failed");
    }
    int ibnot01 = 0;
    CDFT_HIP(hipMemsetAsync(flag, 0, 4, st));
    hipLaunchKernelGGL(mean_ibits_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, p->ib, (int64_t)nij, p->nb, p->ibits, flag);
    CDFT_LAUNCH_CHECK();
    CDFT_HIP(hipMemcpyAsync(&ibnot01, flag, 4, hipMemcpyDeviceToHost, st));
    CDFT_HIP(hipStreamSynchronize(st));
    p->ib01 = ibnot01 == 0;
    p->mesh_set = true;
    return CDFT_OK;
}

int cdft_mean_set_form(cdft_mean_plan *p, int form) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_mean_set_form: NULL plan");
    if (form < CDFT_MEAN_FORM_MEAN || form > CDFT_MEAN_FORM_MEAN_NOVAR)
        return fail(CDFT_ERR_ARG, "cdft_mean_set_form: form %d outside 0..3", form);
    p->form = form;
    return CDFT_OK;
}

int cdft_mean_compute(cdft_mean_plan *p, const float *zv, int in_space, double *sums, int out_space, void *stream) {
    TraceRange trace_("cdft_mean_compute");
    if (!p || !zv || !sums) return fail(CDFT_ERR_ARG, "cdft_mean_compute: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_mean_compute: cdft_mean_set_mesh has not succeeded");
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj;
    const void *dz;
    CDFT_TRY(to_device(zv, (size_t)nij * p->npk * 4, in_space, p->st_zv, st, &dz));
    const int nq = p->nb * 4;
    const size_t ob = (size_t)p->npk * nq * 8;
    double *dout = sums;
    bool mapped = false;
    if (out_space == CDFT_MEM_HOST) {
        if (p->hsums_bytes < ob) {                                 // (once per plan: nb and npk are the plan's)
            if (p->hsums) { (void)hipHostFree(p->hsums); p->hsums = nullptr; p->hsums_bytes = 0; }
            if (hipHostMalloc((void **)&p->hsums, ob, hipHostMallocMapped) == hipSuccess &&
                hipHostGetDevicePointer((void **)&p->hsums_dev, p->hsums, 0) == hipSuccess) p->hsums_bytes = ob;
            else { (void)hipGetLastError(); if (p->hsums) { (void)hipHostFree(p->hsums); p->hsums = nullptr; } }
        }
        if (p->hsums_bytes >= ob) { dout = p->hsums_dev; mapped = true; }
        else { CDFT_TRY(p->st_out.reserve(ob)); dout = (double *)p->st_out.ptr; }
    }
    MeanArgs a;
    a.e1 = p->e1; a.e2 = p->e2; a.e3 = p->e3; a.e31d = p->e31d; a.mask = p->mask; a.zv = (const float *)dz; a.ib = p->ib;
    // (the byte form from two basins on: with one, the r4 value is a single register per cell anyway; CDFT_MEAN_IBVALUES: A/B hook)
    a.ibits = (p->ib01 && p->nb >= 2 && !getenv("CDFT_MEAN_IBVALUES")) ? p->ibits : nullptr;
    a.nij = nij; a.npk = p->npk; a.nb = p->nb; a.partial = p->partial; a.form = p->form;
    {
        const char *e = getenv("CDFT_MEAN_R4MASK");              // test hook: the r4 mask
        a.mask8 = (p->mask01 && !(e && e[0] == '1')) ? p->mask8 : nullptr;
    }
    const bool vec4 = nij >= 4 && !getenv("CDFT_MEAN_SCALAR");         // (A/B hook: the scalar path)
    const int vec = vec4 ? 4 : 1;
    // vectors per lane: 4 (190 against 210 us at eORCA025, 1.72 against 1.73 ms at eORCA12); CDFT_MEAN_VPT = 2 | 4: A/B hook
    int vpt = (vec4 && (p->nb <= 2 || a.ibits)) ? 4 : kVPT;     // (the r4 basin VALUES of three basins and more: 16 x nb registers with four)
    { const char *e = getenv("CDFT_MEAN_VPT"); if (e && atoi(e) == 2) vpt = 2; else if (e && atoi(e) == 4 && vec4) vpt = 4; }
    a.nblk = (int)(((nij + vec - 1) / vec + kThreads * vpt - 1) / (kThreads * vpt));
    a.band = xcd_bands_on();
    const dim3 grid(xcd_band_grid(a.nblk, a.band)), block(kThreads);
    const size_t park_bytes = (size_t)(kThreads / 64) * p->npk * nq * sizeof(double);
    // (the parked sums AND the wave-sum scratch must fit the 64 KB a launch gets without asking for more: npk * nb = 384 --
    // 4 basins x 96 levels -- parks 48 KB and would ask for 65 792 bytes with the scratch)
    const size_t red_bytes = p->nb >= 2 ? (size_t)(kThreads / 64) * 8 * kSum8Stride * sizeof(double) : 0;
    const bool park = park_bytes <= 48 * 1024 && park_bytes + red_bytes <= 64 * 1024 && !getenv("CDFT_MEAN_NOPARK");     // (A/B hook: a barrier pair per level)
    const size_t wsum_bytes = park ? park_bytes : (size_t)(kThreads / 64) * nq * sizeof(double);
    a.redoff = (int)(wsum_bytes / sizeof(double));
    const size_t shmem = wsum_bytes + red_bytes;
    const bool full = p->e31d != nullptr;
#define CDFT_MEAN4(NBC, PK, M8, IB) do { \
        if (full && vpt == 4) hipLaunchKernelGGL((mean_level_kernel<4, true, NBC, PK, 4, M8, IB>), grid, block, shmem, st, a); \
        else if (vpt == 4)    hipLaunchKernelGGL((mean_level_kernel<4, false, NBC, PK, 4, M8, IB>), grid, block, shmem, st, a); \
        else if (full)        hipLaunchKernelGGL((mean_level_kernel<4, true, NBC, PK, kVPT, M8, IB>), grid, block, shmem, st, a); \
        else                  hipLaunchKernelGGL((mean_level_kernel<4, false, NBC, PK, kVPT, M8, IB>), grid, block, shmem, st, a); } while (0)
#define CDFT_MEAN3(NBC, PK, M8) do { \
        if (NBC >= 2 && a.ibits) CDFT_MEAN4(NBC, PK, M8, (NBC >= 2)); \
        else                     CDFT_MEAN4(NBC, PK, M8, false); } while (0)
#define CDFT_MEAN2(NBC, PK) do { \
        if (vec4 && a.mask8)    CDFT_MEAN3(NBC, PK, true); \
        else if (vec4)          CDFT_MEAN3(NBC, PK, false); \
        else if (full)          hipLaunchKernelGGL((mean_level_kernel<1, true, NBC, PK, kVPT, false, false>), grid, block, shmem, st, a); \
        else                    hipLaunchKernelGGL((mean_level_kernel<1, false, NBC, PK, kVPT, false, false>), grid, block, shmem, st, a); } while (0)
#define CDFT_MEAN(NBC) do { if (park) CDFT_MEAN2(NBC, true); else CDFT_MEAN2(NBC, false); } while (0)
    if (p->nb == 1)      CDFT_MEAN(1);
    else if (p->nb == 2) CDFT_MEAN(2);
    else if (p->nb <= 4) CDFT_MEAN(4);
    else if (p->nb == 5) CDFT_MEAN(5);                           // (the five sub-basins of a NEMO basin file)
    else                 CDFT_MEAN(8);
#undef CDFT_MEAN2
#undef CDFT_MEAN3
#undef CDFT_MEAN4
#undef CDFT_MEAN
    CDFT_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_final_kernel, dim3(p->npk), dim3(kThreads), 0, st, p->partial, a.nblk, nq, dout);
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        if (mapped) {
            CDFT_HIP(hipStreamSynchronize(st));
            memcpy(sums, p->hsums, ob);
        } else {
            CDFT_HIP(hipMemcpyAsync(sums, dout, ob, hipMemcpyDeviceToHost, st));
            CDFT_HIP(hipStreamSynchronize(st));
        }
    }
    return CDFT_OK;
}

int cdft_mean_destroy(cdft_mean_plan *p) {
    if (!p) return CDFT_OK;
    void *ptrs[] = {p->e1, p->e2, p->e3, p->e31d, p->mask, p->mask8, p->ib, p->ibits, p->partial};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (p->hsums) (void)hipHostFree(p->hsums);
    p->st_zv.release(); p->st_out.release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"
