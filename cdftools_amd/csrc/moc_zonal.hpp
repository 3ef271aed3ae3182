// moc_zonal.hpp -- the cdfmoc zonal kernel (gfx950), included once per basin capacity NB by
// moc_zonal_nb{1,5,8}.hip so that the variants build in parallel.
//
// Replaces the reference loop src/cdfmoc.f90:355-379
//     dmoc(jbasin,jj,jk) = dmoc(jbasin,jj,jk) - e1v(ji,jj)*e3v(ji,jj,jk)*ibmask(jbasin,ji,jj)*zv(ji,jj)*1.d0
// Arithmetic contract (kept bit-faithful per term): the product is formed in r4, left to right,
// then widened to r8 and accumulated in r8.  With ibmask in {0,1} the r4 product equals
// (e1v*e3v)*zv where the basin bit is set and contributes (+/-)0 elsewhere (NaN when zv is NaN or
// Inf there: 0*NaN, kept), so the kernel keeps ONE r4 weight w = e1v*e3v per cell (rounded once,
// as the reference's first product) plus one byte of basin bits per (i,j), and selects per basin.
// Only the order of the r8 additions along i differs from the sequential reference (fixed tree =>
// run-to-run deterministic).
//
// Work decomposition (HBM-bound, no MFMA):
//   one workgroup            = one (jj,jk) row of npiglo cells, for ALL npt records
//   one wave (64 lanes)      = a contiguous segment of 64*NV float4 vectors of that row
//   weights + basin bits     = loaded once into registers, every record streams past them
//                              => 4*(1+1/npt) algorithmic bytes per cell-record
//   velocity                 = NV x 16-byte coalesced loads per lane per record, the next
//                              record is in flight while the current one is reduced
//   per-record reduction     = per-lane r8 partial sums -> one LDS transposition per wave
//                              (8x8 lanes per basin) -> 3 cross-lane steps; waves of a row are
//                              combined through LDS in wave order once per record chunk
#pragma once
#include <type_traits>
#include "common.hpp"

namespace cdft {
namespace moc {

constexpr int kWave      = 64;
constexpr int kMaxNV     = 16;   // float4 vectors per lane held in registers
constexpr int kMaxWaves  = 16;   // waves per row (1024-thread workgroup)
constexpr int kSmallWaves = 4;   // waves per row of a 256-thread workgroup
constexpr int kSegWaves  = 2;    // class kernel, rows longer than a 256-thread workgroup covers: waves per row segment
constexpr int kMidWaves  = 8;    // waves per row of a 512-thread workgroup (class kernel, 6..9 vectors per lane)
constexpr int kBigNV     = 6;    // vectors per lane a 1024-thread workgroup can hold unspilled
constexpr int kBigNVClass = 5;   // the same for the class kernel (one slot index per cell on top)
constexpr int kTChunk    = 16;   // records combined per LDS flush
constexpr int kScrStride = 72;   // doubles; 64 + 8 keeps the 5 basin rows on different banks

enum Mode { kModeSelect = 1, kModeFma = 2, kModeClass = 3 };
constexpr int kMaxClass = 8;     // distinct basin-bit patterns (the empty one included) the class kernel can hold

struct ZonalArgs {
    const float   *w;           // (npi,npj,npk) r4 weights, base 16-byte aligned, padded by 16 B
    const uint8_t *bits;        // (npi,npj) basin bits
    const uint8_t *cls;         // (npi,npj) class of the cell's bit pattern (class mode)
    uint8_t        pat[kMaxClass];  // bit pattern of each class
    int            nclass;      // patterns in use; 0 = class mode unavailable (more than kMaxClass)
    const float   *zv;          // record 0
    int64_t        rec_stride;  // elements between records
    int64_t        w_row;       // elements between consecutive (jj,jk) rows of w   (npi: the reference's dense arrays)
    int64_t        z_row;       // the same for every record of zv (aligned kernels: w_row and z_row equal modulo 4)
    int            npt;
    double        *out;         // (nbasins,npj,npk,npt): -(zonal sum), levels 1..nlev (of the window)
    int            npi, npj, npk, nbasins;
    int            nlev;        // levels this launch covers: rows 0 .. npj*nlev-1 of w / zv; npk-1 for a whole domain.  A launch on a
                                // window of levels k0 .. k0+nlev-1 passes w / zv of the window and out + k0*npj*nbasins.
    int            rows_per_wg; // class kernel: consecutive rows one workgroup works through (>= 1)
    int            xcd_map;     // class kernel: workgroup -> row so that each XCD takes an eighth of the rows of every level
    int            aligned;     // class kernel: record base and stride are multiples of 16 bytes
    // class kernel, long rows: a row is cut into nseg segments of nwaves*64*NV vectors, one workgroup each (segments of a
    // row are consecutive workgroups); segment s writes its sums to seg_out + s*seg_stride (laid out like out), and
    // moc_seg_sum_kernel adds the segments in order into out.  nseg = 1: straight into out.
    int            nseg;
    double        *seg_out;
    int64_t        seg_stride;
};

typedef int (*ZonalLauncher)(const ZonalArgs &, int nwaves, int nv, bool aligned, int mode,
                             hipStream_t st);
// defined by moc_zonal_nb{1,5,8}.hip
int launch_zonal_nb1(const ZonalArgs &, int, int, bool, int, hipStream_t);
int launch_zonal_nb5(const ZonalArgs &, int, int, bool, int, hipStream_t);
int launch_zonal_nb8(const ZonalArgs &, int, int, bool, int, hipStream_t);
// defined by moc_zonal_class.hip (kMaxClass accumulators, 16-byte aligned records only)
int launch_zonal_class(const ZonalArgs &, int nwaves, int nv, hipStream_t st);

#ifdef CDFT_MOC_NB

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    return __shfl_xor(v, mask, kWave);
}

template <int NV, int NB, bool ALIGNED, int MODE, int LB>
__global__ void __launch_bounds__(LB)
moc_zonal_kernel(ZonalArgs a) {
    extern __shared__ double smem[];
    const int lane   = threadIdx.x & (kWave - 1);
    const int wave   = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    double *scr  = smem + wave * (NB * kScrStride);              // per-wave transposition scratch
    double *part = smem + nwaves * (NB * kScrStride);            // [kTChunk][NB][nwaves]

    const int row = blockIdx.x;                                   // jk*npj + jj, jj fastest
    const int jk  = row / a.npj;
    const int jj  = row - jk * a.npj;
    const int64_t g0w  = (int64_t)row * a.w_row;                  // first cell of the row in the weights
    const int64_t g0   = (int64_t)row * a.z_row;                  // ... and in a record
    const int     head = ALIGNED ? (int)(g0 & 3) : 0;             // cells before the row in its first float4 (the same for w)
    const int64_t a0   = g0 - head, a0w = g0w - head;
    const int     v0   = wave * (kWave * NV) + lane;              // this lane's first vector

    float   w[NV][4];
    bool    ok[NV][4];
    uint32_t bits[NV];                                            // 4 basin-bit bytes per vector
    bool    vec_ok[NV];

    // Branch-free prologue: every load uses an in-range address (clamped) and is masked
    // afterwards, so the weight / basin-bit loads of a row are all in flight together.
    const uint8_t *brow = a.bits + (int64_t)jj * a.npi;
    float4   wraw[NV];
    uint32_t braw[NV][4];
#pragma unroll
    for (int r = 0; r < NV; ++r) {                                // issue every load first
        const int v  = v0 + r * kWave;
        const int i0 = 4 * v - head;                              // column of component 0
        vec_ok[r] = i0 < a.npi;                                   // (i0+3 >= 0 always)
        const int vc = vec_ok[r] ? v : 0;                         // vector 0 of the row is always valid
        if (ALIGNED) {
            const f4a q = ld_stream4(a.w + a0w + 4 * (int64_t)vc);
            wraw[r] = make_float4(q.x, q.y, q.z, q.w);
        } else {                                                  // head = 0 here, so i0 >= 0
            wraw[r].x = a.w[g0w + min(i0 + 0, a.npi - 1)];
            wraw[r].y = a.w[g0w + min(i0 + 1, a.npi - 1)];
            wraw[r].z = a.w[g0w + min(i0 + 2, a.npi - 1)];
            wraw[r].w = a.w[g0w + min(i0 + 3, a.npi - 1)];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) braw[r][c] = brow[min(max(i0 + c, 0), a.npi - 1)];
    }
    // One record's velocity for this lane.  Lanes past the row end re-read vector 0 of the row
    // (in range, and a cache hit); their weight is 0 and their basin bits are clear.
    auto load_rec = [&](float (&z)[NV][4], int t) {
        const float *zr = a.zv + (int64_t)t * a.rec_stride;
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int v  = v0 + r * kWave;
            const int vc = vec_ok[r] ? v : 0;
            if (ALIGNED) {
                const f4a q = ld_stream4(zr + a0 + 4 * (int64_t)vc);
                z[r][0] = q.x; z[r][1] = q.y; z[r][2] = q.z; z[r][3] = q.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) z[r][c] = zr[g0 + min(4 * v + c, a.npi - 1)];
            }
        }
    };

    float zc[NV][4], zn[NV][4];
    load_rec(zc, 0);                                              // first record in flight behind the weights

#pragma unroll
    for (int r = 0; r < NV; ++r) {                                // then mask what lies outside the row
        const int i0 = 4 * (v0 + r * kWave) - head;
        uint32_t bb = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + c;
            ok[r][c] = (i >= 0) && (i < a.npi);
            bb |= (ok[r][c] ? braw[r][c] : 0u) << (8 * c);
        }
        bits[r] = bb;
        w[r][0] = ok[r][0] ? wraw[r].x : 0.f;
        w[r][1] = ok[r][1] ? wraw[r].y : 0.f;
        w[r][2] = ok[r][2] ? wraw[r].z : 0.f;
        w[r][3] = ok[r][3] ? wraw[r].w : 0.f;
    }

    // MODE fma: basin masks as r8 0/1 multipliers held in registers (small NV only).
    double m[MODE == kModeFma ? NV : 1][4][NB];
    if (MODE == kModeFma) {
#pragma unroll
        for (int r = 0; r < NV; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    m[r][c][b] = ((bits[r] >> (8 * c + b)) & 1u) ? 1.0 : 0.0;
    }


    for (int t0 = 0; t0 < a.npt; t0 += kTChunk) {
        const int tcount = min(kTChunk, a.npt - t0);
        for (int tl = 0; tl < tcount; ++tl) {
            const int t = t0 + tl;
            if (t + 1 < a.npt) load_rec(zn, t + 1);               // next record in flight

            // ibmask = 0: the reference's term is ((e1v*e3v)*0)*zv, a signed zero that never changes the
            // r8 sum -- unless zv is NaN/Inf there (a fill value on land), when it is NaN and the sum must
            // keep it.  The usual all-finite record selects pd or 0 (or multiplies by the 0/1 mask); a record
            // in which any r4 product of the wave is not finite -- a NaN/Inf velocity, or a finite one large
            // enough to overflow the r4 product -- is accumulated again with the reference's own zero term
            // (w*0)*zv in place of the 0 (EXACT): NaN for a NaN/Inf velocity, 0 for an overflow.
            double acc[NB];
            float nonfinite = 0.f;                                // sum of 0*pf: 0, or NaN once a product is NaN/Inf
            auto accumulate = [&](auto exact) {
                constexpr bool EXACT = decltype(exact)::value;
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] = 0.0;
#pragma unroll
                for (int r = 0; r < NV; ++r) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        // r4 product, then widen: src/cdfmoc.f90:373-374.  Lanes past the row end re-read
                        // a real cell with weight 0: harmless unless that cell is not finite
                        const float zf = (MODE == kModeFma || EXACT) ? (ok[r][c] ? zc[r][c] : 0.f) : zc[r][c];
                        const float pf = __fmul_rn(w[r][c], zf);
                        const double pd = (double)pf;
                        if (!EXACT) nonfinite = __fmaf_rn(pf, 0.f, nonfinite);
                        const double pz = EXACT ? (double)__fmul_rn(__fmul_rn(w[r][c], 0.f), zf) : 0.0;
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            if (MODE == kModeFma && !EXACT)
                                acc[b] = __fma_rn(pd, m[r][c][b], acc[b]);
                            else
                                acc[b] += ((bits[r] >> (8 * c + b)) & 1u) ? pd : pz;
                        }
                    }
                }
            };
            accumulate(std::false_type{});
            if (__builtin_amdgcn_ballot_w64(nonfinite != 0.f) != 0) accumulate(std::true_type{});

            // wave reduction: transpose through LDS so that lane (b*8+s) sums 8 lanes' partials
#pragma unroll
            for (int b = 0; b < NB; ++b) scr[b * kScrStride + lane] = acc[b];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double s = 0.0;
            if (lane < NB * 8) {
                const double *p = scr + (lane >> 3) * kScrStride + (lane & 7);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += p[8 * j];
            }
            s += shfl_xor_f64(s, 1);
            s += shfl_xor_f64(s, 2);
            s += shfl_xor_f64(s, 4);
            __builtin_amdgcn_wave_barrier();                      // scr is rewritten next record
            if (lane < NB * 8 && (lane & 7) == 0) {
                const int b = lane >> 3;
                if (nwaves == 1) {
                    if (b < a.nbasins)
                        a.out[(((int64_t)t * a.npk + jk) * a.npj + jj) * a.nbasins + b] = -s;
                } else {
                    part[(tl * NB + b) * nwaves + wave] = s;
                }
            }
#pragma unroll
            for (int r = 0; r < NV; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) zc[r][c] = zn[r][c];
        }
        if (nwaves > 1) {
            __syncthreads();
            if ((int)threadIdx.x < tcount * NB) {
                const int tl = threadIdx.x / NB, b = threadIdx.x - tl * NB;
                double s = 0.0;
                for (int k = 0; k < nwaves; ++k) s += part[(tl * NB + b) * nwaves + k];
                if (b < a.nbasins)
                    a.out[(((int64_t)(t0 + tl) * a.npk + jk) * a.npj + jj) * a.nbasins + b] = -s;
            }
            __syncthreads();
        }
    }
}


template <int NV, int NB, bool ALIGNED, int MODE, int LB>
static int launch_lb(const ZonalArgs &a, int nwaves, hipStream_t st) {
    const size_t shmem =
        sizeof(double) * ((size_t)nwaves * NB * kScrStride + (size_t)kTChunk * NB * nwaves);
    const unsigned grid = (unsigned)a.npj * (unsigned)a.nlev;
    hipLaunchKernelGGL((moc_zonal_kernel<NV, NB, ALIGNED, MODE, LB>), dim3(grid),
                       dim3(kWave * nwaves), shmem, st, a);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

// Workgroups of up to kSmallWaves waves get the full register budget (512 VGPRs / lane at one
// wave per SIMD); longer rows use up to kMaxWaves waves with at most kBigNV vectors per lane,
// which fits the 128 VGPRs a 1024-thread workgroup leaves per lane.
template <int NV, int NB, bool ALIGNED, int MODE>
static int launch_one(const ZonalArgs &a, int nwaves, hipStream_t st) {
    if (nwaves <= kSmallWaves) return launch_lb<NV, NB, ALIGNED, MODE, kWave * kSmallWaves>(a, nwaves, st);
    if constexpr (NV <= kBigNV && MODE == kModeSelect)
        return launch_lb<NV, NB, ALIGNED, MODE, kWave * kMaxWaves>(a, nwaves, st);
    else
        return fail(CDFT_ERR_ARG, "%d waves per row need <= %d vectors per lane (select mode), got %d",
                    nwaves, kBigNV, NV);
}

template <int NB>
static int launch_nb(const ZonalArgs &a, int nwaves, int nv, bool aligned, int mode,
                     hipStream_t st) {
    if (mode == kModeClass) return fail(CDFT_ERR_ARG, "class mode is launched through launch_zonal_class");
    if (!aligned) {
        if (nv != 4 || mode != kModeSelect)
            return fail(CDFT_ERR_ARG, "unaligned zonal path supports nv=4, select mode only");
        return launch_one<4, NB, false, kModeSelect>(a, nwaves, st);
    }
    if (mode == kModeFma) {
        switch (nv) {
            case 1: return launch_one<1, NB, true, kModeFma>(a, nwaves, st);
            case 2: return launch_one<2, NB, true, kModeFma>(a, nwaves, st);
            case 3: return launch_one<3, NB, true, kModeFma>(a, nwaves, st);
            case 4: return launch_one<4, NB, true, kModeFma>(a, nwaves, st);
            default: return fail(CDFT_ERR_ARG, "fma mode supports 1..4 vectors per lane, got %d", nv);
        }
    }
    switch (nv) {
#define CDFT_CASE(N) case N: return launch_one<N, NB, true, kModeSelect>(a, nwaves, st);
        CDFT_CASE(1) CDFT_CASE(2) CDFT_CASE(3) CDFT_CASE(4) CDFT_CASE(5) CDFT_CASE(6)
        CDFT_CASE(7) CDFT_CASE(8) CDFT_CASE(9) CDFT_CASE(10) CDFT_CASE(11) CDFT_CASE(12)
        CDFT_CASE(13) CDFT_CASE(14) CDFT_CASE(15) CDFT_CASE(16)
#undef CDFT_CASE
        default: return fail(CDFT_ERR_ARG, "vectors per lane must be 1..16, got %d", nv);
    }
}

#endif  // CDFT_MOC_NB

#ifdef CDFT_MOC_CLASS
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    return __shfl_xor(v, mask, kWave);
}

// Class mode.  The select kernel above spends NB select-adds in r8 per cell; at 5 basins that is
// as expensive as the memory stream itself.  A cell's basin membership is one of a handful of bit
// patterns (open ocean only / Atlantic / Indian / Pacific ...), so each cell is given the CLASS of
// its pattern once (cdft_moc_set_mesh) and its r8 term is added to exactly one accumulator: a
// per-lane, per-class slot in LDS, updated by ds_add_f64 (one LDS instruction per cell instead of
// ~5 x 5 VALU instructions).  A slot only ever receives the terms of its own lane, in program
// order, so the result is run-to-run deterministic.  Basin sums are formed from the class sums at
// the end (basin b = classes whose pattern has bit b, in class order).  The empty pattern (a cell in
// no basin) is a class like the others.  The zero terms ((e1v*e3v)*0)*zv the reference adds for the cells outside a
// basin matter only when a velocity there is NaN/Inf: as long as the sum of those classes is finite nothing is added;
// when it is not, the row's velocities are read again to tell a NaN/Inf (the term is NaN) from a finite velocity whose
// r4 product overflowed (the term is 0) -- a slow path no sane record takes.
// ALIGNED: records whose base and stride are multiples of 16 bytes: the row is covered by vectors aligned to the
// ARRAY and its first / last vector is shared with the neighbouring rows (masked).  Otherwise (eORCA025: 1442 x 1207 x 75
// cells per record, not a multiple of 4): 16-byte loads at r4 alignment, vectors at cells 0, 4, 8, ... of the ROW plus
// one anchored at npiglo-4 whose leading cells repeat the previous vector's and go to the unread slot.
template <bool ALIGNED> struct ZVec;
template <> struct ZVec<true>  { typedef f4a T; static __device__ __forceinline__ f4a ld(const float *p) { return ld_stream4(p); } };
template <> struct ZVec<false> { typedef f4u T; static __device__ __forceinline__ f4u ld(const float *p) { return ld_stream4u(p); } };

template <int NV, int KC, bool ALIGNED>
__device__ __forceinline__ void moc_zonal_class_row(const ZonalArgs &a, const int row, const int seg, double *smem) {
    typedef typename ZVec<ALIGNED>::T V4;
    const int lane   = threadIdx.x & (kWave - 1);
    const int wave   = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    double *scr  = smem + wave * ((KC + 1) * kScrStride);        // slot [class][lane]; class KC = lanes past the row end
    double *part = smem + nwaves * ((KC + 1) * kScrStride);      // [kTChunk][KC][nwaves]
    // [0]: records of the chunk whose zero term needs the velocities again; [1+tl]: classes holding a NaN/Inf velocity
    uint32_t *flag = reinterpret_cast<uint32_t *>(part + kTChunk * KC * nwaves);

    const int jk  = row / a.npj;                                  // row = jk*npj + jj, jj fastest
    const int jj  = row - jk * a.npj;
    const int64_t g0w  = (int64_t)row * a.w_row;                  // the row in the weights
    const int64_t g0   = (int64_t)row * a.z_row;                  // ... and in a record (ALIGNED: the same modulo 4)
    const int     head = ALIGNED ? (int)(g0 & 3) : 0;
    const int     segv = nwaves * (kWave * NV);                   // vectors of a segment
    const int     v0   = seg * segv + wave * (kWave * NV) + lane;
    const int     nfull = a.npi >> 2;                             // (not ALIGNED) vectors 0..nfull-1 start at cell 4v
    double *const outp = a.nseg > 1 ? a.seg_out + (int64_t)seg * a.seg_stride : a.out;

    float    w[NV][4];
    uint32_t cpk[NV];                                             // the 4 cells' classes, a byte each (their slots: cls*kScrStride + lane)
    // column a lane's vector r is loaded from (ALIGNED: -head ..): formed again at every use (two VALU operations) rather
    // than kept in NV registers -- with the packed classes that brings 9 vectors per lane under 168 VGPRs, three waves a SIMD
    auto ioff_of = [&](int r, int base) -> int {
        const int i0 = base + 4 * kWave * r;
        return ALIGNED ? (i0 < a.npi ? i0 : -head) : min(i0, a.npi - 4);
    };
    const int ibase = 4 * v0 - (ALIGNED ? head : 0);
    const uint8_t *crow = a.cls + (int64_t)jj * a.npi;
    V4       wraw[NV];
    typedef uint32_t u32u __attribute__((aligned(1)));
    uint32_t craw[NV];                                            // the classes of columns cfirst .. cfirst+3: ONE unaligned dword per vector
#pragma unroll
    for (int r = 0; r < NV; ++r) {                                // issue every load first
        const int v  = v0 + r * kWave;
        int i0;                                                   // column of component 0
        if (ALIGNED) {
            i0 = 4 * v - head;
            // (lanes past the row end re-read its first vector)
        } else {
            i0 = min(4 * v, a.npi - 4);                           // v >= nfull: the anchored vector (in range for every lane)
        }
        wraw[r] = ZVec<ALIGNED>::ld(a.w + g0w + ioff_of(r, ibase));
        craw[r] = *reinterpret_cast<const u32u *>(crow + max(0, min(i0, a.npi - 4)));      // (the array is padded past its last row)
    }
    auto load_rec = [&](V4 (&z)[NV], int t) {
        const float *zr = a.zv + (int64_t)t * a.rec_stride + g0;
        int base = ibase;
        asm volatile("" : "+v"(base));                            // (or the NV offsets are hoisted into registers again)
#pragma unroll
        for (int r = 0; r < NV; ++r) z[r] = ZVec<ALIGNED>::ld(zr + ioff_of(r, base));
    };
    V4 zc[NV], zn[NV];
    load_rec(zc, 0);
#pragma unroll
    for (int s = 0; s <= KC; ++s) scr[s * kScrStride + lane] = 0.0;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int v = v0 + r * kWave;
        const float wr[4] = {wraw[r].x, wraw[r].y, wraw[r].z, wraw[r].w};
        const int ifirst = ALIGNED ? 4 * v - head : min(4 * v, a.npi - 4);
        const int cfirst = max(0, min(ifirst, a.npi - 4));        // the column the class dword was read at
        cpk[r] = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = ifirst + c;
            const bool ok = ALIGNED ? (i >= 0 && i < a.npi)
                                    : (v < nfull || (v == nfull && i >= 4 * nfull));
            w[r][c] = ok ? wr[c] : 0.f;
            const uint32_t cl = (craw[r] >> ((8 * (i - cfirst)) & 31)) & 0xffu;     // (an owned cell lies in cfirst .. cfirst+3)
            cpk[r] |= (ok ? min(cl, (uint32_t)KC) : (uint32_t)KC) << (8 * c);       // slot KC is never read
        }
    }

    for (int t0 = 0; t0 < a.npt; t0 += kTChunk) {
        const int tcount = min(kTChunk, a.npt - t0);
        if (threadIdx.x <= kTChunk) flag[threadIdx.x] = 0u;       // (ordered before its use by the barrier after the records)
        for (int tl = 0; tl < tcount; ++tl) {
            const int t = t0 + tl;
            if (t + 1 < a.npt) load_rec(zn, t + 1);               // next record in flight
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const float zr[4] = {zc[r].x, zc[r].y, zc[r].z, zc[r].w};
                uint32_t pk = cpk[r];
                asm volatile("" : "+v"(pk));                       // (keeps the 36 slot addresses from being hoisted into registers)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // r4 product, then widen: src/cdfmoc.f90:373-374
                    const double pd = (double)__fmul_rn(w[r][c], zr[c]);
                    const uint32_t cl = (pk >> (8 * c)) & 0xffu;
                    (void)__hip_atomic_fetch_add(&scr[cl * kScrStride + lane], pd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double s = 0.0;
            if (lane < KC * 8) {
                const double *p = scr + (lane >> 3) * kScrStride + (lane & 7);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += p[8 * j];
            }
            s += shfl_xor_f64(s, 1);
            s += shfl_xor_f64(s, 2);
            s += shfl_xor_f64(s, 4);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                      // every lane has read before the slots are cleared
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < KC; ++c) scr[c * kScrStride + lane] = 0.0;
            if (lane < KC * 8 && (lane & 7) == 0) part[(tl * KC + (lane >> 3)) * nwaves + wave] = s;
#pragma unroll
            for (int r = 0; r < NV; ++r) zc[r] = zn[r];
        }
        __syncthreads();
        // basin b of record tl: s = the classes whose pattern has bit b, z = the others, nm = which those are
        auto basin_sums = [&](int tl, int b, double &s, double &z, uint32_t &nm) {
            s = 0.0; z = 0.0; nm = 0u;
            for (int c = 0; c < a.nclass; ++c) {
                const bool in = (a.pat[c] >> b) & 1;
                if (!in) nm |= 1u << c;
                for (int k = 0; k < nwaves; ++k) {
                    const double v = part[(tl * KC + c) * nwaves + k];
                    if (in) s += v;
                    else    z += v;
                }
            }
        };
        for (int q = threadIdx.x; q < tcount * a.nbasins; q += blockDim.x) {     // (one wave holds 64 < 16 records x 5 basins)
            const int tl = q / a.nbasins, b = q - tl * a.nbasins;
            double s, z; uint32_t nm;
            basin_sums(tl, b, s, z, nm);
            // cells outside basin b: the reference adds ((e1v*e3v)*0)*zv, a zero unless zv is NaN/Inf.  While the sum
            // of their products is finite every one of them is: the zero term is a zero.
            if (__builtin_isfinite(z)) {
                s += __dmul_rn(z, 0.0);
                outp[(((int64_t)(t0 + tl) * a.npk + jk) * a.npj + jj) * a.nbasins + b] = -s;
            } else {
                atomicOr(&flag[0], 1u << tl);
            }
        }
        __syncthreads();
        if (flag[0] != 0u) {                                      // (the same word for every lane of the workgroup)
            // A product outside some basin is not finite: a NaN/Inf velocity (the reference's zero term is NaN) or a
            // finite one that overflowed the r4 product (the zero term is 0).  Only the velocities tell: read the
            // row of those records again and note the classes that hold a NaN/Inf.
            const uint32_t need = flag[0];
            // the cells of this segment: those of its vectors (cell i lies in vector (i + head) >> 2; not ALIGNED: head = 0
            // and the anchored vector nfull owns the cells from 4*nfull on)
            const int ifirst = max(0, 4 * seg * segv - head), ilast = (int)min((int64_t)a.npi, (int64_t)4 * (seg + 1) * segv - head);
            for (int tl = 0; tl < tcount; ++tl) {
                if (!((need >> tl) & 1u)) continue;
                const float *zr = a.zv + (int64_t)(t0 + tl) * a.rec_stride + g0;
                for (int i = ifirst + threadIdx.x; i < ilast; i += blockDim.x)
                    if (!__builtin_isfinite(zr[i])) atomicOr(&flag[1 + tl], 1u << crow[i]);
            }
            __syncthreads();
            for (int q = threadIdx.x; q < tcount * a.nbasins; q += blockDim.x) {
                const int tl = q / a.nbasins, b = q - tl * a.nbasins;
                double s, z; uint32_t nm;
                basin_sums(tl, b, s, z, nm);
                if (__builtin_isfinite(z)) continue;             // written above
                if (flag[1 + tl] & nm) s += __builtin_nan("");
                outp[(((int64_t)(t0 + tl) * a.npk + jk) * a.npj + jj) * a.nbasins + b] = -s;
            }
        }
        __syncthreads();
    }
}

// A workgroup takes rows_per_wg consecutive rows (1 in practice: more was measured and brought nothing).
template <int NV, int KC, int LB, bool ALIGNED>
__global__ void __launch_bounds__(LB)
moc_zonal_class_kernel(ZonalArgs a) {
    extern __shared__ double smem[];
    const int nitems = a.npj * a.nlev * a.nseg;              // item = row * nseg + segment
    int wg = blockIdx.x;
    if (a.xcd_map) {
        // Workgroups go round the 8 XCDs in launch order (workgroup b runs on XCD b % 8).  Mapped like this, XCD x takes the
        // x-th eighth of the rows jj of every level, in row order: two consecutive rows share a 128-byte line (a row of 4322
        // cells is 135.06 lines; 13 such lines per row and launch), and it is now asked for twice from the SAME L2, a few
        // microseconds apart, instead of from two.  Measured in one process on one set of records (tools/ab_env.py):
        // 9.86 -> 9.52 ms at eORCA12, the same bits.  (Tiles of rows x all levels per XCD, so that a row's class bytes -- the
        // same for its 74 levels -- would stay in that L2 too, were slower the smaller the tile: 9.59 ms at 128 rows, 9.80
        // at 4; the jumps between levels cost more than the class bytes.)
        const int x = wg & 7, s = wg >> 3;
        const int j0 = (int)((int64_t)x * a.npj / 8), nrows = (int)((int64_t)(x + 1) * a.npj / 8) - j0;
        if (nrows <= 0) return;
        const int lev = s / nrows;
        if (lev >= a.nlev) return;
        wg = lev * a.npj + j0 + (s - lev * nrows);
    }
    const int r0 = wg * a.rows_per_wg, r1 = min(nitems, r0 + a.rows_per_wg);
    for (int item = r0; item < r1; ++item) {
        const int row = item / a.nseg;
        moc_zonal_class_row<NV, KC, ALIGNED>(a, row, item - row * a.nseg, smem);
    }
}

// out = the segments' sums added in segment order (levels 1..npk-1 of every record; level npk is not theirs)
__global__ void moc_seg_sum_kernel(ZonalArgs a) {
    const int64_t per_level = (int64_t)a.npj * a.nbasins, per_rec = per_level * a.npk;
    const int64_t n = (int64_t)a.npt * a.nlev * per_level;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = q / (a.nlev * per_level), e = t * per_rec + (q - t * a.nlev * per_level);
        double s = a.seg_out[e];
        for (int g = 1; g < a.nseg; ++g) s += a.seg_out[(int64_t)g * a.seg_stride + e];
        a.out[e] = s;
    }
}

template <int NV, int KC, int LB>
static int launch_class_lb(const ZonalArgs &a, int nwaves, hipStream_t st) {
    const size_t shmem = sizeof(double) * ((size_t)nwaves * (KC + 1) * kScrStride + (size_t)kTChunk * KC * nwaves + (kTChunk + 2) / 2 + 1);
    const unsigned nrows = (unsigned)a.npj * (unsigned)a.nlev * (unsigned)a.nseg;
    unsigned grid = (nrows + (unsigned)a.rows_per_wg - 1) / (unsigned)a.rows_per_wg;
    if (a.xcd_map) grid = 8u * (unsigned)((a.npj + 7) / 8) * (unsigned)a.nlev;      // (an XCD's rows x levels; a few workgroups find nothing to do)
    if (a.aligned) hipLaunchKernelGGL((moc_zonal_class_kernel<NV, KC, LB, true>), dim3(grid), dim3(kWave * nwaves), shmem, st, a);
    else           hipLaunchKernelGGL((moc_zonal_class_kernel<NV, KC, LB, false>), dim3(grid), dim3(kWave * nwaves), shmem, st, a);
    CDFT_LAUNCH_CHECK();
    if (a.nseg > 1) {
        hipLaunchKernelGGL(moc_seg_sum_kernel, dim3(2048), dim3(256), 0, st, a);
        CDFT_LAUNCH_CHECK();
    }
    return CDFT_OK;
}

template <int NV, int KC>
static int launch_class_one(const ZonalArgs &a, int nwaves, hipStream_t st) {
    if (nwaves <= kSmallWaves) return launch_class_lb<NV, KC, kWave * kSmallWaves>(a, nwaves, st);
    // 5..8 waves (rows of 4.6k .. 18k cells, eORCA36): 512-thread workgroups keep 256 VGPRs per lane, i.e. the 9
    // vectors per lane that stream best; 1024-thread workgroups only have room for kBigNVClass
    if constexpr (NV > kBigNVClass && NV <= 9) {
        if (nwaves <= kMidWaves) return launch_class_lb<NV, KC, kWave * kMidWaves>(a, nwaves, st);
    }
    if constexpr (NV <= kBigNVClass) return launch_class_lb<NV, KC, kWave * kMaxWaves>(a, nwaves, st);
    else return fail(CDFT_ERR_ARG, "%d waves per row need <= %d vectors per lane, got %d", nwaves, kBigNVClass, NV);
}

template <int KC>
static int launch_class(const ZonalArgs &a, int nwaves, int nv, hipStream_t st) {
    if (a.nclass < 1 || a.nclass > KC)
        return fail(CDFT_ERR_ARG, "class mode holds 1..%d basin patterns, the masks have %d", KC, a.nclass);
    switch (nv) {
#define CDFT_CASE(N) case N: return launch_class_one<N, KC>(a, nwaves, st);
        CDFT_CASE(1) CDFT_CASE(2) CDFT_CASE(3) CDFT_CASE(4) CDFT_CASE(5) CDFT_CASE(6)
        CDFT_CASE(7) CDFT_CASE(8) CDFT_CASE(9) CDFT_CASE(10) CDFT_CASE(11) CDFT_CASE(12)
#undef CDFT_CASE
        default: return fail(CDFT_ERR_ARG, "class mode: vectors per lane must be 1..12, got %d", nv);
    }
}

#endif  // CDFT_MOC_CLASS

}  // namespace moc
}  // namespace cdft
