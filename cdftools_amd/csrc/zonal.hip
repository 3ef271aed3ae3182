// zonal.hip -- cdfzonalsum / cdfzonalmean: zonal (along i) integral or mean of any field, per level, row
// and sub-basin (C ABI section "cdfzonal").
// Reference loops: src/cdfzonalsum.f90:302-310 and src/cdfzonalmean.f90:319-327, per (variable, record, level, basin):
//     sum  form  dtmp = zmask(b,ji,jj)*zmaskvar(ji,jj)*zv(ji,jj)*1.d0          (r4 product, then widened)
//     mean form  dtmp = 1.d0*zmask(b,ji,jj)*zmaskvar(ji,jj)*zv(ji,jj)          (every product in r8)
//     dzo(jj,jk)   += dl_surf(ji,jj)*dtmp                     with dl_surf = 1.d0*e1*e2 (r8)
//     darea(jj,jk) += dl_surf(ji,jj)*zmask(b,ji,jj)*zmaskvar(ji,jj)             (mean form)
//     rzomax / rzomin = MAX / MIN of dtmp (min over dtmp /= 0) stored in r4     (mean form, -max)
// Per-term values are formed exactly as written; the r8 additions along i are regrouped in a fixed order
// (|d| <= 1e-13 * sum|terms|).  One workgroup owns a row jj for ALL levels of the variable: the 2-D
// operands of the row (surface, basin masks) are re-read per level from L2 / Infinity Cache, the field
// and its 3-D mask stream from HBM once (8 bytes per cell and level).  darea does not depend on the field: it is
// summed once per mesh (cdft_zonal_set_mesh) and copied out by every mean-form call.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"
#include "reduce.hpp"

using namespace cdft;

struct cdft_zonal_plan {
    int npi = 0, npj = 0, npk = 0, nb = 0;
    double *surf = nullptr;       // (npi,npj) 1.d0*e1*e2
    float  *mask3 = nullptr;      // (npi,npj,npk) mask of the variable's grid point
    uint8_t *mask8 = nullptr;     // the same as bytes when every value is 0 or 1 (class kernel: 1 byte per cell instead of 4)
    bool mask01 = false;
    float  *bmask = nullptr;      // (nb,npi,npj) basin masks, basin index fastest
    uint8_t *bits = nullptr;      // (npi,npj) basin bits when every basin mask is 0/1 (fast kernel)
    int *flag = nullptr;          // device: 1 when a basin-mask value is neither 0 nor 1
    int *rowflag = nullptr;       // device (npj): rows of the last fast-kernel launch that met a NaN/Inf
    double *area = nullptr;       // device (nb,npk,npj): masked area of every row, level and basin
    uint8_t *cls = nullptr;       // device (npi,npj) + 256: class of the cell's bit pattern, then the lookup table
    int *seen = nullptr;          // device (256): bit patterns present
    uint8_t pat[8] = {0};
    int nclass = 0;               // distinct bit patterns, 0 when the class kernel cannot be used
    bool binary = false;
    bool mesh_set = false;
    DevBuf st_a, st_b, st_zv, st_o[4];
};

namespace {

constexpr int kThreads = 256;
constexpr int kMaxNB   = 8;

__global__ void zonal_surf_kernel(const float *__restrict__ e1, const float *__restrict__ e2, double *__restrict__ surf, int64_t nij) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nij) surf[p] = __dmul_rn(__dmul_rn(1.0, (double)e1[p]), (double)e2[p]);      // dl_surf = 1.d0 * e1 * e2
}

struct ZonalArgs2 {
    const double *surf;
    const float *mask3, *bmask, *zv;
    const uint8_t *mask8;      // or null
    int npi, npj, nlev, nb;
    double *dsum, *darea;      // (npj,nlev,nb)
    float *rmax, *rmin;        // (npj,nlev,nb) or null
};

template <int NB, bool MEAN, bool MINMAX, bool AREA>
__global__ void __launch_bounds__(kThreads) zonal_row_kernel(ZonalArgs2 a, const int *__restrict__ rowflag) {
    extern __shared__ double lds[];                     // [nlev][nwaves][NB][4]: sum, area, max, min
    const int jj = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = kThreads / 64;
    if (rowflag && !rowflag[jj]) return;                // second pass behind the fast kernel: flagged rows only
    const int64_t nij = (int64_t)a.npi * a.npj, r2 = (int64_t)jj * a.npi;
    for (int k = 0; k < a.nlev; ++k) {
        const float *zv = a.zv + (int64_t)k * nij + r2, *mv = a.mask3 + (int64_t)k * nij + r2;
        double s[NB], ar[NB], mx[NB], mn[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { s[b] = 0.0; ar[b] = 0.0; mx[b] = -1.e20f; mn[b] = 1.e20f; }
        for (int i = threadIdx.x; i < a.npi; i += kThreads) {
            const float z = ld_stream(zv + i), m = ld_stream(mv + i);
            const double sf = a.surf[r2 + i];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b < a.nb) {
                    const float zb = a.bmask[(r2 + i) * a.nb + b];
                    double dtmp;
                    if (MEAN) dtmp = __dmul_rn(__dmul_rn(__dmul_rn(1.0, (double)zb), (double)m), (double)z);
                    else      dtmp = (double)__fmul_rn(__fmul_rn(zb, m), z);
                    s[b] += __dmul_rn(sf, dtmp);
                    if (AREA) ar[b] += __dmul_rn(__dmul_rn(sf, (double)zb), (double)m);
                    if (MINMAX) {
                        mx[b] = fmax(mx[b], dtmp);
                        if (dtmp != 0.0) mn[b] = fmin(mn[b], dtmp);
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double ts = wave_sum(s[b]);
            double ta = 0.0, tx = 0.0, tn = 0.0;
            if (AREA) ta = wave_sum(ar[b]);
            if (MINMAX) {
                tx = mx[b]; tn = mn[b];
                for (int o = 32; o > 0; o >>= 1) { tx = fmax(tx, __shfl_xor(tx, o, 64)); tn = fmin(tn, __shfl_xor(tn, o, 64)); }
            }
            if (lane == 0) {
                double *w = lds + (((int64_t)k * nwaves + wave) * NB + b) * 4;
                w[0] = ts; w[1] = ta; w[2] = tx; w[3] = tn;
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < a.nlev * a.nb; t += kThreads) {
        const int k = t / a.nb, b = t - k * a.nb;
        double ts = 0.0, ta = 0.0, tx = -1.e20f, tn = 1.e20f;
        for (int w = 0; w < nwaves; ++w) {
            const double *q = lds + (((int64_t)k * nwaves + w) * NB + b) * 4;
            ts += q[0]; ta += q[1]; tx = fmax(tx, q[2]); tn = fmin(tn, q[3]);
        }
        const int64_t o = ((int64_t)b * a.nlev + k) * a.npj + jj;
        if (AREA) a.darea[o] = ta;
        else      a.dsum[o] = ts;
        if (MINMAX) { a.rmax[o] = (float)tx; a.rmin[o] = (float)tn; }
    }
}

// bits(i,j) = basin membership bits; flag raised when a mask value is neither 0 nor 1
__global__ void zonal_bits_kernel(const float *__restrict__ bmask, uint8_t *__restrict__ bits, int nb, int64_t nij, int *flag) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    uint32_t v = 0;
    bool bad = false;
    for (int b = 0; b < nb; ++b) {
        const float x = bmask[p * nb + b];
        bad |= (x != 0.f && x != 1.f);
        v |= (x != 0.f ? 1u : 0u) << b;
    }
    bits[p] = (uint8_t)v;
    if (bad) atomicOr(flag, 1);
}

// Fast path for 0/1 basin masks (the usual case).  zmask*zmaskvar*zv is then (zmaskvar*zv) or an exact
// zero, and dl_surf*zmask*zmaskvar is (dl_surf*zmaskvar) or zero, so the per-cell work is done once and
// each basin only selects: bit-identical terms to the general kernel.  A workgroup owns a row jj for all
// levels; every thread keeps the surface and the basin bits of its M cells (i = tid, tid+T, ...) in
// registers and streams the field and its mask past them: 8 bytes per cell and level from HBM.
// M8: the level mask as one byte per cell (it holds only 0 and 1: the same m * z products), 5 bytes per cell and level instead of 8
template <int NB, int M, bool MEAN, bool MINMAX, int T, bool M8>
__global__ void __launch_bounds__(T) zonal_row_fast_kernel(ZonalArgs2 a, const uint8_t *__restrict__ bits,
                                                            int *__restrict__ rowflag) {
    extern __shared__ double lds[];                     // [nlev][nwaves][NB][4]
    const int jj = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = T / 64;
    const int64_t nij = (int64_t)a.npi * a.npj, r2 = (int64_t)jj * a.npi;
    double sf[M];
    uint32_t bb[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int i = threadIdx.x + m * T;
        const bool ok = i < a.npi;
        sf[m] = ok ? a.surf[r2 + i] : 0.0;
        bb[m] = ok ? bits[r2 + i] : 0u;
    }
    float zc[M], mc[M], zn[M], mn_[M];
    float nonfinite = 0.f;                              // sum of 0*z over every value this thread reads: 0 or NaN
    auto fetch = [&](float (&z)[M], float (&mk)[M], int k) {
        const float *zv = a.zv + (int64_t)k * nij + r2, *mv = a.mask3 + (int64_t)k * nij + r2;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int i = min((int)threadIdx.x + m * T, a.npi - 1);   // clamped: cells past the row have no bit set
            z[m] = ld_stream(zv + i);
            mk[m] = M8 ? (float)ld_stream(a.mask8 + (int64_t)k * nij + r2 + i) : ld_stream(mv + i);
        }
    };
    fetch(zc, mc, 0);
    for (int k = 0; k < a.nlev; ++k) {
        if (k + 1 < a.nlev) fetch(zn, mn_, k + 1);      // next level in flight
        double s[NB], mx[NB], mi[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { s[b] = 0.0; mx[b] = -1.e20f; mi[b] = 1.e20f; }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            nonfinite = __fmaf_rn(zc[m], 0.f, nonfinite);      // 0*NaN = 0*Inf = NaN sticks (see below)
            // zmask = 1: sum form (1.f*m)*z = m*z in r4; mean form ((1.d0*1)*m)*z = m*z in r8
            const double dt = MEAN ? __dmul_rn((double)mc[m], (double)zc[m]) : (double)__fmul_rn(mc[m], zc[m]);
            const double u = __dmul_rn(sf[m], dt);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bool in = (bb[m] >> b) & 1u;
                // zmask = 0: the reference's term is an exact (signed) zero, which never changes an r8 sum
                // that already holds a value; and its dtmp = 0 never enters the minimum
                s[b] += in ? u : 0.0;
                if (MINMAX) {
                    const double d = in ? dt : 0.0;
                    mx[b] = fmax(mx[b], ((int)threadIdx.x + m * T < a.npi) ? d : mx[b]);
                    if (in && dt != 0.0) mi[b] = fmin(mi[b], dt);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double ts = wave_sum(s[b]);
            double tx = 0.0, tn = 0.0;
            if (MINMAX) {
                tx = mx[b]; tn = mi[b];
                for (int o = 32; o > 0; o >>= 1) { tx = fmax(tx, __shfl_xor(tx, o, 64)); tn = fmin(tn, __shfl_xor(tn, o, 64)); }
            }
            if (lane == 0) {
                double *q = lds + (((int64_t)k * nwaves + wave) * NB + b) * 4;
                q[0] = ts; q[1] = 0.0; q[2] = tx; q[3] = tn;
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) { zc[m] = zn[m]; mc[m] = mn_[m]; }
    }
    // A NaN/Inf in the row (a fill value on land): where zmask = 0 the reference's term is 0*NaN = NaN, not the
    // zero selected above.  The row is flagged and zonal_row_kernel, which multiplies as the reference does,
    // recomputes it.
    const int redo = __syncthreads_or(!__builtin_isfinite(nonfinite));
    if (threadIdx.x == 0) rowflag[jj] = redo;
    for (int t = threadIdx.x; t < a.nlev * a.nb; t += T) {
        const int k = t / a.nb, b = t - k * a.nb;
        double ts = 0.0, tx = -1.e20f, tn = 1.e20f;
        for (int w = 0; w < nwaves; ++w) {
            const double *q = lds + (((int64_t)k * nwaves + w) * NB + b) * 4;
            ts += q[0]; tx = fmax(tx, q[2]); tn = fmin(tn, q[3]);
        }
        const int64_t o = ((int64_t)b * a.nlev + k) * a.npj + jj;
        a.dsum[o] = ts;
        if (MINMAX) { a.rmax[o] = (float)tx; a.rmin[o] = (float)tn; }
    }
}

template <int NB, int M, int T>
int launch_fast(const ZonalArgs2 &a, const uint8_t *bits, int *rowflag, bool mean, bool minmax, hipStream_t st) {
    const size_t shmem = (size_t)a.nlev * (T / 64) * NB * 4 * sizeof(double);
    const dim3 grid(a.npj), block(T);
    if (a.mask8) {
        if (mean && minmax)  hipLaunchKernelGGL((zonal_row_fast_kernel<NB, M, true, true, T, true>), grid, block, shmem, st, a, bits, rowflag);
        else if (mean)       hipLaunchKernelGGL((zonal_row_fast_kernel<NB, M, true, false, T, true>), grid, block, shmem, st, a, bits, rowflag);
        else                 hipLaunchKernelGGL((zonal_row_fast_kernel<NB, M, false, false, T, true>), grid, block, shmem, st, a, bits, rowflag);
    } else {
        if (mean && minmax)  hipLaunchKernelGGL((zonal_row_fast_kernel<NB, M, true, true, T, false>), grid, block, shmem, st, a, bits, rowflag);
        else if (mean)       hipLaunchKernelGGL((zonal_row_fast_kernel<NB, M, true, false, T, false>), grid, block, shmem, st, a, bits, rowflag);
        else                 hipLaunchKernelGGL((zonal_row_fast_kernel<NB, M, false, false, T, false>), grid, block, shmem, st, a, bits, rowflag);
    }
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

// rows up to 1536 / 4352 cells with 256 threads; longer rows would need more registers per thread than
// pay off: they take the general kernel
template <int NB>
int launch_fast_nb(const ZonalArgs2 &a, const uint8_t *bits, int *rowflag, bool mean, bool minmax, hipStream_t st, bool *done) {
    *done = true;
    if (a.npi <= 256 * 6)  return launch_fast<NB, 6, 256>(a, bits, rowflag, mean, minmax, st);
    if (a.npi <= 256 * 17) return launch_fast<NB, 17, 256>(a, bits, rowflag, mean, minmax, st);
    *done = false;
    return CDFT_OK;
}

// Class kernel for 0/1 basin masks without -max.  The fast kernel above spends NB r8 select-adds per cell, as much as
// the memory stream costs at 5 basins.  A cell's basin membership is one of a few bit patterns, so each cell is given
// the class of its pattern once (cdft_zonal_set_mesh) and its term goes to ONE accumulator: a per-lane, per-class slot
// in LDS, updated by ds_add_f64 (same scheme as cdfmoc's class kernel, moc_zonal.hpp).  A slot only receives its own
// lane's terms in program order => run-to-run identical results.  Basin sums are formed from the class sums at the
// end: basin b = the classes whose pattern has bit b, plus 0 * (the sums of the other classes), which is the NaN the
// reference's 0*zv gives when a NaN/Inf sits on a point outside b, and a zero otherwise.
constexpr int kZC = 8;            // bit patterns the class kernel holds (the empty pattern included)
constexpr int kZStride = 72;      // doubles between the slot rows of two classes (64 lanes + 8: different banks)

#ifndef ZONAL_STEP
#define ZONAL_STEP 1
#endif
struct ZonalClassArgs {
    const uint8_t *cls;           // (npi,npj) class of the cell's basin bit pattern
    uint8_t pat[kZC];
    int nclass;
    int lc;                       // level chunks: workgroup = (chunk, row)
    int lev_max;                  // levels of the longest chunk (the LDS layout behind the slots)
    int *rowflag;                 // MINMAX: rows holding a NaN / Inf (zeroed before the launch)
    double *moc_out;              // NOMASK: (nb, npj, nlev + 1) r8, the negated sums in cdfmoc's order
    int xcd_map;                  // workgroup -> (chunk, row) so that each XCD takes an eighth of the rows
};

// Loads: the row is covered by 16-byte vectors starting at cells 0, 4, 8, ... (r4-aligned only: rows start anywhere),
// and, when npiglo is not a multiple of 4, one more vector anchored at npiglo-4 whose leading cells repeat the previous
// vector's and are sent to an unread slot -- so every load is a full vector inside the row, without a branch.
typedef uint32_t u32u __attribute__((aligned(1)));               // 4 mask bytes at any address

// MINMAX (cdfzonalmean -max, src/cdfzonalmean.f90:316-326): the zonal maximum and minimum by class too -- two more slots per lane and
// class, r4 (the reference keeps its extrema in r4: MAX(r4, r8) rounded back at every cell, which is the r4 rounding of the r8
// extremum because rounding is monotone), updated by ds_max_f32 / ds_min_f32; a cell outside a basin enters its maximum as 0
// (dtmp = 0 there) and never its minimum (dtmp /= 0).  A row that holds a NaN / Inf is flagged for the general kernel, as the
// select kernel does (fmax / fmin drop a NaN the reference's MAX keeps or not by position).
// NOMASK (cdfmoc -decomp, src/cdfmoc.f90:397-419, :472-478): no level mask -- the streamed field IS the masked weight e1v*e3v of
// cdfmoc and `surf` the r8 field of the row (the barotropic or the shear velocity): the term is surf * (double)z, and the result goes out
// negated in cdfmoc's (basin, row, level) order (ca.moc_out), level nlev + 1 written as -0.
template <int NV, bool MEAN, int T, bool M8, bool MINMAX, bool NOMASK = false>
__global__ void __launch_bounds__(T) zonal_row_class_kernel(ZonalArgs2 a, ZonalClassArgs ca) {
    extern __shared__ double lds[];
    constexpr int nwaves = T / 64;
    int chunk = blockIdx.x / a.npj, jj = blockIdx.x - chunk * a.npj;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (ca.xcd_map) {
        // workgroup b runs on XCD b % 8: XCD x takes the x-th eighth of the rows of every level chunk, in row order, so that
        // the 128-byte lines two consecutive rows share (field and byte mask) are asked for twice from the same L2
        // (moc_zonal.hpp has the measurement this follows; CDFT_ZONAL_XCDMAP=0: launch order = row order)
        const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
        const int j0 = (int)((int64_t)x * a.npj / 8), nrows = (int)((int64_t)(x + 1) * a.npj / 8) - j0;
        if (nrows <= 0) return;
        chunk = s / nrows;
        if (chunk >= ca.lc) return;                                  // (uniform: before any barrier)
        jj = j0 + s - chunk * nrows;
    }
    const int k0 = (int)((int64_t)chunk * a.nlev / ca.lc), k1 = (int)((int64_t)(chunk + 1) * a.nlev / ca.lc);
    double *scr  = lds + wave * ((kZC + 1) * kZStride);              // slot [class][lane]; class kZC = cells not owned
    double *part = lds + nwaves * ((kZC + 1) * kZStride);            // [k1-k0][kZC][nwaves]
    // MINMAX: behind part ([lev_max][kZC][nwaves] doubles), r4 slots [wave][max | min][class][lane] and their per-level results
    float *xscr  = reinterpret_cast<float *>(part + (size_t)ca.lev_max * kZC * nwaves) + wave * (2 * (kZC + 1) * kZStride);
    float *xpart = reinterpret_cast<float *>(part + (size_t)ca.lev_max * kZC * nwaves) + nwaves * (2 * (kZC + 1) * kZStride);   // [k1-k0][kZC][nwaves][2]
    uint32_t present = 0u;                                           // classes this lane owns a cell of
    float nonfinite = 0.f;
    const int64_t nij = (int64_t)a.npi * a.npj, r2 = (int64_t)jj * a.npi;
    const int nfull = a.npi >> 2;                                    // vectors 0..nfull-1 start at 4v
    double sf[NV][4];
    // the classes of a vector's four cells, a byte each, and the column it is loaded from are formed again at every use
    // (a bit-field extract and a multiply-add; min) instead of living in 5 * NV registers: one more wave per SIMD
    uint32_t cpk[NV];
    auto i0_of = [&](int r, int tx) -> int { return min(4 * (tx + r * T), a.npi - 4); };   // anchored vector for v >= nfull
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int v = threadIdx.x + r * T;
        const int i0 = i0_of(r, threadIdx.x);
        cpk[r] = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + c;
            const bool own = v < nfull || (v == nfull && i >= 4 * nfull);
            sf[r][c] = a.surf[r2 + i];
            const int cl = own ? min((int)ca.cls[r2 + i], kZC) : kZC;
            cpk[r] |= (uint32_t)cl << (8 * c);      // slot kZC is never read
            if (MINMAX && cl < kZC) present |= 1u << cl;
        }
    }
#pragma unroll
    for (int c = 0; c <= kZC; ++c) scr[c * kZStride + lane] = 0.0;
    if (MINMAX) {
#pragma unroll
        for (int c = 0; c <= kZC; ++c) { xscr[c * kZStride + lane] = -1.e20f; xscr[(kZC + 1 + c) * kZStride + lane] = 1.e20f; }
        for (int o = 32; o > 0; o >>= 1) present |= __shfl_xor(present, o, 64);
    }
    // the level mask: r4 as given (M8 = false), or one byte per cell when it holds only 0 and 1 (the same m*z products)
    struct Mask { f4u v; uint32_t b; };
    f4u zc[NV], zn[NV];
    Mask mc[NV], mn_[NV];
    auto fetch = [&](f4u (&z)[NV], Mask (&mk)[NV], int k) {
        const float *zv = a.zv + (int64_t)k * nij + r2;
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));                    // (or the NV offsets are hoisted into registers)
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int i0 = i0_of(r, tx);
            z[r] = ld_stream4u(zv + i0);
            if (NOMASK) continue;
            if (M8) mk[r].b = __builtin_nontemporal_load(reinterpret_cast<const u32u *>(a.mask8 + (int64_t)k * nij + r2 + i0));
            else    mk[r].v = ld_stream4u(a.mask3 + (int64_t)k * nij + r2 + i0);
        }
    };
    fetch(zc, mc, k0);
    for (int k = k0; k < k1; ++k) {
        if (k + 1 < k1) fetch(zn, mn_, k + 1);          // next level in flight
        if (ZONAL_STEP) __syncthreads();                // nothing shared here: keeps the waves of the row on one level
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const float z4[4] = {zc[r].x, zc[r].y, zc[r].z, zc[r].w};
            uint32_t pk = cpk[r];
            asm volatile("" : "+v"(pk));                // (or the 4 * NV slot addresses are hoisted into registers)
            float m4[4] = {1.f, 1.f, 1.f, 1.f};
            if (NOMASK) {
            } else if (M8) {
#pragma unroll
                for (int c = 0; c < 4; ++c) m4[c] = (float)((mc[r].b >> (8 * c)) & 0xffu);
            } else {
                m4[0] = mc[r].v.x; m4[1] = mc[r].v.y; m4[2] = mc[r].v.z; m4[3] = mc[r].v.w;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // zmask = 1: sum form (1.f*m)*z = m*z in r4; mean form ((1.d0*1)*m)*z = m*z in r8
                const double dt = NOMASK ? (double)z4[c] : MEAN ? __dmul_rn((double)m4[c], (double)z4[c]) : (double)__fmul_rn(m4[c], z4[c]);
                const uint32_t cl = (pk >> (8 * c)) & 0xffu;
                (void)__hip_atomic_fetch_add(&scr[cl * kZStride + lane], __dmul_rn(sf[r][c], dt), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MINMAX) {
                    const float ft = (float)dt;
                    nonfinite = __fmaf_rn(z4[c], 0.f, nonfinite);
                    (void)__hip_atomic_fetch_max(&xscr[cl * kZStride + lane], ft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void)__hip_atomic_fetch_min(&xscr[(kZC + 1 + (dt != 0.0 ? cl : (uint32_t)kZC)) * kZStride + lane], ft, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double s = 0.0;
        {                                               // lane c*8+q sums 8 lanes of class c, then 3 cross-lane steps
            const double *q = scr + (lane >> 3) * kZStride + (lane & 7);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += q[8 * j];
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        float fx = -1.e20f, fn = 1.e20f;
        if (MINMAX) {                                   // the same transposition for the extrema: lane c*8+q takes 8 lanes of class c
            const float *qx = xscr + (lane >> 3) * kZStride + (lane & 7), *qn = qx + (kZC + 1) * kZStride;
#pragma unroll
            for (int j = 0; j < 8; ++j) { fx = fmaxf(fx, qx[8 * j]); fn = fminf(fn, qn[8 * j]); }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { fx = fmaxf(fx, __shfl_xor(fx, o, 64)); fn = fminf(fn, __shfl_xor(fn, o, 64)); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                // every lane has read before the slots are cleared
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < kZC; ++c) scr[c * kZStride + lane] = 0.0;
        if (MINMAX) {
#pragma unroll
            for (int c = 0; c <= kZC; ++c) { xscr[c * kZStride + lane] = -1.e20f; xscr[(kZC + 1 + c) * kZStride + lane] = 1.e20f; }
        }
        if ((lane & 7) == 0) {
            part[((int64_t)(k - k0) * kZC + (lane >> 3)) * nwaves + wave] = s;
            if (MINMAX) {
                float *xp = xpart + (((int64_t)(k - k0) * kZC + (lane >> 3)) * nwaves + wave) * 2;
                xp[0] = fx; xp[1] = fn;
            }
        }
#pragma unroll
        for (int r = 0; r < NV; ++r) { zc[r] = zn[r]; mc[r] = mn_[r]; }
    }
    if (MINMAX) {                                       // a NaN / Inf in the row: the general kernel recomputes it (all its levels)
        const int redo = __syncthreads_or(!__builtin_isfinite(nonfinite));
        if (threadIdx.x == 0 && redo) atomicOr(ca.rowflag + jj, 1);
        __shared__ uint32_t pres[nwaves];
        if (lane == 0) pres[wave] = present;
        __syncthreads();
        present = 0u;
#pragma unroll
        for (int w = 0; w < nwaves; ++w) present |= pres[w];
    } else {
        __syncthreads();
    }
    if (NOMASK && chunk == 0 && (int)threadIdx.x < a.nb) ca.moc_out[((int64_t)a.nlev * a.npj + jj) * a.nb + threadIdx.x] = -0.0;
    for (int t = threadIdx.x; t < (k1 - k0) * a.nb; t += T) {
        const int kk = t / a.nb, b = t - kk * a.nb, k = k0 + kk;
        double s = 0.0, z = 0.0;
        for (int c = 0; c < ca.nclass; ++c) {
            const bool in = (ca.pat[c] >> b) & 1;
            for (int w = 0; w < nwaves; ++w) {
                const double v = part[((int64_t)kk * kZC + c) * nwaves + w];
                if (in) s += v;
                else    z += v;
            }
        }
        if (NOMASK) ca.moc_out[((int64_t)k * a.npj + jj) * a.nb + b] = -(s + __dmul_rn(z, 0.0));
        else        a.dsum[((int64_t)b * a.nlev + k) * a.npj + jj] = s + __dmul_rn(z, 0.0);
        if (MINMAX) {
            float tx = -1.e20f, tn = 1.e20f;
            for (int c = 0; c < ca.nclass; ++c) {
                if ((ca.pat[c] >> b) & 1) {
                    for (int w = 0; w < nwaves; ++w) {
                        const float *xp = xpart + (((int64_t)kk * kZC + c) * nwaves + w) * 2;
                        tx = fmaxf(tx, xp[0]); tn = fminf(tn, xp[1]);
                    }
                } else if ((present >> c) & 1u) {
                    tx = fmaxf(tx, 0.f);                       // a cell of the row outside the basin: dtmp = 0
                }
            }
            a.rmax[((int64_t)b * a.nlev + k) * a.npj + jj] = tx;
            a.rmin[((int64_t)b * a.nlev + k) * a.npj + jj] = tn;
        }
    }
}

constexpr int kClassT = 256, kClassMaxNV = 5;

template <int NV, int T = kClassT>
int launch_class_nv(const ZonalArgs2 &a, const ZonalClassArgs &ca, bool mean, bool minmax, hipStream_t st) {
    constexpr int nw = T / 64;
    size_t shmem = ((size_t)nw * (kZC + 1) * kZStride + (size_t)ca.lev_max * kZC * nw) * sizeof(double);
    if (minmax) shmem += ((size_t)nw * 2 * (kZC + 1) * kZStride + (size_t)ca.lev_max * kZC * nw * 2) * sizeof(float);
    const dim3 grid(ca.xcd_map ? 8 * ((a.npj + 7) / 8) * ca.lc : a.npj * ca.lc);
    if (ca.moc_out) {                                             // cdfmoc -decomp: the weight streamed past an r8 field, no level mask
        hipLaunchKernelGGL((zonal_row_class_kernel<NV, false, T, false, false, true>), grid, dim3(T), shmem, st, a, ca);
        CDFT_LAUNCH_CHECK();
        return CDFT_OK;
    }
    if (minmax) {                                                 // (-max is an option of the MEAN form, src/cdfzonalmean.f90:147)
        const void *kf = a.mask8 ? (const void *)zonal_row_class_kernel<NV, true, T, true, true> : (const void *)zonal_row_class_kernel<NV, true, T, false, true>;
        if (shmem > 48 * 1024) CDFT_TRY(dyn_lds_once(kf, (int)shmem));
        if (a.mask8) hipLaunchKernelGGL((zonal_row_class_kernel<NV, true, T, true, true>), grid, dim3(T), shmem, st, a, ca);
        else         hipLaunchKernelGGL((zonal_row_class_kernel<NV, true, T, false, true>), grid, dim3(T), shmem, st, a, ca);
    } else if (a.mask8) {
        if (mean) hipLaunchKernelGGL((zonal_row_class_kernel<NV, true, T, true, false>), grid, dim3(T), shmem, st, a, ca);
        else      hipLaunchKernelGGL((zonal_row_class_kernel<NV, false, T, true, false>), grid, dim3(T), shmem, st, a, ca);
    } else {
        if (mean) hipLaunchKernelGGL((zonal_row_class_kernel<NV, true, T, false, false>), grid, dim3(T), shmem, st, a, ca);
        else      hipLaunchKernelGGL((zonal_row_class_kernel<NV, false, T, false, false>), grid, dim3(T), shmem, st, a, ca);
    }
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

// rows of 4 .. 5116 cells
inline bool class_fits(int npi) { return npi >= 4 && (npi / 4 + 1 + kClassT - 1) / kClassT <= kClassMaxNV; }

int launch_class(const ZonalArgs2 &a, ZonalClassArgs ca, bool mean, bool minmax, hipStream_t st) {
    const char *e = getenv("CDFT_ZONAL_CHUNKS");                  // A/B hook
    // one workgroup per row for all levels is 1207 workgroups at eORCA025, less than the chip holds: the levels are cut
    // so that there are about 4096 (measured: 238 -> 209 us there, nothing lost at eORCA12)
    ca.lc = e ? std::max(1, std::min(a.nlev, atoi(e))) : std::max(1, std::min(std::min(a.nlev, 8), (4096 + a.npj - 1) / a.npj));
    // -max: its r4 slots double the workgroup's LDS, and a chunk's per-level results are the rest of it -- at least four chunks keep three
    // workgroups on a CU instead of two (eORCA12: 1.77 -> 1.51 ms; profiles/r05 section 11)
    if (minmax && !e) ca.lc = std::max(ca.lc, std::min(a.nlev, 4));
    { const char *x = getenv("CDFT_ZONAL_XCDMAP"); ca.xcd_map = (a.npj >= 64 && !(x && x[0] == '0')) ? 1 : 0; }
    ca.lev_max = (a.nlev + ca.lc - 1) / ca.lc;
    // Workgroups of 128 lanes where they waste fewer lanes on the end of the row than 256 do (a row of 1442 cells is 362
    // vectors: 2 x 256 slots leave 29 % of the lanes idle, 3 x 128 leave 6 %); CDFT_ZONAL_T = 128 | 256: A/B hook
    {
        const int nvec = a.npi / 4 + 1, nv128 = (nvec + 127) / 128, nv256 = (nvec + 255) / 256;
        const char *t = getenv("CDFT_ZONAL_T");
        const bool small = t ? atoi(t) == 128 : nv128 * 128 < nv256 * 256;
        if (small && nv128 <= kClassMaxNV)                        // (rows of up to 2556 cells; at 4322 cells 9 x 128 was measured: 3 % slower than 5 x 256)
            switch (nv128) {
                case 1: return launch_class_nv<1, 128>(a, ca, mean, minmax, st);
                case 2: return launch_class_nv<2, 128>(a, ca, mean, minmax, st);
                case 3: return launch_class_nv<3, 128>(a, ca, mean, minmax, st);
                case 4: return launch_class_nv<4, 128>(a, ca, mean, minmax, st);
                default: return launch_class_nv<5, 128>(a, ca, mean, minmax, st);
            }
    }
    switch ((a.npi / 4 + 1 + kClassT - 1) / kClassT) {
        case 1: return launch_class_nv<1>(a, ca, mean, minmax, st);
        case 2: return launch_class_nv<2>(a, ca, mean, minmax, st);
        case 3: return launch_class_nv<3>(a, ca, mean, minmax, st);
        case 4: return launch_class_nv<4>(a, ca, mean, minmax, st);
        default: return launch_class_nv<5>(a, ca, mean, minmax, st);
    }
}

// mask8 = mask3 as bytes; flag[0] is raised by a value that is neither +0 nor 1
__global__ void zonal_mask8_kernel(const float *__restrict__ m, int64_t n, uint8_t *__restrict__ m8, int *__restrict__ flag) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float v = m[p];
    if (v == 1.f) m8[p] = 1;
    else { m8[p] = 0; if (!(v == 0.f) || signbit(v)) flag[0] = 1; }
}

// cls(i,j) = lut[bits(i,j)]; seen[v] = 1 for every bit pattern v present
__global__ void zonal_seen_kernel(const uint8_t *__restrict__ bits, int64_t nij, int *__restrict__ seen) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nij && !seen[bits[p]]) seen[bits[p]] = 1;              // benign race: every writer stores 1
}
__global__ void zonal_cls_kernel(const uint8_t *__restrict__ bits, int64_t nij, const uint8_t *__restrict__ lut, uint8_t *__restrict__ cls) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nij) cls[p] = lut[bits[p]];
}

int keepz(void **dst, const void *src, size_t bytes, int space, hipStream_t st) {
    if (!*dst && hipMalloc(dst, bytes + 64) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdfzonal: device allocation of %zu bytes failed", bytes);
    }
    CDFT_HIP(hipMemcpyAsync(*dst, src, bytes, space == CDFT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    return CDFT_OK;
}

template <int NB>
int launch_rows(const ZonalArgs2 &a, const int *rowflag, bool mean, bool minmax, hipStream_t st) {
    const size_t shmem = (size_t)a.nlev * (kThreads / 64) * NB * 4 * sizeof(double);
    const dim3 grid(a.npj), block(kThreads);
    if (mean && minmax)  hipLaunchKernelGGL((zonal_row_kernel<NB, true, true, false>), grid, block, shmem, st, a, rowflag);
    else if (mean)       hipLaunchKernelGGL((zonal_row_kernel<NB, true, false, false>), grid, block, shmem, st, a, rowflag);
    else                 hipLaunchKernelGGL((zonal_row_kernel<NB, false, false, false>), grid, block, shmem, st, a, rowflag);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

// darea(b,jj,jk) = sum_i dl_surf*zmask(b)*zmaskvar(jk) (src/cdfzonalmean.f90:326): it does not depend on the field, so it
// is summed once per mesh, for every level, in the order the field sums use
template <int NB>
int launch_area(const ZonalArgs2 &a, hipStream_t st) {
    const size_t shmem = (size_t)a.nlev * (kThreads / 64) * NB * 4 * sizeof(double);
    hipLaunchKernelGGL((zonal_row_kernel<NB, true, false, true>), dim3(a.npj), dim3(kThreads), shmem, st, a, nullptr);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

}  // namespace

namespace cdft {
// cdfmoc -decomp's r8 zonal sums (moc.hip) on this file's class kernel: raw(b, jj, jk) = -sum_i (double)w(i, jj, jk) * field(i, jj) over the cells of
// basin b, jk < nlev; raw(b, jj, nlev) = -0.  cls / pat / nclass: the plan's classes of basin-bit patterns (moc.hip).  false: the shape does not fit
// (rows under 4 or over 5116 cells, more than 8 patterns): the caller keeps its select kernel.
bool zonal_class_fits_moc(int npi, int nclass) { return nclass > 0 && nclass <= kZC && class_fits(npi); }
int zonal_class_moc_r8(const float *w, const double *field, const uint8_t *cls, const uint8_t *pat, int nclass, int npi, int npj, int nlev, int nb,
                       double *raw, hipStream_t st) {
    ZonalArgs2 a;
    a.surf = field; a.mask3 = nullptr; a.bmask = nullptr; a.zv = w; a.mask8 = nullptr;
    a.npi = npi; a.npj = npj; a.nlev = nlev; a.nb = nb;
    a.dsum = nullptr; a.darea = nullptr; a.rmax = nullptr; a.rmin = nullptr;
    ZonalClassArgs ca;
    ca.cls = cls; ca.nclass = nclass; ca.rowflag = nullptr; ca.moc_out = raw;
    for (int c = 0; c < kZC; ++c) ca.pat[c] = c < nclass ? pat[c] : 0;
    return launch_class(a, ca, false, false, st);
}
}  // namespace cdft

extern "C" {

int cdft_zonal_create(cdft_zonal_plan **plan, int npiglo, int npjglo, int npk, int nbasins) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_zonal_create: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 1 || nbasins < 1 || nbasins > kMaxNB)
        return fail(CDFT_ERR_ARG, "cdft_zonal_create: bad sizes %d x %d x %d, %d basins (1..%d)", npiglo, npjglo, npk,
                    nbasins, kMaxNB);
    // the per-level partials of a row live in LDS (64 KB without opting in to more): 5 basins x 4 waves x 4
    // values x 8 bytes = 640 bytes per level, i.e. up to 102 levels
    if ((size_t)npk * 4 * (nbasins <= 1 ? 1 : (nbasins <= 5 ? 5 : 8)) * 4 * 8 > 64 * 1024)
        return fail(CDFT_ERR_ARG, "cdft_zonal_create: %d levels x %d basins exceed the LDS budget of a row", npk, nbasins);
    CDFT_TRY(require_device());
    cdft_zonal_plan *p = new cdft_zonal_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk; p->nb = nbasins;
    *plan = p;
    return CDFT_OK;
}

int cdft_zonal_set_mesh(cdft_zonal_plan *p, const float *e1, const float *e2, const float *zmaskvar, const float *zmask,
                        int memspace, void *stream) {
    if (!p || !e1 || !e2 || !zmaskvar || !zmask) return fail(CDFT_ERR_ARG, "cdft_zonal_set_mesh: NULL argument");
    hipStream_t st = as_stream(stream);
    const size_t nij = (size_t)p->npi * p->npj;
    const void *d1, *d2;
    CDFT_TRY(to_device(e1, nij * 4, memspace, p->st_a, st, &d1));
    CDFT_TRY(to_device(e2, nij * 4, memspace, p->st_b, st, &d2));
    if (!p->surf && hipMalloc((void **)&p->surf, nij * 8 + 64) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdft_zonal_set_mesh: device allocation failed");
    }
    hipLaunchKernelGGL(zonal_surf_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, (const float *)d1,
                       (const float *)d2, p->surf, (int64_t)nij);
    CDFT_LAUNCH_CHECK();
    CDFT_TRY(keepz((void **)&p->mask3, zmaskvar, nij * p->npk * 4, memspace, st));
    CDFT_TRY(keepz((void **)&p->bmask, zmask, nij * p->nb * 4, memspace, st));
    if (!p->mask8 && hipMalloc((void **)&p->mask8, nij * p->npk + 64) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdft_zonal_set_mesh: device allocation failed");
    }
    if (!p->bits && (hipMalloc((void **)&p->bits, nij + 64) != hipSuccess || hipMalloc((void **)&p->flag, sizeof(int)) != hipSuccess ||
                     hipMalloc((void **)&p->rowflag, (size_t)p->npj * sizeof(int)) != hipSuccess)) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdft_zonal_set_mesh: device allocation failed");
    }
    CDFT_HIP(hipMemsetAsync(p->flag, 0, sizeof(int), st));
    hipLaunchKernelGGL(zonal_bits_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, p->bmask, p->bits, p->nb,
                       (int64_t)nij, p->flag);
    CDFT_LAUNCH_CHECK();
    int flag = 0;
    CDFT_HIP(hipMemcpyAsync(&flag, p->flag, sizeof(int), hipMemcpyDeviceToHost, st));
    CDFT_HIP(hipStreamSynchronize(st));
    p->binary = (flag == 0);
    {                                                            // the level masks as bytes, when they are 0 / 1
        const int64_t n3 = (int64_t)nij * p->npk;
        int f8 = 0;
        CDFT_HIP(hipMemsetAsync(p->flag, 0, sizeof(int), st));
        hipLaunchKernelGGL(zonal_mask8_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, p->mask3, n3, p->mask8, p->flag);
        CDFT_LAUNCH_CHECK();
        CDFT_HIP(hipMemcpyAsync(&f8, p->flag, sizeof(int), hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
        p->mask01 = (f8 == 0);
    }
    // class kernel: number the bit patterns present (the empty one included), at most kZC of them
    p->nclass = 0;
    if (p->binary) {                                             // (one basin too: two patterns; the class kernel runs it at 0.65 where the byte-select kernel had 0.39)
        if (!p->cls && (hipMalloc((void **)&p->cls, nij + 64 + 256) != hipSuccess || hipMalloc((void **)&p->seen, 256 * sizeof(int)) != hipSuccess)) {
            (void)hipGetLastError();
            return fail(CDFT_ERR_ALLOC, "cdft_zonal_set_mesh: device allocation failed");
        }
        CDFT_HIP(hipMemsetAsync(p->seen, 0, 256 * sizeof(int), st));
        hipLaunchKernelGGL(zonal_seen_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, p->bits, (int64_t)nij, p->seen);
        CDFT_LAUNCH_CHECK();
        int seen[256];
        CDFT_HIP(hipMemcpyAsync(seen, p->seen, sizeof(seen), hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
        uint8_t lut[256];
        for (int v = 0; v < 256; ++v) {
            lut[v] = 0xFF;
            if (seen[v]) {
                if (p->nclass < kZC) { p->pat[p->nclass] = (uint8_t)v; lut[v] = (uint8_t)p->nclass; }
                ++p->nclass;
            }
        }
        if (p->nclass > kZC) p->nclass = 0;
        if (p->nclass > 0) {
            uint8_t *dlut = p->cls + nij + 64;
            CDFT_HIP(hipMemcpyAsync(dlut, lut, 256, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(zonal_cls_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, p->bits, (int64_t)nij, dlut, p->cls);
            CDFT_LAUNCH_CHECK();
            CDFT_HIP(hipStreamSynchronize(st));                  // lut[] is a stack array
        }
    }
    if (!p->area && hipMalloc((void **)&p->area, nij / p->npi * p->npk * p->nb * sizeof(double) + 64) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdft_zonal_set_mesh: device allocation failed");
    }
    ZonalArgs2 a;
    a.surf = p->surf; a.mask3 = p->mask3; a.mask8 = nullptr; a.bmask = p->bmask; a.zv = p->mask3;      // any finite field: only darea is kept
    a.npi = p->npi; a.npj = p->npj; a.nlev = p->npk; a.nb = p->nb;
    a.dsum = nullptr; a.darea = p->area; a.rmax = nullptr; a.rmin = nullptr;
    if (p->nb == 1)      CDFT_TRY(launch_area<1>(a, st));
    else if (p->nb <= 5) CDFT_TRY(launch_area<5>(a, st));
    else                 CDFT_TRY(launch_area<8>(a, st));
    p->st_a.release(); p->st_b.release();
    p->mesh_set = true;
    return CDFT_OK;
}

int cdft_zonal_compute(cdft_zonal_plan *p, const float *zv, int nlev, int in_space, int mean_form, double *dzo,
                       double *darea, float *rzomax, float *rzomin, int out_space, void *stream) {
    TraceRange trace_("cdft_zonal_compute");
    if (!p || !zv || !dzo) return fail(CDFT_ERR_ARG, "cdft_zonal_compute: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_zonal_compute: cdft_zonal_set_mesh has not succeeded");
    if (nlev < 1 || nlev > p->npk) return fail(CDFT_ERR_ARG, "cdft_zonal_compute: nlev = %d outside 1..%d", nlev, p->npk);
    if (out_space != CDFT_MEM_HOST && out_space != CDFT_MEM_DEVICE && out_space != CDFT_MEM_PINNED)
        return fail(CDFT_ERR_ARG, "cdft_zonal_compute: bad memory-space tag %d", out_space);
    const bool to_host = out_space != CDFT_MEM_DEVICE;
    const bool minmax = rzomax != nullptr || rzomin != nullptr;
    if (minmax && (!rzomax || !rzomin || !mean_form))
        return fail(CDFT_ERR_ARG, "cdft_zonal_compute: rzomax and rzomin go together, with the mean form");
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj;
    const void *dz;
    CDFT_TRY(to_device(zv, (size_t)nij * nlev * 4, in_space, p->st_zv, st, &dz));
    const size_t n = (size_t)p->npj * nlev * p->nb;
    ZonalArgs2 a;
    a.surf = p->surf; a.mask3 = p->mask3; a.bmask = p->bmask; a.zv = (const float *)dz;
    {
        const char *e = getenv("CDFT_ZONAL_R4MASK");             // test hook: the class kernel on the r4 level masks
        a.mask8 = (p->mask01 && !(e && e[0] == '1')) ? p->mask8 : nullptr;
    }
    a.npi = p->npi; a.npj = p->npj; a.nlev = nlev; a.nb = p->nb;
    a.dsum = dzo; a.darea = darea; a.rmax = rzomax; a.rmin = rzomin;
    // Page-locked results arrive by ONE copy queued behind the kernel (no staging copy on the host, no synchronisation here):
    // 3.6 MB = 0.09 ms behind a 0.19 ms kernel at eORCA025.  Having the kernels store them where the host reads them
    // (CDFT_ZONAL_PINNED_DIRECT=1, kept as the A/B hook) was measured and is twice as slow, 0.52-0.56 ms a call: 450 000
    // isolated 8-byte stores across PCIe, one per (basin, level, row), where the copy moves whole pages.
    bool direct = false;
    if (out_space == CDFT_MEM_PINNED && !minmax && getenv("CDFT_ZONAL_PINNED_DIRECT")) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, dzo, 0) == hipSuccess && dp) { a.dsum = (double *)dp; direct = true; }
        else (void)hipGetLastError();
    }
    if (to_host && !direct) {
        CDFT_TRY(p->st_o[0].reserve(n * 8)); a.dsum = (double *)p->st_o[0].ptr;
        if (minmax) {
            CDFT_TRY(p->st_o[2].reserve(n * 4)); CDFT_TRY(p->st_o[3].reserve(n * 4));
            a.rmax = (float *)p->st_o[2].ptr; a.rmin = (float *)p->st_o[3].ptr;
        }
    }
    bool done = false;
    const bool force_general = getenv("CDFT_ZONAL_GENERAL") != nullptr;            // A/B and test hooks
    const bool no_class = getenv("CDFT_ZONAL_NOCLASS") != nullptr;
    const bool class_minmax = !(getenv("CDFT_ZONAL_MAX_SELECT"));                    // A/B hook: -max on the select kernel (rounds 2-4)
    if (p->binary && !force_general && !no_class && (!minmax || class_minmax) && p->nclass > 0 && class_fits(p->npi)) {
        ZonalClassArgs ca;
        ca.cls = p->cls; ca.nclass = p->nclass; ca.rowflag = p->rowflag; ca.moc_out = nullptr;
        for (int c = 0; c < kZC; ++c) ca.pat[c] = p->pat[c];
        if (minmax) CDFT_HIP(hipMemsetAsync(p->rowflag, 0, (size_t)p->npj * sizeof(int), st));
        CDFT_TRY(launch_class(a, ca, mean_form != 0, minmax, st));
        if (minmax) {                                             // the rows it flagged (a NaN / Inf met): the general kernel
            if (p->nb == 1)      CDFT_TRY(launch_rows<1>(a, p->rowflag, true, true, st));
            else if (p->nb <= 5) CDFT_TRY(launch_rows<5>(a, p->rowflag, true, true, st));
            else                 CDFT_TRY(launch_rows<8>(a, p->rowflag, true, true, st));
        }
    } else {
    if (p->binary && !force_general) {
        if (p->nb == 1)      CDFT_TRY(launch_fast_nb<1>(a, p->bits, p->rowflag, mean_form != 0, minmax, st, &done));
        else if (p->nb <= 5) CDFT_TRY(launch_fast_nb<5>(a, p->bits, p->rowflag, mean_form != 0, minmax, st, &done));
        else                 CDFT_TRY(launch_fast_nb<8>(a, p->bits, p->rowflag, mean_form != 0, minmax, st, &done));
    }
    // the general kernel: every row when the fast one did not run, else the rows it flagged (NaN/Inf met)
    const int *only = done ? p->rowflag : nullptr;
    if (p->nb == 1)      CDFT_TRY(launch_rows<1>(a, only, mean_form != 0, minmax, st));
    else if (p->nb <= 5) CDFT_TRY(launch_rows<5>(a, only, mean_form != 0, minmax, st));
    else                 CDFT_TRY(launch_rows<8>(a, only, mean_form != 0, minmax, st));
    }
    if (mean_form && darea)      // the first nlev levels of every basin's precomputed area (a pitched copy: measured faster than plain ones)
        CDFT_HIP(hipMemcpy2DAsync(darea, (size_t)nlev * p->npj * 8, p->area, (size_t)p->npk * p->npj * 8, (size_t)nlev * p->npj * 8,
                                  p->nb, to_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    if (to_host && !direct) {
        CDFT_HIP(hipMemcpyAsync(dzo, a.dsum, n * 8, hipMemcpyDeviceToHost, st));
        if (minmax) {
            CDFT_HIP(hipMemcpyAsync(rzomax, a.rmax, n * 4, hipMemcpyDeviceToHost, st));
            CDFT_HIP(hipMemcpyAsync(rzomin, a.rmin, n * 4, hipMemcpyDeviceToHost, st));
        }
        if (out_space == CDFT_MEM_HOST) CDFT_HIP(hipStreamSynchronize(st));      // (page-locked results: the caller synchronises)
    }
    return CDFT_OK;
}

int cdft_zonal_area(cdft_zonal_plan *p, int nlev, double *darea, int out_space, void *stream) {
    if (!p || !darea) return fail(CDFT_ERR_ARG, "cdft_zonal_area: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_zonal_area: cdft_zonal_set_mesh has not succeeded");
    if (nlev < 1 || nlev > p->npk) return fail(CDFT_ERR_ARG, "cdft_zonal_area: nlev = %d outside 1..%d", nlev, p->npk);
    hipStream_t st = as_stream(stream);
    CDFT_HIP(hipMemcpy2DAsync(darea, (size_t)nlev * p->npj * 8, p->area, (size_t)p->npk * p->npj * 8, (size_t)nlev * p->npj * 8, p->nb,
                              out_space == CDFT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    if (out_space != CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(st));
    return CDFT_OK;
}

int cdft_zonal_destroy(cdft_zonal_plan *p) {
    if (!p) return CDFT_OK;
    void *ptrs[] = {p->surf, p->mask3, p->mask8, p->bmask, p->bits, p->flag, p->rowflag, p->area, p->cls, p->seen};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    p->st_a.release(); p->st_b.release(); p->st_zv.release();
    for (DevBuf &b : p->st_o) b.release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"
