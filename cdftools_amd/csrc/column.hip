// column.hip -- per-column vertical integrals: cdfvint and cdfvertmean (C ABI section "cdfvint / cdfvertmean",
// SURVEY.md 8f rank 2).  Every cell (i,j) walks its levels in order on its own: the r8 sums are sequential
// exactly as in the reference, so both results are BIT-IDENTICAL to the reference loops.
//   cdfvint     src/cdfvint.f90:266-311     cumulated vertical integral, written at every output depth
//   cdfvertmean src/cdfvertmean.f90:236-269 mean between two depths
// One thread per 4 consecutive cells (16-byte nontemporal loads at any r4 alignment, 4 levels of loads in flight);
// 12 bytes per cell and level in, cdfvint also 4 bytes per cell and output depth out.
#include "common.hpp"

#include <cstdlib>
#include <vector>

using namespace cdft;

namespace {

DevBuf g_ci[4], g_co, g_cs;

// schedule of cdfvint's level loop, which does not depend on (i,j): for model level k, kout[k] = index of
// the output depth written after this level (-1: none), hit[k] = 1 when :285 holds (the level contains the
// output depth: its thickness is clipped to clip[k]), div[k] = rdep2 for -tmean / -smean.
struct VintSched { const int *kout; const int *hit; const float *clip; const float *div; };

// VEC cells per thread (VEC = 4: one 16-byte load per field and level; the vectors start at cells 0, 4, 8, ... and the
// last one is anchored at nij-4, so its first cells repeat the previous thread's and both store the same values), KU
// levels of loads issued before the sums that use them.
constexpr int kColKU = 4;
#ifndef CDFT_COL_THREADS
#define CDFT_COL_THREADS 1024
#endif
constexpr int kColThreads = CDFT_COL_THREADS;
// Nothing is shared between threads; the barrier keeps the waves of a workgroup on the same levels, so that their
// loads form one contiguous run per field and level.
#ifndef CDFT_COL_NOSYNC
#define COL_SYNC() __syncthreads()
#else
#define COL_SYNC() (void)0
#endif

template <int VEC> struct ColVec;
template <> struct ColVec<1> {
    static __device__ __forceinline__ void ld(const float *p, float (&v)[1]) { v[0] = ld_stream(p); }
    static __device__ __forceinline__ void st(float *p, const float (&v)[1]) { __builtin_nontemporal_store(v[0], p); }
};
template <> struct ColVec<4> {
    static __device__ __forceinline__ void ld(const float *p, float (&v)[4]) {
        const f4u x = ld_stream4u(p);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    }
    static __device__ __forceinline__ void st(float *p, const float (&v)[4]) {
        f4u x; x.x = v[0]; x.y = v[1]; x.z = v[2]; x.w = v[3];
        __builtin_nontemporal_store(x, reinterpret_cast<f4u *>(p));
    }
};

// First cell of this thread's vector; threads past the end repeat the last vector (same loads, same values stored), so
// every thread of a workgroup runs the same loop and meets the same barriers.
template <int VEC>
__device__ __forceinline__ int64_t col_first_cell(int64_t nij) {
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    return p + VEC <= nij ? p : nij - VEC;
}

template <int VEC, bool E3D>
__global__ void __launch_bounds__(kColThreads)
vint_cumul_kernel(const float *__restrict__ zt, const float *__restrict__ e3, const float *__restrict__ e31d,
                  const float *__restrict__ tmask, int64_t nij, int npk, const int *__restrict__ kout,
                  const int *__restrict__ hit, const float *__restrict__ clip, const float *__restrict__ dv,
                  float sclf, int mean, float *__restrict__ out) {
    const int64_t p = col_first_cell<VEC>(nij);
    double v1[VEC];                                             // dl_vint1
#pragma unroll
    for (int c = 0; c < VEC; ++c) v1[c] = 0.0;
    for (int k0 = 0; k0 < npk; k0 += kColKU) {
        float t[kColKU][VEC], m[kColKU][VEC], ez[kColKU][VEC];
        COL_SYNC();
#pragma unroll
        for (int u = 0; u < kColKU; ++u) {
            if (k0 + u < npk) {
                const int64_t e = (int64_t)(k0 + u) * nij + p;
                ColVec<VEC>::ld(zt + e, t[u]);
                ColVec<VEC>::ld(tmask + e, m[u]);
                if (E3D) ColVec<VEC>::ld(e3 + e, ez[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kColKU; ++u) {
            const int k = k0 + u;
            if (k >= npk) break;
            const int ko = kout[k], h = hit[k];
            const float cl = clip[k], e1 = E3D ? 0.f : e31d[k];
            float r[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float ezc = E3D ? ez[u][c] : e1;
                double v2 = v1[c];                              // :275 dl_vint2 = dl_vint1
                v1[c] = __dadd_rn(v1[c], (double)__fmul_rn(__fmul_rn(t[u][c], ezc), m[u][c]));   // :283 zt*e3t*tmask*1.d0
                if (h) {                                        // :285-290
                    if (ezc > cl) ezc = cl;
                    v2 = __dadd_rn(v2, (double)__fmul_rn(__fmul_rn(t[u][c], ezc), m[u][c]));
                }
                v2 = __dmul_rn(v2, (double)sclf);               // :292-311
                r[c] = (float)v2;
                if (mean) r[c] = __fdiv_rn(r[c], dv[k]);        // REAL(dl_vint2)/rdep2
            }
            if (ko >= 0) ColVec<VEC>::st(out + (int64_t)ko * nij + p, r);
        }
    }
}

template <int VEC, bool E3D>
__global__ void __launch_bounds__(kColThreads)
vertmean_kernel(const float *__restrict__ zv, const float *__restrict__ e3, const float *__restrict__ e31d,
                const float *__restrict__ zmask, int64_t nij, int nlev, float gdep_first, float gdep_last,
                float rdep_up, float rdep_down, double *__restrict__ dmean) {
    const int64_t p = col_first_cell<VEC>(nij);
    double dvol[VEC], dsum[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) { dvol[c] = 0.0; dsum[c] = 0.0; }
    for (int k0 = 0; k0 < nlev; k0 += kColKU) {
        float v[kColKU][VEC], m[kColKU][VEC], ez[kColKU][VEC];
        COL_SYNC();
#pragma unroll
        for (int u = 0; u < kColKU; ++u) {
            if (k0 + u < nlev) {
                const int64_t e = (int64_t)(k0 + u) * nij + p;
                ColVec<VEC>::ld(zv + e, v[u]);
                ColVec<VEC>::ld(zmask + e, m[u]);
                if (E3D) ColVec<VEC>::ld(e3 + e, ez[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kColKU; ++u) {
            const int k = k0 + u;
            if (k >= nlev) break;
            const float e1 = E3D ? 0.f : e31d[k];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float ezc = E3D ? ez[u][c] : e1;
                if (k == 0) {                                   // src/cdfvertmean.f90:250-253
                    const float hdep = __fadd_rn(gdep_first, ezc);
                    ezc = fminf(ezc, __fsub_rn(hdep, rdep_up));
                }
                if (k == nlev - 1) ezc = fminf(ezc, __fsub_rn(rdep_down, gdep_last));   // :255-257
                dvol[c] = __dadd_rn((double)__fmul_rn(ezc, m[u][c]), dvol[c]);                          // :260 e3*zmask*1.d0 + dvol2d
                dsum[c] = __dadd_rn((double)__fmul_rn(__fmul_rn(v[u][c], ezc), m[u][c]), dsum[c]);      // :261
            }
        }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) dmean[p + c] = (dvol[c] != 0.0) ? __ddiv_rn(dsum[c], dvol[c]) : dsum[c];   // :270
}

// CDFT_COLUMN_SCALAR=1 keeps one cell per thread (tests compare the two).
// Workgroup size: 1024 threads (16 waves in step, the longest runs); 256 measured the same or slower at eORCA025 and
// eORCA12.  CDFT_COLUMN_THREADS overrides (A/B hook).
inline int column_threads() {
    if (const char *e = getenv("CDFT_COLUMN_THREADS")) {
        const int t = atoi(e);
        if (t == 256 || t == 512 || t == 1024) return t;
    }
    return kColThreads;
}

inline int column_vec(int64_t nij) {
    const char *e = getenv("CDFT_COLUMN_SCALAR");
    return (nij >= 4 && !(e && e[0] == '1')) ? 4 : 1;
}

}  // namespace

extern "C" {

int cdft_vint_cumul(const float *zt, const float *e3t, const float *e31d, const float *tmask, int npiglo, int npjglo,
                    int npk, const int *kout, const int *hit, const float *clip, const float *rdep2, int npko, float sclf,
                    int mean, int in_space, float *out, int out_space, void *stream) {
    if (!zt || !tmask || (!e3t && !e31d) || !kout || !hit || !clip || !rdep2 || !out)
        return fail(CDFT_ERR_ARG, "cdft_vint_cumul: NULL argument");
    if (npiglo < 1 || npjglo < 1 || npk < 1 || npko < 1) return fail(CDFT_ERR_ARG, "cdft_vint_cumul: bad sizes");
    for (int k = 0; k < npk; ++k)
        if (kout[k] >= npko) return fail(CDFT_ERR_ARG, "cdft_vint_cumul: kout(%d) = %d outside the %d output depths", k + 1, kout[k], npko);
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)npiglo * npjglo;
    const void *dz, *dm, *de = nullptr, *d1 = nullptr;
    CDFT_TRY(to_device(zt, (size_t)nij * npk * 4, in_space, g_ci[0], st, &dz));
    CDFT_TRY(to_device(tmask, (size_t)nij * npk * 4, in_space, g_ci[1], st, &dm));
    if (e3t) CDFT_TRY(to_device(e3t, (size_t)nij * npk * 4, in_space, g_ci[2], st, &de));
    else     CDFT_TRY(to_device(e31d, (size_t)npk * 4, in_space, g_ci[3], st, &d1));
    // the schedule is always a set of small HOST arrays
    CDFT_TRY(g_cs.reserve((size_t)npk * 16));
    char *sb = (char *)g_cs.ptr;
    CDFT_HIP(hipMemcpyAsync(sb, kout, (size_t)npk * 4, hipMemcpyHostToDevice, st));
    CDFT_HIP(hipMemcpyAsync(sb + (size_t)npk * 4, hit, (size_t)npk * 4, hipMemcpyHostToDevice, st));
    CDFT_HIP(hipMemcpyAsync(sb + (size_t)npk * 8, clip, (size_t)npk * 4, hipMemcpyHostToDevice, st));
    CDFT_HIP(hipMemcpyAsync(sb + (size_t)npk * 12, rdep2, (size_t)npk * 4, hipMemcpyHostToDevice, st));
    float *o = out;
    const size_t ob = (size_t)nij * npko * 4;
    if (out_space == CDFT_MEM_HOST) { CDFT_TRY(g_co.reserve(ob)); o = (float *)g_co.ptr; }
    {                                                          // an output depth no level writes stays 0
        std::vector<char> written((size_t)npko, 0);
        for (int k = 0; k < npk; ++k)
            if (kout[k] >= 0) written[(size_t)kout[k]] = 1;
        for (int q = 0; q < npko; ++q)
            if (!written[(size_t)q]) CDFT_HIP(hipMemsetAsync(o + (size_t)q * nij, 0, (size_t)nij * 4, st));
    }
    const int vec = column_vec(nij);
    const int64_t nth = (nij + vec - 1) / vec;
    const int threads = column_threads();
    const dim3 grid((unsigned)((nth + threads - 1) / threads));
#define CDFT_VINT_LAUNCH(V, E)                                                                                         \
    hipLaunchKernelGGL((vint_cumul_kernel<V, E>), grid, dim3(threads), 0, st, (const float *)dz, (const float *)de,        \
                       (const float *)d1, (const float *)dm, nij, npk, (const int *)sb,                                \
                       (const int *)(sb + (size_t)npk * 4), (const float *)(sb + (size_t)npk * 8),                     \
                       (const float *)(sb + (size_t)npk * 12), sclf, mean, o)
    if (vec == 4) { if (de) CDFT_VINT_LAUNCH(4, true); else CDFT_VINT_LAUNCH(4, false); }
    else          { if (de) CDFT_VINT_LAUNCH(1, true); else CDFT_VINT_LAUNCH(1, false); }
#undef CDFT_VINT_LAUNCH
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(out, o, ob, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_vertmean(const float *zv, const float *e3, const float *e31d, const float *zmask, int npiglo, int npjglo, int nlev,
                  float gdep_first, float gdep_last, float rdep_up, float rdep_down, int in_space, double *dvertmean,
                  int out_space, void *stream) {
    if (!zv || !zmask || (!e3 && !e31d) || !dvertmean) return fail(CDFT_ERR_ARG, "cdft_vertmean: NULL argument");
    if (npiglo < 1 || npjglo < 1 || nlev < 1) return fail(CDFT_ERR_ARG, "cdft_vertmean: bad sizes");
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)npiglo * npjglo;
    const void *dz, *dm, *de = nullptr, *d1 = nullptr;
    CDFT_TRY(to_device(zv, (size_t)nij * nlev * 4, in_space, g_ci[0], st, &dz));
    CDFT_TRY(to_device(zmask, (size_t)nij * nlev * 4, in_space, g_ci[1], st, &dm));
    if (e3) CDFT_TRY(to_device(e3, (size_t)nij * nlev * 4, in_space, g_ci[2], st, &de));
    else    CDFT_TRY(to_device(e31d, (size_t)nlev * 4, in_space, g_ci[3], st, &d1));
    double *o = dvertmean;
    if (out_space == CDFT_MEM_HOST) { CDFT_TRY(g_co.reserve((size_t)nij * 8)); o = (double *)g_co.ptr; }
    const int vec = column_vec(nij);
    const int64_t nth = (nij + vec - 1) / vec;
    const int threads = column_threads();
    const dim3 grid((unsigned)((nth + threads - 1) / threads));
#define CDFT_VERTMEAN_LAUNCH(V, E)                                                                                     \
    hipLaunchKernelGGL((vertmean_kernel<V, E>), grid, dim3(threads), 0, st, (const float *)dz, (const float *)de,          \
                       (const float *)d1, (const float *)dm, nij, nlev, gdep_first, gdep_last, rdep_up, rdep_down, o)
    if (vec == 4) { if (de) CDFT_VERTMEAN_LAUNCH(4, true); else CDFT_VERTMEAN_LAUNCH(4, false); }
    else          { if (de) CDFT_VERTMEAN_LAUNCH(1, true); else CDFT_VERTMEAN_LAUNCH(1, false); }
#undef CDFT_VERTMEAN_LAUNCH
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(dvertmean, o, (size_t)nij * 8, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_column_release(void) {
    for (DevBuf &b : g_ci) b.release();
    g_co.release(); g_cs.release();
    return CDFT_OK;
}

}  // extern "C"
