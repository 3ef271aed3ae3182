// column.hip -- per-column vertical integrals: cdfvint and cdfvertmean (C ABI section "cdfvint / cdfvertmean",
// SURVEY.md 8f rank 2).  Every cell (i,j) walks its levels in order on its own: the r8 sums are sequential
// exactly as in the reference, so both results are BIT-IDENTICAL to the reference loops.
//   cdfvint     src/cdfvint.f90:266-311     cumulated vertical integral, written at every output depth
//   cdfvertmean src/cdfvertmean.f90:236-269 mean between two depths
// One thread per cell; a wave reads 256 contiguous bytes of each field per level (12 bytes per cell and
// level in, cdfvint also 4 bytes per cell and output depth out).
#include "common.hpp"

using namespace cdft;

namespace {

DevBuf g_ci[4], g_co, g_cs;

// schedule of cdfvint's level loop, which does not depend on (i,j): for model level k, kout[k] = index of
// the output depth written after this level (-1: none), hit[k] = 1 when :285 holds (the level contains the
// output depth: its thickness is clipped to clip[k]), div[k] = rdep2 for -tmean / -smean.
struct VintSched { const int *kout; const int *hit; const float *clip; const float *div; };

__global__ void vint_cumul_kernel(const float *__restrict__ zt, const float *__restrict__ e3, const float *__restrict__ e31d,
                                  const float *__restrict__ tmask, int64_t nij, int npk, const int *__restrict__ kout,
                                  const int *__restrict__ hit, const float *__restrict__ clip, const float *__restrict__ dv,
                                  float sclf, int mean, float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    double v1 = 0.0;                                            // dl_vint1
    for (int k = 0; k < npk; ++k) {
        const int64_t e = (int64_t)k * nij + p;
        const float t = ld_stream(zt + e), m = ld_stream(tmask + e);
        float ez = e3 ? ld_stream(e3 + e) : e31d[k];
        double v2 = v1;                                         // :275 dl_vint2 = dl_vint1
        v1 = __dadd_rn(v1, (double)__fmul_rn(__fmul_rn(t, ez), m));   // :283 zt*e3t*tmask*1.d0
        if (hit[k]) {                                           // :285-290
            if (ez > clip[k]) ez = clip[k];
            v2 = __dadd_rn(v2, (double)__fmul_rn(__fmul_rn(t, ez), m));
        }
        if (kout[k] >= 0) {                                     // :292-311
            v2 = __dmul_rn(v2, (double)sclf);
            float r = (float)v2;
            if (mean) r = __fdiv_rn(r, dv[k]);                  // REAL(dl_vint2)/rdep2
            out[(int64_t)kout[k] * nij + p] = r;
        }
    }
}

__global__ void vertmean_kernel(const float *__restrict__ zv, const float *__restrict__ e3, const float *__restrict__ e31d,
                                const float *__restrict__ zmask, int64_t nij, int nlev, float gdep_first, float gdep_last,
                                float rdep_up, float rdep_down, double *__restrict__ dmean) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    double dvol = 0.0, dsum = 0.0;
    for (int k = 0; k < nlev; ++k) {
        const int64_t e = (int64_t)k * nij + p;
        const float v = ld_stream(zv + e), m = ld_stream(zmask + e);
        float ez = e3 ? ld_stream(e3 + e) : e31d[k];
        if (k == 0) {                                           // src/cdfvertmean.f90:250-253
            const float hdep = __fadd_rn(gdep_first, ez);
            ez = fminf(ez, __fsub_rn(hdep, rdep_up));
        }
        if (k == nlev - 1) ez = fminf(ez, __fsub_rn(rdep_down, gdep_last));   // :255-257
        dvol = __dadd_rn((double)__fmul_rn(ez, m), dvol);                       // :260 e3*zmask*1.d0 + dvol2d
        dsum = __dadd_rn((double)__fmul_rn(__fmul_rn(v, ez), m), dsum);         // :261
    }
    dmean[p] = (dvol != 0.0) ? __ddiv_rn(dsum, dvol) : dsum;                    // :270
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
    CDFT_HIP(hipMemsetAsync(o, 0, ob, st));                    // an output depth below the last level stays 0
    hipLaunchKernelGGL(vint_cumul_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, (const float *)dz,
                       (const float *)de, (const float *)d1, (const float *)dm, nij, npk, (const int *)sb,
                       (const int *)(sb + (size_t)npk * 4), (const float *)(sb + (size_t)npk * 8),
                       (const float *)(sb + (size_t)npk * 12), sclf, mean, o);
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
    hipLaunchKernelGGL(vertmean_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, (const float *)dz,
                       (const float *)de, (const float *)d1, (const float *)dm, nij, nlev, gdep_first, gdep_last, rdep_up,
                       rdep_down, o);
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
