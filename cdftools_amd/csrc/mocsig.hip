// mocsig.hip -- cdfmocsig: meridional overturning in density classes (C ABI section "cdfmocsig",
// SURVEY.md 8f rank 3).  Reference loop src/cdfmocsig.f90:368-470 for one record:
//   per level: zv (fill -> 0, + eddy-induced velocity), zt, zs (fill -> 0), zarea = e1v*e3v,
//              dens = sigmai(zt, zs, pref) ; zttmp = REAL(dens*itmask) ; ibin = INT((zttmp-sigmin)/sigstp)
//              clamped to 1..nbins ; dmoc(b,ibin,jj) -= zv*zarea * ibmask(b,ji,jj)  over ji = 2..npiglo-1
//              [-isodep: depi += gdep(jk)*itmask*zarea, wdep += itmask*zarea, binned the same way]
//   then    : dmoc(:,nbins,:) /= 1.e6 ; dmoc(:,jbin,:) = dmoc(:,jbin+1,:) + dmoc(:,jbin,:)/1.e6, high to low density
// The scatter into density bins is a histogram: one workgroup owns a row jj for all levels, every wave keeps
// its own (basin, bin) accumulators in LDS and adds with ds_add_f64 (an LDS atomic: one instruction per
// cell and basin).  The waves' histograms are added in wave order at the end.  Terms are formed exactly
// as the reference writes them; the r8 additions along i are regrouped (|d| <= 1e-13 * sum|terms|).
#include "eos_dev.hpp"

using namespace cdft;

namespace {

DevBuf g_mi[8], g_mo[2];

struct SigArgs {
    const float *zv, *zeiv, *zt, *zs, *e1v, *e3v, *e31d, *gdep;
    const int16_t *ib;                 // (nb,npi,npj)
    int npi, npj, npk, nb, nbins, j0, j1;
    float zspv, zspt, zsps, pref, sigmin, sigstp;
    double *dmoc, *depi;               // (nb,nbins,npj)
};

template <int TEOS, bool ISODEP>
__global__ void mocsig_row_kernel(SigArgs a) {
    extern __shared__ double hist[];                    // [nwaves][nq][nb][nbins]
    const int jj = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    constexpr int nq = ISODEP ? 3 : 1;
    const int hsz = nq * a.nb * a.nbins;
    double *h = hist + (int64_t)wave * hsz;
    for (int t = lane; t < hsz; t += 64) h[t] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int64_t nij = (int64_t)a.npi * a.npj, r2 = (int64_t)jj * a.npi;
    if (jj >= a.j0 && jj <= a.j1) {                     // rows ij1..ij2 (:405-407); the others stay 0
        for (int k = 0; k < a.npk - 1; ++k) {
            const float e31 = a.e31d ? a.e31d[k] : 0.f, gd = ISODEP ? a.gdep[k] : 0.f;
            for (int i = 1 + (int)threadIdx.x; i < a.npi - 1; i += blockDim.x) {     // ji = 2..npiglo-1
                const int64_t p = r2 + i, e = (int64_t)k * nij + p;
                float v = ld_stream(a.zv + e), t = ld_stream(a.zt + e), s = ld_stream(a.zs + e);
                if (v == a.zspv) v = 0.f;                                                  // :374
                if (a.zeiv) v = __fadd_rn(v, ld_stream(a.zeiv + e));                       // :375-378
                if (t == a.zspt) t = 0.f;                                                  // :382
                if (s == a.zsps) s = 0.f;                                                  // :383
                const float ez = a.e3v ? ld_stream(a.e3v + e) : e31;
                const float zarea = __fmul_rn(a.e1v[p], ez);                               // :389
                const int itm = (s == a.zsps) ? 0 : 1;                                     // :392-393
                const double dens = sigmai_point<TEOS, false>(t, s, a.pref);               // :395
                const float zttmp = (float)__dmul_rn(dens, (double)itm);                   // :398
                int ibin = (int)__fdiv_rn(__fsub_rn(zttmp, a.sigmin), a.sigstp);           // :400
                ibin = min(max(ibin, 1), a.nbins) - 1;                                     // :401-402
                const double term = -(double)__fmul_rn(v, zarea);                          // :417
                const double dd = ISODEP ? (double)__fmul_rn(__fmul_rn(gd, (float)itm), zarea) : 0.0;   // :424
                const double dw = ISODEP ? (double)__fmul_rn((float)itm, zarea) : 0.0;                   // :425
                for (int b = 0; b < a.nb; ++b) {
                    const int16_t m = a.ib[p * a.nb + b];
                    if (m == 0) continue;                                                  // adds an exact zero
                    const double fm = (double)m;
                    (void)__hip_atomic_fetch_add(&h[b * a.nbins + ibin], __dmul_rn(term, fm), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);            // :435
                    if (ISODEP) {
                        (void)__hip_atomic_fetch_add(&h[(a.nb + b) * a.nbins + ibin], __dmul_rn(dd, fm), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);        // :444
                        (void)__hip_atomic_fetch_add(&h[(2 * a.nb + b) * a.nbins + ibin], __dmul_rn(dw, fm), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);        // :445
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < hsz; t += blockDim.x) {      // waves added in wave order
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += hist[(int64_t)w * hsz + t];
        hist[t] = s;                                            // wave 0's slot t is only read by this thread
    }
    __syncthreads();
    if ((int)threadIdx.x < a.nb) {                              // :463-467: from the highest density down
        const int b = threadIdx.x;
        const double *hm = hist + b * a.nbins;
        double acc = __ddiv_rn(hm[a.nbins - 1], (double)1.e6f);
        a.dmoc[((int64_t)jj * a.nbins + (a.nbins - 1)) * a.nb + b] = acc;
        for (int n = a.nbins - 2; n >= 0; --n) {
            acc = __dadd_rn(acc, __ddiv_rn(hm[n], (double)1.e6f));
            a.dmoc[((int64_t)jj * a.nbins + n) * a.nb + b] = acc;
        }
    }
    if (ISODEP) {                                               // :454-460
        for (int t = threadIdx.x; t < a.nb * a.nbins; t += blockDim.x) {
            const int b = t / a.nbins, n = t - b * a.nbins;
            const double d = hist[(a.nb + b) * a.nbins + n], w = hist[(2 * a.nb + b) * a.nbins + n];
            a.depi[((int64_t)jj * a.nbins + n) * a.nb + b] = (w != 0.0) ? __ddiv_rn(d, w) : 99999.0;
        }
    }
}

}  // namespace

extern "C" {

int cdft_mocsig_compute(const float *zv, const float *zveiv, const float *zt, const float *zs, const float *e1v,
                        const float *e3v, const float *e31d, const int16_t *ibmask, const float *gdep, int npiglo,
                        int npjglo, int npk, int nbasins, float zspv, float zspt, float zsps, float pref, int eos,
                        int nbins, float sigmin, float sigstp, int in_space, double *dmoc, double *depi, int out_space,
                        void *stream) {
    if (!zv || !zt || !zs || !e1v || (!e3v && !e31d) || !ibmask || !dmoc) return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: NULL argument");
    if (npiglo < 1 || npjglo < 1 || npk < 2 || nbasins < 1 || nbasins > 8 || nbins < 1 || nbins > 2048 || !(sigstp > 0.f))
        return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: bad sizes or binning (%d x %d x %d, %d basins, %d bins)", npiglo, npjglo,
                    npk, nbasins, nbins);
    if (depi && !gdep) return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: -isodep needs gdep");
    if (eos != CDFT_EOS_80 && eos != CDFT_EOS_TEOS10) return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: eos must be CDFT_EOS_80 or CDFT_EOS_TEOS10");
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const size_t nij = (size_t)npiglo * npjglo, n3 = nij * npk;
    const void *dv, *dei = nullptr, *dt, *ds, *d1, *de = nullptr, *d31 = nullptr, *dib, *dg = nullptr;
    CDFT_TRY(to_device(zv, n3 * 4, in_space, g_mi[0], st, &dv));
    if (zveiv) CDFT_TRY(to_device(zveiv, n3 * 4, in_space, g_mi[1], st, &dei));
    CDFT_TRY(to_device(zt, n3 * 4, in_space, g_mi[2], st, &dt));
    CDFT_TRY(to_device(zs, n3 * 4, in_space, g_mi[3], st, &ds));
    CDFT_TRY(to_device(e1v, nij * 4, in_space, g_mi[4], st, &d1));
    if (e3v) CDFT_TRY(to_device(e3v, n3 * 4, in_space, g_mi[5], st, &de));
    static DevBuf s31, sg;
    if (!e3v) CDFT_TRY(to_device(e31d, (size_t)npk * 4, in_space, s31, st, &d31));
    CDFT_TRY(to_device(ibmask, nij * nbasins * 2, in_space, g_mi[6], st, &dib));
    if (depi) CDFT_TRY(to_device(gdep, (size_t)npk * 4, CDFT_MEM_HOST, sg, st, &dg));       // gdep: npk HOST values
    const size_t ob = (size_t)nbasins * nbins * npjglo * 8;
    double *om = dmoc, *od = depi;
    if (out_space == CDFT_MEM_HOST) {
        CDFT_TRY(g_mo[0].reserve(ob)); om = (double *)g_mo[0].ptr;
        if (depi) { CDFT_TRY(g_mo[1].reserve(ob)); od = (double *)g_mo[1].ptr; }
    }
    SigArgs a;
    a.zv = (const float *)dv; a.zeiv = (const float *)dei; a.zt = (const float *)dt; a.zs = (const float *)ds;
    a.e1v = (const float *)d1; a.e3v = (const float *)de; a.e31d = (const float *)d31; a.gdep = (const float *)dg;
    a.ib = (const int16_t *)dib; a.npi = npiglo; a.npj = npjglo; a.npk = npk; a.nb = nbasins; a.nbins = nbins;
    a.j0 = npjglo > 1 ? 1 : 0; a.j1 = npjglo > 1 ? npjglo - 2 : 0;                         // :405-407
    a.zspv = zspv; a.zspt = zspt; a.zsps = zsps; a.pref = pref; a.sigmin = sigmin; a.sigstp = sigstp;
    a.dmoc = om; a.depi = od;
    const int nq = depi ? 3 : 1;
    const size_t per_wave = (size_t)nq * nbasins * nbins * 8;
    int nwaves = 4;
    while (nwaves > 1 && per_wave * nwaves > 60 * 1024) nwaves /= 2;
    if (per_wave * nwaves > 64 * 1024)
        return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: %d bins x %d basins do not fit the LDS histogram", nbins, nbasins);
    const dim3 grid(npjglo), block(64 * nwaves);
    const size_t shmem = per_wave * nwaves;
    if (eos == CDFT_EOS_TEOS10) {
        if (depi) hipLaunchKernelGGL((mocsig_row_kernel<1, true>), grid, block, shmem, st, a);
        else      hipLaunchKernelGGL((mocsig_row_kernel<1, false>), grid, block, shmem, st, a);
    } else {
        if (depi) hipLaunchKernelGGL((mocsig_row_kernel<0, true>), grid, block, shmem, st, a);
        else      hipLaunchKernelGGL((mocsig_row_kernel<0, false>), grid, block, shmem, st, a);
    }
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(dmoc, om, ob, hipMemcpyDeviceToHost, st));
        if (depi) CDFT_HIP(hipMemcpyAsync(depi, od, ob, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_mocsig_release(void) {
    for (DevBuf &b : g_mi) b.release();
    for (DevBuf &b : g_mo) b.release();
    return CDFT_OK;
}

}  // extern "C"
