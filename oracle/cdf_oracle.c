/*
 * cdf_oracle.c -- CPU restatement of the CDFTOOLS hot-path loops.  TEST INFRASTRUCTURE ONLY:
 * see cdf_oracle.h for who may use it and for the pinning status of each function.
 *
 * Build: gcc -O2 -ffp-contract=off [-fopenmp] (oracle/Makefile).  x86-64 baseline, no FMA
 * contraction, like the reference's plain -O/-O2 builds (Macrolib/macro.gfortran:26).
 */
#include "cdf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include "eos_coef.h"
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads of the following parallel loops (bench.py picks the count that serves the sample best) */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* Fortran (ji,jj[,jk]) -> C offset, 0-based indices */
#define IJ(i, j) ((size_t)(j) * npi + (i))
#define IJK(i, j, k) (((size_t)(k) * npj + (j)) * npi + (i))

/*
 * src/cdfmoc.f90:352      dmoc(:,:,:) = 0.d0
 * src/cdfmoc.f90:355-379  DO jk = 1, npk-1 ; DO jbasin ; !$OMP PARALLEL DO ; DO jj ; DO ji
 *                            dmoc(jbasin,jj,jk) = dmoc(jbasin,jj,jk) -
 *                                 e1v(ji,jj)*e3v(ji,jj,jk)*ibmask(jbasin,ji,jj)*zv(ji,jj)*1.d0
 *   (r4 product evaluated left to right, the INTEGER*2 mask converted to r4, then *1.d0 widens)
 * src/cdfmoc.f90:382-388  DO jj ; DO jk = npk-1, 1, -1
 *                            dmoc(:,jj,jk) = dmoc(:,jj,jk+1) + dmoc(:,jj,jk)/1.d6
 */
void oracle_moc(int npi, int npj, int npk, int nb, const float *e1v, const float *e3v,
                const int16_t *ibmask, const float *zv, double *dmoc) {
    memset(dmoc, 0, sizeof(double) * (size_t)nb * npj * npk);
    for (int jk = 0; jk < npk - 1; ++jk) {
        for (int jb = 0; jb < nb; ++jb) {
#pragma omp parallel for schedule(runtime)
            for (int jj = 0; jj < npj; ++jj) {
                double d = dmoc[((size_t)jk * npj + jj) * nb + jb];
                for (int ji = 0; ji < npi; ++ji) {
                    float p = e1v[IJ(ji, jj)] * e3v[IJK(ji, jj, jk)];
                    p = p * (float)ibmask[IJ(ji, jj) * nb + jb];
                    p = p * zv[IJK(ji, jj, jk)];
                    d = d - (double)p * 1.0;
                }
                dmoc[((size_t)jk * npj + jj) * nb + jb] = d;
            }
        }
    }
#pragma omp parallel for schedule(runtime)
    for (int jj = 0; jj < npj; ++jj)
        for (int jk = npk - 2; jk >= 0; --jk)
            for (int jb = 0; jb < nb; ++jb)
                dmoc[((size_t)jk * npj + jj) * nb + jb] =
                    dmoc[((size_t)(jk + 1) * npj + jj) * nb + jb] +
                    dmoc[((size_t)jk * npj + jj) * nb + jb] / 1.0e6;
}

/* ------------------------------------------------------------------------------------------ */
/* equation of state                                                                           */
/* ------------------------------------------------------------------------------------------ */
static const double eos_c[2][CDFT_EOS_NK][CDFT_EOS_NJ][CDFT_EOS_NI] = CDFT_EOS_COEF_INIT;
static const double eos_r[6] = CDFT_EOS_REF_INIT;

/* r_k(t,s): the reference nests Horner in t around Horner in s and writes each t-level as
 *   ( ... )*dlt + (((c_n*dls + c_{n-1})*dls + ...)+c_1)*dls + c_0        (src/eos.f90:858-875)
 * i.e. acc = (acc*t + Y*s) + c_0 with Y the Horner form of c_n..c_1; the top t-level is a bare
 * coefficient.  Reproduced with that exact association. */
static double eos_rk(const double c[CDFT_EOS_NJ][CDFT_EOS_NI], int k, double t, double s) {
    const int jmax = CDFT_EOS_JMAX(k);
    double acc = c[jmax][0];
    for (int j = jmax - 1; j >= 0; --j) {
        const int imax = CDFT_EOS_IMAX(k, j);
        double y = c[j][imax];
        for (int i = imax - 1; i >= 1; --i) y = y * s + c[j][i];
        acc = (acc * t + y * s) + c[j][0];
    }
    return acc;
}

void oracle_sigmai(const float *ptem, const float *psal, long n, float pref, const float *pref2d,
                   int teos10, double *sig) {
    const int set = teos10 ? 1 : 0;
    const double rdeltas = teos10 ? CDFT_EOS_RDELTAS_T10 : CDFT_EOS_RDELTAS_80;
    const double r1_s0 = teos10 ? CDFT_EOS_R1_S0_T10 : CDFT_EOS_R1_S0_80;
    const double r1_t0 = CDFT_EOS_R1_T0, r1_z0 = CDFT_EOS_R1_Z0, rau0 = CDFT_EOS_RAU0;
    for (long p = 0; p < n; ++p) {
        /* :842-845 / :1046,:1062: depth and reference density at that depth */
        const double h = (double)(pref2d ? pref2d[p] : pref) * r1_z0;
        const double ref = (((((eos_r[5] * h + eos_r[4]) * h + eos_r[3]) * h + eos_r[2]) * h + eos_r[1]) * h + eos_r[0]) * h;
        const double t = (double)ptem[p] * r1_t0;                              /* :850 */
        const double s = sqrt(fabs((double)psal[p] + rdeltas) * r1_s0);        /* :851 */
        const double tm = (psal[p] == 0.0f) ? 0.0 : 1.0;                       /* :852-856 */
        const double r3 = eos_rk(eos_c[set][3], 3, t, s);
        const double r2 = eos_rk(eos_c[set][2], 2, t, s);
        const double r1 = eos_rk(eos_c[set][1], 1, t, s);
        const double r0 = eos_rk(eos_c[set][0], 0, t, s);
        const double r = ((r3 * h + r2) * h + r1) * h + r0;                    /* :877 */
        sig[p] = (r + ref - rau0) * tm;                                        /* :880 */
    }
}

void oracle_sig_tool(const float *ptem, const float *psal, long n, float pref, float spval, int teos10,
                     float *sig4) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    oracle_sigmai(ptem, psal, n, pref, NULL, teos10, tmp);
    for (long p = 0; p < n; ++p) {
        const float zmask = (psal[p] == spval) ? 0.0f : 1.0f;                  /* cdfsig0.f90:148 */
        sig4[p] = (float)(tmp[p] * (double)zmask);                             /* :150 r8*r4 -> r4 */
    }
    free(tmp);
}

/* Jackett & McDougall (1994) form: src/eos.f90:889-1000 (one reference depth) and :1089-1201
 * (per-point depth).  Every polynomial in t is a plain Horner chain in the reference (:942-991);
 * coefficients highest power first. */
static double horner(const double *c, int n, double t) {
    double a = c[0];
    for (int i = 1; i < n; ++i) a = a * t + c[i];
    return a;
}

void oracle_sigmai_jmd94(const float *ptem, const float *psal, long n, float pref, const float *pref2d,
                         double *sig) {
    static const double c_r1[] = {6.536332e-9, -1.120083e-6, 1.001685e-4, -9.095290e-3, 6.793952e-2, 999.842594};
    static const double c_r2[] = {5.3875e-9, -8.2467e-7, 7.6438e-5, -4.0899e-3, 0.824493};
    static const double c_r3[] = {-1.6546e-6, 1.0227e-4, -5.72466e-3};
    static const double c_e[] = {-3.508914e-8, -1.248266e-8, -2.595994e-6};
    static const double c_bw[] = {1.296821e-6, -5.782165e-9, 1.045941e-4};
    static const double c_c[] = {-7.267926e-5, 2.598241e-3, 0.1571896};
    static const double c_aw[] = {5.939910e-6, 2.512549e-3, -0.1028859, -4.721788};
    static const double c_b1[] = {-0.1909078, 7.390729, -55.87545};
    static const double c_a1[] = {2.326469e-3, 1.553190, -65.00517, 1044.077};
    static const double c_kw[] = {-1.361629e-4, -1.852732e-2, -30.41638, 2098.925, 190925.6};
    const double dpr4 = 4.8314e-4, dpd = -2.042967e-2, dprau0 = 1000.0;       /* :920 */
    for (long p = 0; p < n; ++p) {
        const double ref = (double)(pref2d ? pref2d[p] : pref);                /* :928 */
        const double rs = (double)sqrtf(fabsf(psal[p]));    /* :931 SQRT(ABS(psal)) is evaluated in r4 */
        const double t = (double)ptem[p], s = (double)psal[p];                 /* :939-940 */
        const double r1 = horner(c_r1, 6, t), r2 = horner(c_r2, 5, t), r3 = horner(c_r3, 3, t);
        const double rhop = (dpr4 * s + r3 * rs + r2) * s + r1;                /* :960 */
        const double e = horner(c_e, 3, t), bw = horner(c_bw, 3, t);
        const double b = bw + e * s;                                           /* :969 */
        const double c = horner(c_c, 3, t), aw = horner(c_aw, 4, t);
        const double a = (dpd * rs + c) * s + aw;                              /* :978 */
        const double b1 = horner(c_b1, 3, t), a1 = horner(c_a1, 4, t), kw = horner(c_kw, 5, t);
        const double k0 = (b1 * rs + a1) * s + kw;                             /* :992 */
        sig[p] = rhop / (1.0 - ref / (k0 - ref * (a - ref * b))) - dprau0;     /* :995 */
    }
}

/* Neutral density, McDougall and Jackett (2005) as the reference writes it (src/eos.f90:634-687), each
 * product and sum in r8 in the written order; result rho - 1000, NOT masked.
 * Check value of the reference's comment (:646): rho(20,35) - 1000 = 24.59416751197. */
void oracle_sigmantr(const float *ptem, const float *psal, long n, double *out) {
    for (long p = 0; p < n; ++p) {
        const double t = (double)ptem[p], s = (double)psal[p], sr = sqrt(fabs(s));
        const double r1 = ((-4.3159255086706703e-4 * t + 8.1157118782170051e-2) * t + 2.2280832068441331e-1) * t + 1002.3063688892480;
        const double r2 = (-1.7052298331414675e-7 * s - 3.1710675488863952e-3 * t - 1.0304537539692924e-4) * s;
        const double r3 = (((-2.3850178558212048e-9 * t - 1.6212552470310961e-7) * t + 7.8717799560577725e-5) * t + 4.3907692647825900e-5) * t + 1.0;
        const double r4 = ((-2.2744455733317707e-9 * t * t + 6.0399864718597388e-6) * t - 5.1268124398160734e-4) * s;
        const double r5 = (-1.3409379420216683e-9 * t * t - 3.6138532339703262e-5) * s * sr;
        out[p] = (r1 + r2) / (r3 + r4 + r5) - 1000.0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* cdfmoc -decomp, src/cdfmoc.f90:353,:360-365 (dvbt, hdep in the main loop) and :390-517, ONE record.
 *   e1v,e1u,zcoef (npi,npj) r4; e3v (npi,npj,npk) r4 already * vmask(level); ibmask (nb,npi,npj) i2,
 *   basin 0 = global; zv,ztemp,zsal (npi,npj,npk) r4; tmask,umask (npi,npj,npk) 0/1 bytes (the
 *   reference reads them into INTEGER(2)); gdept (npk) r4.
 *   zcoef is the reference's r4 array after :428-433 (-g/rau0/f at V points, 0 where f = 0): it is
 *   computed by the CALLER because it goes through the compiler's r4 SIN.
 *   dmoc (nb,npj,npk): total MOC as oracle_moc gives it (input).
 *   dmoc_sh (nb,npj,npk): IN/OUT.  The reference never clears it (:471 accumulates onto whatever
 *   the array holds): zeros for the first record of a run, the previous record's result afterwards.
 *   dmoc_bt, dmoc_ag (nb,npj,npk): out.
 * Quirk kept: zv(:,:) is only assigned for ji=2..npiglo-1, jj=2..npjglo-1 (:447-462); its border
 * keeps the velocity of level npk-1 read last by the main loop (:358), and enters :466 and :471. */
void oracle_moc_decomp(int npi, int npj, int npk, int nb, const float *e1v, const float *e3v,
                       const int16_t *ibmask, const float *zv3, const float *ztemp, const float *zsal,
                       const uint8_t *tmask, const uint8_t *umask, const float *e1u, const float *zcoef,
                       const float *gdept, int teos10, const double *dmoc, double *dmoc_sh,
                       double *dmoc_bt, double *dmoc_ag) {
    const size_t nij = (size_t)npi * npj, nbj = (size_t)nb * npj;
    double *dvbt = (double *)calloc(nij, 8), *dvgeo = (double *)calloc(2 * nij, 8);
    double *btw = (double *)calloc(nbj * npk, 8), *sig8 = (double *)malloc(nij * 8);
    float *hdep = (float *)calloc(nij, 4), *zv = (float *)malloc(nij * 4), *zsig0 = (float *)malloc(nij * 4);
    /* :360-365 */
    for (int jk = 0; jk < npk - 1; ++jk)
        for (size_t p = 0; p < nij; ++p) {
            dvbt[p] = dvbt[p] + (double)(e3v[jk * nij + p] * zv3[jk * nij + p]) * 1.0;
            hdep[p] = hdep[p] + e3v[jk * nij + p];
        }
    memcpy(zv, zv3 + (size_t)(npk - 2) * nij, nij * 4);       /* what zv(:,:) holds when :390 is reached */
    /* :397-401 */
    for (size_t p = 0; p < nij; ++p) dvbt[p] = (hdep[p] != 0.f) ? dvbt[p] / (double)hdep[p] : 0.0;
    /* :403-419 */
    memset(dmoc_bt, 0, nbj * npk * 8);
    for (int jk = 0; jk < npk - 1; ++jk)
        for (int ji = 0; ji < npi; ++ji)
            for (int jb = 0; jb < nb; ++jb)
                for (int jj = 0; jj < npj; ++jj) {
                    const float w = e1v[IJ(ji, jj)] * e3v[IJK(ji, jj, jk)] * (float)ibmask[IJ(ji, jj) * nb + jb];
                    dmoc_bt[((size_t)jk * npj + jj) * nb + jb] -= (double)w * dvbt[IJ(ji, jj)];
                }
    for (int jk = npk - 2; jk >= 0; --jk)
        for (size_t q = 0; q < nbj; ++q) dmoc_bt[jk * nbj + q] = dmoc_bt[(jk + 1) * nbj + q] + dmoc_bt[jk * nbj + q] / 1.0e6;
    /* :435-482 */
    memset(dvbt, 0, nij * 8);
    int iup = 0, ido = 1;
    for (int jk = npk - 2; jk >= 0; --jk) {
        const uint8_t *ium = umask + jk * nij, *itm = tmask + jk * nij;
        oracle_sigmai(ztemp + jk * nij, zsal + jk * nij, (long)nij, gdept[jk], NULL, teos10, sig8);
        for (size_t p = 0; p < nij; ++p) zsig0[p] = (float)(sig8[p] * (double)(int16_t)itm[p]);          /* :443 */
        for (int jj = 1; jj < npj - 1; ++jj)
            for (int ji = 1; ji < npi - 1; ++ji) {
                int16_t isum = (int16_t)(ium[IJ(ji - 1, jj + 1)] + ium[IJ(ji, jj + 1)] + ium[IJ(ji - 1, jj)] + ium[IJ(ji, jj)]);
                if (isum < 1) isum = 1;
                const float zmsv = 1.f / (float)isum;                                                     /* :449 */
                float g = (zsig0[IJ(ji, jj + 1)] - zsig0[IJ(ji - 1, jj + 1)]) * (float)ium[IJ(ji - 1, jj + 1)] / e1u[IJ(ji - 1, jj + 1)];
                g = g + (zsig0[IJ(ji + 1, jj + 1)] - zsig0[IJ(ji, jj + 1)]) * (float)ium[IJ(ji, jj + 1)] / e1u[IJ(ji, jj + 1)];
                g = g + (zsig0[IJ(ji, jj)] - zsig0[IJ(ji - 1, jj)]) * (float)ium[IJ(ji - 1, jj)] / e1u[IJ(ji - 1, jj)];
                g = g + (zsig0[IJ(ji + 1, jj)] - zsig0[IJ(ji, jj)]) * (float)ium[IJ(ji, jj)] / e1u[IJ(ji, jj)];
                const double dgeo = (double)g * 1.0;                                                      /* :450-453 */
                double inc = (double)zcoef[IJ(ji, jj)] * dgeo;
                inc = inc * (double)zmsv;
                inc = inc * (double)ibmask[IJ(ji, jj) * nb + 0];
                inc = inc * (double)e3v[IJK(ji, jj, jk)];
                dvgeo[iup * nij + IJ(ji, jj)] = dvgeo[ido * nij + IJ(ji, jj)] + inc;                      /* :457 */
                zv[IJ(ji, jj)] = (float)(0.5 * (dvgeo[iup * nij + IJ(ji, jj)] + dvgeo[ido * nij + IJ(ji, jj)]));   /* :459 */
            }
        for (size_t p = 0; p < nij; ++p) dvbt[p] = dvbt[p] + (double)(e3v[jk * nij + p] * zv[p]) * 1.0;  /* :466 */
        for (int ji = 0; ji < npi; ++ji)
            for (int jb = 0; jb < nb; ++jb)
                for (int jj = 0; jj < npj; ++jj) {
                    float p = e1v[IJ(ji, jj)] * e3v[IJK(ji, jj, jk)];
                    p = p * (float)ibmask[IJ(ji, jj) * nb + jb];
                    p = p * zv[IJ(ji, jj)];
                    dmoc_sh[((size_t)jk * npj + jj) * nb + jb] -= (double)p * 1.0;                        /* :471 */
                }
        const int t = iup; iup = ido; ido = t;
    }
    for (size_t p = 0; p < nij; ++p) dvbt[p] = (hdep[p] != 0.f) ? dvbt[p] / (double)hdep[p] : 0.0;        /* :480-484 */
    /* :489-502 */
    for (int jk = 0; jk < npk - 1; ++jk)
        for (int ji = 0; ji < npi; ++ji)
            for (int jb = 0; jb < nb; ++jb)
                for (int jj = 0; jj < npj; ++jj) {
                    const float w = e1v[IJ(ji, jj)] * e3v[IJK(ji, jj, jk)] * (float)ibmask[IJ(ji, jj) * nb + jb];
                    btw[((size_t)jk * npj + jj) * nb + jb] -= (double)w * dvbt[IJ(ji, jj)];
                }
    for (size_t q = 0; q < nbj * npk; ++q) dmoc_sh[q] = dmoc_sh[q] - btw[q];                              /* :505 */
    for (int jk = npk - 2; jk >= 0; --jk)
        for (size_t q = 0; q < nbj; ++q) dmoc_sh[jk * nbj + q] = dmoc_sh[(jk + 1) * nbj + q] + dmoc_sh[jk * nbj + q] / (double)1.e6f;   /* :509 */
    for (size_t q = 0; q < nbj * npk; ++q) dmoc_ag[q] = dmoc[q] - dmoc_sh[q] - dmoc_bt[q];                /* :516 */
    free(dvbt); free(dvgeo); free(btw); free(sig8); free(hdep); free(zv); free(zsig0);
}

/* ------------------------------------------------------------------------------------------ */
/* cdfmocsig, src/cdfmocsig.f90:368-470 for ONE record (sigmai density, not -ntr).
 *   zv, zt, zs (npi,npj,npk) r4 as read (fill values -> 0 with zspv, zspt, zsps, :374,:382-383); zveiv or
 *   NULL (-eiv, :375-378); e1v (npi,npj); e3v (npi,npj,npk) or NULL with e31d (-full); ibmask (nb,npi,npj)
 *   i2; gdep (npk) the negative depths of :336, or NULL without -isodep.
 *   dmoc (nb,nbins,npj) r8; depi (nb,nbins,npj) r8 or NULL. */
void oracle_mocsig(int npi, int npj, int npk, int nb, const float *zv3, const float *zveiv, const float *zt3,
                   const float *zs3, const float *e1v, const float *e3v, const float *e31d, const int16_t *ibmask,
                   const float *gdep, float zspv, float zspt, float zsps, float pref, int teos10, int nbins,
                   float sigmin, float sigstp, double *dmoc, double *depi) {
    const size_t nij = (size_t)npi * npj, nout = (size_t)nb * nbins * npj;
    double *dens = (double *)malloc(nij * 8), *wdep = depi ? (double *)calloc(nout, 8) : NULL;
    double *tmp = (double *)malloc((size_t)nbins * npi * 8), *dtmp = NULL, *wtmp = NULL;
    float *zv = (float *)malloc(nij * 4), *zt = (float *)malloc(nij * 4), *zs = (float *)malloc(nij * 4);
    int *ibin = (int *)malloc(nij * sizeof(int)), *itm = (int *)malloc(nij * sizeof(int));
    if (depi) { dtmp = (double *)malloc((size_t)nbins * npi * 8); wtmp = (double *)malloc((size_t)nbins * npi * 8); memset(depi, 0, nout * 8); }
    memset(dmoc, 0, nout * 8);
    const int j0 = npj > 1 ? 1 : 0, j1 = npj > 1 ? npj - 2 : 0;                                    /* :405-407 */
    for (int jk = 0; jk < npk - 1; ++jk) {
        for (size_t p = 0; p < nij; ++p) {
            zv[p] = zv3[jk * nij + p]; if (zv[p] == zspv) zv[p] = 0.f;
            if (zveiv) zv[p] = zv[p] + zveiv[jk * nij + p];
            zt[p] = zt3[jk * nij + p]; if (zt[p] == zspt) zt[p] = 0.f;
            zs[p] = zs3[jk * nij + p]; if (zs[p] == zsps) zs[p] = 0.f;
            itm[p] = (zs[p] == zsps) ? 0 : 1;                                                      /* :392-393 */
        }
        if (teos10 == 2) oracle_sigmantr(zt, zs, (long)nij, dens);                  /* -ntr (:395) */
        else oracle_sigmai(zt, zs, (long)nij, pref, NULL, teos10, dens);
        for (size_t p = 0; p < nij; ++p) {
            const float zttmp = (float)(dens[p] * (double)itm[p]);                                 /* :398 */
            int ib = (int)((zttmp - sigmin) / sigstp);
            if (ib < 1) ib = 1;
            if (ib > nbins) ib = nbins;
            ibin[p] = ib - 1;
        }
        for (int jj = j0; jj <= j1; ++jj) {
            memset(tmp, 0, (size_t)nbins * npi * 8);
            if (depi) { memset(dtmp, 0, (size_t)nbins * npi * 8); memset(wtmp, 0, (size_t)nbins * npi * 8); }
            for (int ji = 1; ji < npi - 1; ++ji) {
                const size_t p = IJ(ji, jj);
                const float zarea = e1v[p] * (e3v ? e3v[IJK(ji, jj, jk)] : e31d[jk]);              /* :389 */
                const size_t q = (size_t)ibin[p] + (size_t)nbins * ji;
                tmp[q] = tmp[q] - (double)(zv[p] * zarea);                                         /* :417 */
                if (depi) {
                    dtmp[q] = dtmp[q] + (double)(gdep[jk] * (float)itm[p] * zarea);                /* :424 */
                    wtmp[q] = wtmp[q] + (double)((float)itm[p] * zarea);                           /* :425 */
                }
            }
            for (int jb = 0; jb < nb; ++jb)
                for (int jbin = 0; jbin < nbins; ++jbin)
                    for (int ji = 1; ji < npi - 1; ++ji) {
                        const double m = (double)ibmask[IJ(ji, jj) * nb + jb];
                        const size_t o = ((size_t)jj * nbins + jbin) * nb + jb, q = (size_t)jbin + (size_t)nbins * ji;
                        dmoc[o] = dmoc[o] + tmp[q] * m;                                            /* :435 */
                        if (depi) { depi[o] = depi[o] + dtmp[q] * m; wdep[o] = wdep[o] + wtmp[q] * m; }
                    }
        }
    }
    if (depi)
        for (size_t o = 0; o < nout; ++o) depi[o] = (wdep[o] != 0.0) ? depi[o] / wdep[o] : 99999.0;   /* :454-460 */
    for (int jj = 0; jj < npj; ++jj)
        for (int jb = 0; jb < nb; ++jb) {
            double *col = dmoc + (size_t)jj * nbins * nb + jb;
            col[(size_t)(nbins - 1) * nb] = col[(size_t)(nbins - 1) * nb] / (double)1.e6f;          /* :463 */
            for (int n = nbins - 2; n >= 0; --n) col[(size_t)n * nb] = col[(size_t)(n + 1) * nb] + col[(size_t)n * nb] / (double)1.e6f;
        }
    free(dens); free(tmp); free(zv); free(zt); free(zs); free(ibin); free(itm);
    if (depi) { free(dtmp); free(wtmp); free(wdep); }
}
