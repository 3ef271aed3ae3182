/*
 * cdf_oracle_tools.c -- CPU restatement of the cdfheatc / cdfmhst / cdfpsi / cdfvT / cdfcurl loops.
 * TEST INFRASTRUCTURE ONLY (see cdf_oracle.h).  These five are pinned bit for bit to harnesses built from the
 * reference's own statements (oracle/ref_lift.py, tests/golden/lift_*.npz, tests/test_lift.py); each function follows
 * the cited source line by line, in its evaluation order and r4/r8 mixing.
 */
#include "cdf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IJ(i, j) ((size_t)(j) * npi + (i))
#define IJK(i, j, k) (((size_t)(k) * npj + (j)) * npi + (i))

/* ------------------------------------------------------------------------------------------ */
/* src/cdfheatc.f90:205-230, all levels of one record.
 *   e3t: (npi,npj,npk) or NULL with e31d(npk) (-full, :213); tmask r4 (npi,npj,npk);
 *   mxlopt 0/1/-1 with mxl(npi,npj), gdepw(npk) (:218-223).
 *   r4sum = 1: the array expression AND the SUM in r4, sequentially (what flang -O1/-O2 inlines for
 *              the reference; compiler dependent);
 *   r4sum = 0: same r4 products, r8 sequential accumulation (comparator of the HIP path).
 *   sums(3,npk) = dsurf, dvol2d, dsum2d. */
void oracle_heatc(int npi, int npj, int npk, const float *e1t, const float *e2t, const float *e3t,
                  const float *e31d, const float *tmask, const float *temp, const float *mxl,
                  const float *gdepw, int mxlopt, int r4sum, double *sums) {
    const size_t nij = (size_t)npi * npj;
    for (int k = 0; k < npk; ++k) {
        double d_surf = 0.0, d_vol = 0.0, d_sum = 0.0;
        float f_surf = 0.f, f_vol = 0.f, f_sum = 0.f;
        for (size_t p = 0; p < nij; ++p) {
            float e3 = e3t ? e3t[(size_t)k * nij + p] : e31d[k];
            if (mxlopt == 1) e3 = fmaxf(0.f, fminf(e3, mxl[p] - gdepw[k]));                 /* :220 */
            else if (mxlopt == -1) e3 = fminf(e3, fmaxf(0.f, gdepw[k] + e3 - mxl[p]));      /* :222 */
            const float m = tmask[(size_t)k * nij + p];
            const float a = e1t[p] * e2t[p];
            const float ps = a * m;                                                         /* :225 */
            const float pv = a * e3 * m;                                                    /* :226 */
            const float pt = a * e3 * temp[(size_t)k * nij + p] * m;                        /* :229 */
            if (r4sum) { f_surf += ps; f_vol += pv; f_sum += pt; }
            else { d_surf += (double)ps; d_vol += (double)pv; d_sum += (double)pt; }
        }
        sums[3 * k + 0] = r4sum ? (double)f_surf : d_surf;
        sums[3 * k + 1] = r4sum ? (double)f_vol : d_vol;
        sums[3 * k + 2] = r4sum ? (double)f_sum : d_sum;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/cdfmhst.f90:306-367 for one record.
 *   vt, vs (npi,npj,npk) r4 (-vt file, :325-326); or vt == NULL and zv, zt, zs given (:312-322):
 *       zvt(ji,jj) = 0.5*(zt(ji,jj)+zt(ji,jj+1))*zv(ji,jj) for jj < npj, 0 on the last row;
 *   e3v (npi,npj,npk) or NULL with e31d (-full, :329); e1v (npi,npj);
 *   masks (npi,npj,nb) r4: basin 0 is the global one (vmask level 1) and is summed over
 *       ji = 2..npi-1 only (:343-344), the others over the whole row (:349-367);
 *   zdim = 0: heat/salt (nb,npj) after the last level; zdim = 1: (nb,npj,npk) running totals
 *       after every level (:389-392, -Zdim).
 *   heat = sum dtrph*zmask with dtrph += (zvt*e1v*e3v*1.d0)*pprau0*pprcp (:332,:336),
 *   salt = sum dtrps*zmask with dtrps += zvs*e1v*e3v*1.d0 (:333,:337). */
void oracle_mhst(int npi, int npj, int npk, int nb, const float *e1v, const float *e3v, const float *e31d,
                 const float *vt, const float *vs, const float *zv, const float *zt, const float *zs,
                 const float *masks, int zdim, double *heat, double *salt) {
    const size_t nij = (size_t)npi * npj;
    const float pprau0 = 1000.f, pprcp = 4000.f;                  /* src/cdfmhst.f90:53-54 */
    double *dtrph = (double *)calloc(nij, sizeof(double));
    double *dtrps = (double *)calloc(nij, sizeof(double));
    for (int k = 0; k < npk; ++k) {
        for (int j = 0; j < npj; ++j)
            for (int i = 0; i < npi; ++i) {
                float zvt, zvs;
                if (vt) {
                    zvt = vt[IJK(i, j, k)];
                    zvs = vs[IJK(i, j, k)];
                } else if (j < npj - 1) {
                    zvt = 0.5f * (zt[IJK(i, j, k)] + zt[IJK(i, j + 1, k)]) * zv[IJK(i, j, k)];   /* :318 */
                    zvs = 0.5f * (zs[IJK(i, j, k)] + zs[IJK(i, j + 1, k)]) * zv[IJK(i, j, k)];
                } else {
                    zvt = 0.f; zvs = 0.f;                                                        /* :314-315 */
                }
                const float e3 = e3v ? e3v[IJK(i, j, k)] : e31d[k];
                const double dwkh = (double)(zvt * e1v[IJ(i, j)] * e3) * 1.0;                    /* :332 */
                const double dwks = (double)(zvs * e1v[IJ(i, j)] * e3) * 1.0;                    /* :333 */
                dtrph[IJ(i, j)] = dtrph[IJ(i, j)] + dwkh * (double)pprau0 * (double)pprcp;       /* :336 */
                dtrps[IJ(i, j)] = dtrps[IJ(i, j)] + dwks;                                        /* :337 */
            }
        if (zdim || k == npk - 1) {
            double *oh = heat + (zdim ? (size_t)k * npj * nb : 0);
            double *os = salt + (zdim ? (size_t)k * npj * nb : 0);
            for (int b = 0; b < nb; ++b) {
                const float *zm = masks + (size_t)b * nij;
                const int i0 = (b == 0) ? 1 : 0, i1 = (b == 0) ? npi - 1 : npi;                  /* :343 */
                for (int j = 0; j < npj; ++j) {
                    double sh = 0.0, ss = 0.0;
                    for (int i = i0; i < i1; ++i) {
                        sh += dtrph[IJ(i, j)] * (double)zm[IJ(i, j)];
                        ss += dtrps[IJ(i, j)] * (double)zm[IJ(i, j)];
                    }
                    oh[(size_t)j * nb + b] = sh;
                    os[(size_t)j * nb + b] = ss;
                }
            }
        }
    }
    free(dtrph);
    free(dtrps);
}

/* ------------------------------------------------------------------------------------------ */
/* src/cdfpsi.f90:256-438 for one record, default / -V / -mean / -mask / -ref paths (-open: see
 * oracle_psi_open below; --ssh-file and -lev are these same functions applied to other operands).
 *   ll_u: zu, e2u, e3u (or e31d) -> dtrpu = sum_k zu*e2u*e3u*1.d0 (:296-297), then
 *         dpsiu(:,1) = 0 ; dpsiu(:,jj) = dpsiu(:,jj-1) - dtrpu(:,jj), jj = 2..npj (:395-398)
 *   ll_v: zv, e1v, e3v          -> dtrpv = sum_k zv*e1v*e3v*1.d0 (:277-278), then
 *         dpsiv(npi,:) = 0 ; dpsiv(ji,:) = dpsiv(ji+1,:) - dtrpv(ji,:), ji = npi-1..1 (:380-383)
 *   both (-mean): dpsi = 0.5*(dpsiu + dpsiv) (:413)
 *   dpsi = dpsi - dpsi(iiref,ijref) (:423, 1-based reference point)
 *   psi4 = SNGL(dpsi) [* fmask] (:426,:433); dpsi8 (optional) = the r8 field before SNGL. */
void oracle_psi(int npi, int npj, int npk, const float *zu, const float *e2u, const float *e3u,
                const float *zv, const float *e1v, const float *e3v, const float *e31d, int iiref,
                int ijref, const float *fmask, float *psi4, double *dpsi8) {
    const size_t nij = (size_t)npi * npj;
    double *dtrpu = (double *)calloc(nij, sizeof(double)), *dtrpv = (double *)calloc(nij, sizeof(double));
    double *dpsiu = (double *)calloc(nij, sizeof(double)), *dpsiv = (double *)calloc(nij, sizeof(double));
    for (int k = 0; k < npk; ++k)
        for (int j = 0; j < npj; ++j)
            for (int i = 0; i < npi; ++i) {
                if (zv) {
                    const float e3 = e3v ? e3v[IJK(i, j, k)] : e31d[k];
                    dtrpv[IJ(i, j)] = dtrpv[IJ(i, j)] + (double)(zv[IJK(i, j, k)] * e1v[IJ(i, j)] * e3) * 1.0;
                }
                if (zu) {
                    const float e3 = e3u ? e3u[IJK(i, j, k)] : e31d[k];
                    dtrpu[IJ(i, j)] = dtrpu[IJ(i, j)] + (double)(zu[IJK(i, j, k)] * e2u[IJ(i, j)] * e3) * 1.0;
                }
            }
    double *dpsi = NULL;
    if (zv) {
        for (int j = 0; j < npj; ++j) {
            dpsiv[IJ(npi - 1, j)] = 0.0;
            for (int i = npi - 2; i >= 0; --i) dpsiv[IJ(i, j)] = dpsiv[IJ(i + 1, j)] - dtrpv[IJ(i, j)];
        }
        dpsi = dpsiv;
    }
    if (zu) {
        for (int j = 1; j < npj; ++j)
            for (int i = 0; i < npi; ++i) dpsiu[IJ(i, j)] = dpsiu[IJ(i, j - 1)] - dtrpu[IJ(i, j)];
        dpsi = dpsiu;
    }
    if (zu && zv) {
        for (size_t p = 0; p < nij; ++p) dpsiu[p] = 0.5 * (dpsiu[p] + dpsiv[p]);
        dpsi = dpsiu;
    }
    const double ref = dpsi[IJ(iiref - 1, ijref - 1)];
    for (size_t p = 0; p < nij; ++p) {
        const double d = dpsi[p] - ref;
        if (dpsi8) dpsi8[p] = d;
        psi4[p] = fmask ? (float)d * fmask[p] : (float)d;
    }
    free(dtrpu); free(dtrpv); free(dpsiu); free(dpsiv);
}

/* src/cdfpsi.f90:316-366, the -open path (implies both transports): psi from a first line inside the
 * domain instead of a closed boundary.
 *   dpsiu(1,J) = dtrpv(1,J), dpsiu(ji,J) = dpsiu(ji-1,J) + dtrpv(ji,J) on the row J = npjglo-2 (:335-338),
 *   then southwards dpsiu(ji,jj) = dpsiu(ji,jj+1) + dtrpu(ji,jj+1), jj = npjglo-3..1 (:340-344); rows
 *   north of J keep the 0 of :259.
 *   lmean (:346-356): the other estimate from the column I = npiglo-2,
 *   dpsiv(I,npjglo) = dtrpu(I,npjglo), dpsiv(I,jj) = dpsiv(I,jj+1) + dtrpu(I,jj+1), jj = npjglo-1..1, then
 *   westwards dpsiv(ji,jj) = dpsiv(ji+1,jj) - dtrpv(ji+1,jj), ji = npiglo-3..1; columns east of I stay 0;
 *   dpsi = 0.5*(dpsiu + dpsiv).
 *   Then dpsi = dpsi - dpsi(iiref,ijref) unless noref (the ssh part is never normalised, :423),
 *   psi4 = SNGL(dpsi) [* fmask]. */
void oracle_psi_open(int npi, int npj, int npk, const float *zu, const float *e2u, const float *e3u,
                     const float *zv, const float *e1v, const float *e3v, const float *e31d, int lmean,
                     int iiref, int ijref, int noref, const float *fmask, float *psi4, double *dpsi8) {
    const size_t nij = (size_t)npi * npj;
    double *dtrpu = (double *)calloc(nij, 8), *dtrpv = (double *)calloc(nij, 8);
    double *dpsiu = (double *)calloc(nij, 8), *dpsiv = (double *)calloc(nij, 8);
    for (int k = 0; k < npk; ++k)
        for (int j = 0; j < npj; ++j)
            for (int i = 0; i < npi; ++i) {
                const float ev = e3v ? e3v[IJK(i, j, k)] : e31d[k], eu = e3u ? e3u[IJK(i, j, k)] : e31d[k];
                dtrpv[IJ(i, j)] = dtrpv[IJ(i, j)] + (double)(zv[IJK(i, j, k)] * e1v[IJ(i, j)] * ev) * 1.0;
                dtrpu[IJ(i, j)] = dtrpu[IJ(i, j)] + (double)(zu[IJK(i, j, k)] * e2u[IJ(i, j)] * eu) * 1.0;
            }
    const int J = npj - 3;                                   /* row npjglo-2, 0-based */
    dpsiu[IJ(0, J)] = dtrpv[IJ(0, J)];
    for (int i = 1; i < npi; ++i) dpsiu[IJ(i, J)] = dpsiu[IJ(i - 1, J)] + dtrpv[IJ(i, J)];
    for (int j = J - 1; j >= 0; --j)
        for (int i = 0; i < npi; ++i) dpsiu[IJ(i, j)] = dpsiu[IJ(i, j + 1)] + dtrpu[IJ(i, j + 1)];
    if (lmean) {
        const int I = npi - 3;                               /* column npiglo-2, 0-based */
        dpsiv[IJ(I, npj - 1)] = dtrpu[IJ(I, npj - 1)];
        for (int j = npj - 2; j >= 0; --j) dpsiv[IJ(I, j)] = dpsiv[IJ(I, j + 1)] + dtrpu[IJ(I, j + 1)];
        for (int j = npj - 1; j >= 0; --j)
            for (int i = I - 1; i >= 0; --i) dpsiv[IJ(i, j)] = dpsiv[IJ(i + 1, j)] - dtrpv[IJ(i + 1, j)];
        for (size_t p = 0; p < nij; ++p) dpsiu[p] = 0.5 * (dpsiu[p] + dpsiv[p]);
    }
    const double ref = noref ? 0.0 : dpsiu[IJ(iiref - 1, ijref - 1)];
    for (size_t p = 0; p < nij; ++p) {
        const double d = noref ? dpsiu[p] : dpsiu[p] - ref;
        if (dpsi8) dpsi8[p] = d;
        psi4[p] = fmask ? (float)d * fmask[p] : (float)d;
    }
    free(dtrpu); free(dtrpv); free(dpsiu); free(dpsiv);
}

/* ------------------------------------------------------------------------------------------ */
/* src/cdfvT.f90:160-259 for ONE level over nfr time frames (all tags concatenated).
 *   u, v, t, s: frame f of the level at ptr + f*fstride (npi,npj) r4; e3u/e3v likewise with -vvl
 *   (else NULL).  Outputs (npi,npj) r8: vt, vs, ut, us = the values handed to putvar (:247-250,
 *   or :240-245 with -vvl, then also e3um, e3vm). */
void oracle_vt(int npi, int npj, int nfr, long fstride, const float *u, const float *v, const float *t,
               const float *s, const float *e3u, const float *e3v, double *vt, double *vs, double *ut,
               double *us, double *e3um, double *e3vm) {
    const size_t nij = (size_t)npi * npj;
    double *cut = (double *)calloc(nij, 8), *cvt = (double *)calloc(nij, 8), *cus = (double *)calloc(nij, 8),
           *cvs = (double *)calloc(nij, 8), *ce3u = (double *)calloc(nij, 8), *ce3v = (double *)calloc(nij, 8);
    float *wu = (float *)malloc(nij * 4), *wv = (float *)malloc(nij * 4);
    for (int f = 0; f < nfr; ++f) {
        const float *zu = u + (size_t)f * fstride, *zv = v + (size_t)f * fstride;
        const float *e3uf = e3u ? e3u + (size_t)f * fstride : NULL, *e3vf = e3v ? e3v + (size_t)f * fstride : NULL;
        for (int pass = 0; pass < 2; ++pass) {                       /* temperature, then salinity */
            const float *sc = (pass ? s : t) + (size_t)f * fstride;
            double *au = pass ? cus : cut, *av = pass ? cvs : cvt;
            memset(wu, 0, nij * 4); memset(wv, 0, nij * 4);                                     /* :201 */
            for (int i = 0; i < npi - 1; ++i)
                for (int j = 0; j < npj - 1; ++j) {
                    wu[IJ(i, j)] = 0.5f * (sc[IJ(i, j)] + sc[IJ(i + 1, j)]);                      /* :204 */
                    wv[IJ(i, j)] = 0.5f * (sc[IJ(i, j)] + sc[IJ(i, j + 1)]);                      /* :205 */
                }
            for (size_t p = 0; p < nij; ++p) {
                if (e3uf) {
                    au[p] = au[p] + (double)(wu[p] * e3uf[p] * zu[p]) * 1.0;                      /* :209 */
                    av[p] = av[p] + (double)(wv[p] * e3vf[p] * zv[p]) * 1.0;
                } else {
                    au[p] = au[p] + (double)(wu[p] * zu[p]) * 1.0;                                /* :214 */
                    av[p] = av[p] + (double)(wv[p] * zv[p]) * 1.0;
                }
            }
        }
        if (e3uf)
            for (size_t p = 0; p < nij; ++p) {
                ce3u[p] = ce3u[p] + (double)e3uf[p] * 1.0;                                        /* :211 */
                ce3v[p] = ce3v[p] + (double)e3vf[p] * 1.0;
            }
    }
    for (size_t p = 0; p < nij; ++p) {
        if (e3u) {                                                                               /* :240-245 */
            vt[p] = cvt[p] / ce3v[p] * 1.0; vs[p] = cvs[p] / ce3v[p] * 1.0;
            ut[p] = cut[p] / ce3u[p] * 1.0; us[p] = cus[p] / ce3u[p] * 1.0;
            e3um[p] = ce3u[p] / (double)nfr * 1.0; e3vm[p] = ce3v[p] / (double)nfr * 1.0;
        } else {                                                                                 /* :247-250 */
            vt[p] = cvt[p] / (double)nfr * 1.0; vs[p] = cvs[p] / (double)nfr * 1.0;
            ut[p] = cut[p] / (double)nfr * 1.0; us[p] = cus[p] / (double)nfr * 1.0;
        }
    }
    free(cut); free(cvt); free(cus); free(cvs); free(ce3u); free(ce3v); free(wu); free(wv);
}

/* ------------------------------------------------------------------------------------------ */
/* src/cdfcurl.f90:262-301 for one level.
 *   un, vn (npi,npj) r4 (C-grid velocities, :256-257); e1u, e2v, e1f, e2f (npi,npj) r4;
 *   fmask = 1 where un(i,j)*un(i,j+1)*vn(i,j)*vn(i+1,j) /= 0 (:267-273);
 *   rotn(i,j) = (e2v(i+1,j)vn(i+1,j) - e2v(i,j)vn(i,j) - e1u(i,j+1)un(i,j+1) + e1u(i,j)un(i,j))
 *               * fmask / (e1f*e2f)   -- one r4 expression stored into r8 (:276-282), i<npi, j<npj;
 *   lperio: rotn(npi,:) = rotn(2,:) (:285);
 *   ltpoint (-T): 4-point mean to T points with the fmask product (:286-297); the reference never
 *       assigns fmask(npi,:) and fmask(:,npj), they are taken as 0 here;
 *   ff (npi,npj) r8 or NULL: -overf, rotn = rotn/ff where ff /= 0 (:299-301). */
void oracle_curl(int npi, int npj, const float *un, const float *vn, const float *e1u, const float *e2v,
                 const float *e1f, const float *e2f, int lperio, int ltpoint, const double *ff,
                 double *rotn) {
    const size_t nij = (size_t)npi * npj;
    float *fm = (float *)calloc(nij, 4);
    memset(rotn, 0, nij * 8);
    for (int j = 0; j < npj - 1; ++j)
        for (int i = 0; i < npi - 1; ++i) {
            float f = un[IJ(i, j)] * un[IJ(i, j + 1)] * vn[IJ(i, j)] * vn[IJ(i + 1, j)];
            if (f != 0.f) f = 1.f;
            fm[IJ(i, j)] = f;
            const float r = (e2v[IJ(i + 1, j)] * vn[IJ(i + 1, j)] - e2v[IJ(i, j)] * vn[IJ(i, j)] -
                             e1u[IJ(i, j + 1)] * un[IJ(i, j + 1)] + e1u[IJ(i, j)] * un[IJ(i, j)]) *
                            f / (e1f[IJ(i, j)] * e2f[IJ(i, j)]);
            rotn[IJ(i, j)] = (double)r;
        }
    if (lperio && npi > 2)   /* column 2 exists and is not the wrap column itself */
        for (int j = 0; j < npj; ++j) rotn[IJ(npi - 1, j)] = rotn[IJ(1, j)];
    if (ltpoint) {
        double *dl = (double *)calloc(nij, 8);
        for (int i = 1; i < npi; ++i)
            for (int j = 1; j < npj; ++j) {
                const float zm = fm[IJ(i, j)] * fm[IJ(i, j - 1)] * fm[IJ(i - 1, j)] * fm[IJ(i - 1, j - 1)];
                dl[IJ(i, j)] = 0.25 * (rotn[IJ(i, j)] + rotn[IJ(i, j - 1)] + rotn[IJ(i - 1, j)] + rotn[IJ(i - 1, j - 1)]) * (double)zm;
            }
        if (lperio)
            for (int j = 0; j < npj; ++j) dl[IJ(0, j)] = dl[IJ(npi - 1, j)];
        memcpy(rotn, dl, nij * 8);
        free(dl);
    }
    if (ff)
        for (size_t p = 0; p < nij; ++p)
            if (ff[p] != 0.0) rotn[p] = rotn[p] / ff[p];
    free(fm);
}

/* ------------------------------------------------------------------------------------------ */
/* src/cdfvtrp.f90:211-218: dtrp = sum_k vel*m2d*e3*1.d0 (the product in r4, left to right) */
void oracle_vint_transport(int npi, int npj, int npk, const float *vel, const float *m2d, const float *e3,
                           const float *e31d, double *dtrp) {
    const size_t nij = (size_t)npi * npj;
    memset(dtrp, 0, nij * 8);
    for (int k = 0; k < npk; ++k)
        for (size_t p = 0; p < nij; ++p) {
            const float e = e3 ? e3[k * nij + p] : e31d[k];
            dtrp[p] = dtrp[p] + (double)(vel[k * nij + p] * m2d[p] * e) * 1.0;
        }
}

/* src/cdfvtrp.f90:210-252 (-bathy): the transports brought to the T point, rotated onto the bathymetric slope.
 *   dwku(i,j) = 0.5*(dtrpu(i,j) + dtrpu(i-1,j)), i >= 2; dwku(1,j) = dwku(npi-1,j)          (:213-219)
 *   dwkv(i,j) = 0.5*(dtrpv(i,j) + dtrpv(i,j-1)), j >= 2; row 1 = 0                           (:220-225)
 *   zdhdx = (hdepw(i+1,j) - hdepw(i-1,j)) / (e1u(i,j) + e1u(i-1,j)) * tmask, 2 <= i <= npi-1, else 0   (r4, :228-233)
 *   zdhdy = (hdepw(i,j+1) - hdepw(i,j-1)) / (e2v(i,j) + e2v(i,j-1)) * tmask, 2 <= j <= npj-1, else 0   (r4, :235-240)
 *   zalpha = ATAN2(zdhdx, zdhdy) * tmask                                                      (r4, :243)
 *   along = ( dwku*COS(zalpha) + dwkv*SIN(zalpha)) * tmask ; cross = (-dwku*SIN(zalpha) + dwkv*COS(zalpha)) * tmask
 * (r4 COS / SIN widened, r8 arithmetic, :246-247).  ATAN2 / COS / SIN are the C library's single precision functions. */
void oracle_vtrp_slope(int npi, int npj, const double *dtrpu, const double *dtrpv, const float *hdepw, const float *e1u,
                       const float *e2v, const float *tmask, double *along, double *cross) {
#define AT(a, i, j) (a)[(size_t)(j) * npi + (i)]
    const size_t nij = (size_t)npi * npj;
    double *dwku = (double *)calloc(nij, 8), *dwkv = (double *)calloc(nij, 8);
    for (int j = 0; j < npj; ++j) {
        for (int i = 1; i < npi; ++i) AT(dwku, i, j) = 0.5f * (AT(dtrpu, i, j) + AT(dtrpu, i - 1, j));
        if (npi >= 2) AT(dwku, 0, j) = AT(dwku, npi - 2, j);
    }
    for (int j = 1; j < npj; ++j)
        for (int i = 0; i < npi; ++i) AT(dwkv, i, j) = 0.5f * (AT(dtrpv, i, j) + AT(dtrpv, i, j - 1));
    for (int j = 0; j < npj; ++j)
        for (int i = 0; i < npi; ++i) {
            float zdhdx = 0.f, zdhdy = 0.f;
            const float tm = AT(tmask, i, j);
            if (i >= 1 && i <= npi - 2) zdhdx = (AT(hdepw, i + 1, j) - AT(hdepw, i - 1, j)) / (AT(e1u, i, j) + AT(e1u, i - 1, j)) * tm;
            if (j >= 1 && j <= npj - 2) zdhdy = (AT(hdepw, i, j + 1) - AT(hdepw, i, j - 1)) / (AT(e2v, i, j) + AT(e2v, i, j - 1)) * tm;
            const float zalpha = atan2f(zdhdx, zdhdy) * tm;
            const double c = (double)cosf(zalpha), sn = (double)sinf(zalpha);
            AT(along, i, j) = (AT(dwku, i, j) * c + AT(dwkv, i, j) * sn) * (double)tm;
            AT(cross, i, j) = (-AT(dwku, i, j) * sn + AT(dwkv, i, j) * c) * (double)tm;
        }
    free(dwku); free(dwkv);
#undef AT
}

/* src/cdfvhst.f90:167-178: e3m = e3*m2d (r4, :170-171), then
 *   dheat += xt*e3m*pp_rau0*pp_rcp*1.d0 ; dsalt += xs*e3m*1.d0   (pp_rau0 = 1000., pp_rcp = 4000., r4) */
void oracle_vhst_transport(int npi, int npj, int npk, const float *xt, const float *xs, const float *m2d,
                           const float *e3, const float *e31d, double *dheat, double *dsalt) {
    const size_t nij = (size_t)npi * npj;
    const float pp_rau0 = 1000.f, pp_rcp = 4000.f;
    memset(dheat, 0, nij * 8);
    memset(dsalt, 0, nij * 8);
    for (int k = 0; k < npk; ++k)
        for (size_t p = 0; p < nij; ++p) {
            const float e3m = (e3 ? e3[k * nij + p] : e31d[k]) * m2d[p];
            dheat[p] = dheat[p] + (double)(xt[k * nij + p] * e3m * pp_rau0 * pp_rcp) * 1.0;
            dsalt[p] = dsalt[p] + (double)(xs[k * nij + p] * e3m) * 1.0;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* src/cdfmean.f90:405-431 for one record: zv = zv*zmask (:410), then per level and basin
 *   dsurf  = SUM(DBLE(e1*e2*zmask*ibmask)), dvol2d = SUM(DBLE(e1*e2*e3*zmask*ibmask)),
 *   dsum2d = SUM(DBLE(zv*e1*e2*e3*ibmask)), dvar2d = SUM(DBLE(zv*zv*e1*e2*e3*ibmask))
 * with every product in r4, left to right (ibmask is INTEGER(2): converted to r4 by the multiply).
 * sums (4,nb,npk).  The SUM is taken sequentially here (its order is the compiler's in the reference). */
void oracle_mean_form(int npi, int npj, int npk, int nb, const float *e1, const float *e2, const float *e3,
                      const float *e31d, const float *zmask, const int16_t *ibmask, const float *zv, int form,
                      double *sums);

void oracle_mean(int npi, int npj, int npk, int nb, const float *e1, const float *e2, const float *e3,
                 const float *e31d, const float *zmask, const int16_t *ibmask, const float *zv, double *sums) {
    oracle_mean_form(npi, npj, npk, nb, e1, e2, e3, e31d, zmask, ibmask, zv, 0, sums);
}

/* form 0: src/cdfmean.f90:415-431.  form 1: src/cdfsum.f90:306-309 (dsum2d = SUM(DBLE(zv*e1*e2)), no variance);
 * form 2: src/cdfsum.f90:319-320, a file without depth (dsum2d = SUM(DBLE(zv*e1*e2*zmask))).  zv = zv*zmask first. */
void oracle_mean_form(int npi, int npj, int npk, int nb, const float *e1, const float *e2, const float *e3,
                      const float *e31d, const float *zmask, const int16_t *ibmask, const float *zv, int form,
                      double *sums) {
    const size_t nij = (size_t)npi * npj;
    for (int k = 0; k < npk; ++k)
        for (int b = 0; b < nb; ++b) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (size_t p = 0; p < nij; ++p) {
                const float m = zmask[k * nij + p], fb = (float)ibmask[p * nb + b];
                const float e = e3 ? e3[k * nij + p] : e31d[k];
                const float z = zv[k * nij + p] * m;
                s0 += (double)(e1[p] * e2[p] * m * fb);
                s1 += (double)(e1[p] * e2[p] * e * m * fb);
                if (form == 0) {
                    s2 += (double)(z * e1[p] * e2[p] * e * fb);
                    s3 += (double)(z * z * e1[p] * e2[p] * e * fb);
                } else if (form == 1) {
                    s2 += (double)(z * e1[p] * e2[p] * fb);
                } else {
                    s2 += (double)(z * e1[p] * e2[p] * m * fb);
                }
            }
            double *o = sums + ((size_t)k * nb + b) * 4;
            o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* src/cdfzonalsum.f90:296-313 (mean_form = 0) and src/cdfzonalmean.f90:313-330 (mean_form = 1) for one
 * variable and record of nlev levels.  e1,e2 (npi,npj); zmaskvar (npi,npj,nlev); zmask (nb,npi,npj) r4;
 * outputs (npj,nlev,nb): dzo, darea (mean form) r8, rmax / rmin r4 (mean form, may be NULL).
 *   sum  form: dtmp = zmask*zmaskvar*zv*1.d0 (r4 product) ; dzo += dl_surf*dtmp
 *   mean form: dtmp = 1.d0*zmask*zmaskvar*zv (r8 products); dzo += dl_surf*dtmp ;
 *              darea += dl_surf*zmask*zmaskvar ; rzomax = MAX(rzomax,dtmp) ; dtmp /= 0: rzomin = MIN(rzomin,dtmp)
 *   dl_surf = 1.d0*e1*e2 */
void oracle_zonal(int npi, int npj, int nlev, int nb, const float *e1, const float *e2, const float *zmaskvar,
                  const float *zmask, const float *zv, int mean_form, double *dzo, double *darea, float *rmax,
                  float *rmin) {
    const size_t nij = (size_t)npi * npj;
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < nlev; ++k)
            for (int j = 0; j < npj; ++j) {
                double s = 0.0, a = 0.0;
                float mx = -1.e20f, mn = 1.e20f;
                for (int i = 0; i < npi; ++i) {
                    const size_t p = IJ(i, j);
                    const float zb = zmask[p * nb + b], m = zmaskvar[k * nij + p], z = zv[k * nij + p];
                    const double surf = 1.0 * (double)e1[p] * (double)e2[p];
                    const double dtmp = mean_form ? 1.0 * (double)zb * (double)m * (double)z : (double)(zb * m * z) * 1.0;
                    s = s + surf * dtmp;
                    if (mean_form) {
                        a = a + surf * (double)zb * (double)m;
                        mx = (float)((double)mx > dtmp ? (double)mx : dtmp);
                        if (dtmp != 0.0) mn = (float)((double)mn < dtmp ? (double)mn : dtmp);
                    }
                }
                const size_t o = ((size_t)b * nlev + k) * npj + j;
                dzo[o] = s;
                if (mean_form) { darea[o] = a; if (rmax) rmax[o] = mx; if (rmin) rmin[o] = mn; }
            }
}

/* ------------------------------------------------------------------------------------------ */
/* src/cdfvint.f90:266-311 for one record, with the reference's own scalar schedule inside:
 *   gdepo (npko) output depths, e31d (npk) the 1-D thickness (rdep2 = rdep1 + e31d(jk), :282),
 *   lstd = 1 with -GSOP / -OCCI (lout starts .FALSE. at each level, :273), tol = 1.0.
 * out (npi,npj,npko) r4, zero where an output depth is never reached. */
void oracle_vint_cumul(int npi, int npj, int npk, int npko, const float *zt, const float *e3t, const float *e31d_full,
                       const float *tmask, const float *e31d, const float *gdepo, int lstd, float sclf, int mean,
                       float *out) {
    const size_t nij = (size_t)npi * npj;
    const float tol = 1.0f;
    double *v1 = (double *)calloc(nij, 8), *v2 = (double *)calloc(nij, 8);
    memset(out, 0, nij * npko * 4);
    int iko = 0, lout = 1;
    float rdep1 = 0.f, rdep2 = 0.f;
    for (int jk = 0; jk < npk; ++jk) {
        if (lstd) lout = 0;
        rdep1 = rdep2;
        memcpy(v2, v1, nij * 8);
        rdep2 = rdep1 + e31d[jk];
        const int hit = (iko < npko) && (rdep2 >= (gdepo[iko] - tol));
        for (size_t p = 0; p < nij; ++p) {
            float e = e3t ? e3t[jk * nij + p] : e31d_full[jk];
            const float t = zt[jk * nij + p], m = tmask[jk * nij + p];
            v1[p] = v1[p] + (double)(t * e * m) * 1.0;
            if (hit) {
                if (e > gdepo[iko] - rdep1) e = gdepo[iko] - rdep1;
                v2[p] = v2[p] + (double)(t * e * m) * 1.0;
            }
        }
        if (hit) lout = 1;
        if (lout && iko < npko) {
            for (size_t p = 0; p < nij; ++p) {
                v2[p] = v2[p] * (double)sclf;
                out[(size_t)iko * nij + p] = mean ? (float)v2[p] / rdep2 : (float)v2[p];
            }
            iko = iko + 1;
        }
    }
    free(v1); free(v2);
}

/* src/cdfvertmean.f90:236-270 for one variable and record; the levels ik1..ik2 are passed stacked. */
void oracle_vertmean(int npi, int npj, int nlev, const float *zv, const float *e3, const float *e31d, const float *zmask,
                     float gdep_first, float gdep_last, float rdep_up, float rdep_down, double *dmean) {
    const size_t nij = (size_t)npi * npj;
    for (size_t p = 0; p < nij; ++p) {
        double dvol = 0.0, ds = 0.0;
        for (int k = 0; k < nlev; ++k) {
            float e = e3 ? e3[k * nij + p] : e31d[k];
            if (k == 0) { const float hdep = gdep_first + e; const float c = hdep - rdep_up; if (c < e) e = c; }
            if (k == nlev - 1) { const float c = rdep_down - gdep_last; if (c < e) e = c; }
            dvol = (double)(e * zmask[k * nij + p]) * 1.0 + dvol;
            ds   = (double)(zv[k * nij + p] * e * zmask[k * nij + p]) * 1.0 + ds;
        }
        dmean[p] = (dvol != 0.0) ? ds / dvol : ds;
    }
}
