/*
 * cdf_oracle.h -- CPU restatement of the CDFTOOLS hot-path loops (TEST INFRASTRUCTURE ONLY).
 *
 * Nothing under cdftools_amd/ may include, link or call this.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Every function restates one reference loop in the reference's own evaluation order and r4/r8
 * mixing; the reference file:line it follows is cited at each definition in cdf_oracle.c.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - sigma0 / sigmai / sigmantr, and spice / alpha / beta / albet / eosbn2 (every generic member) : PINNED bit for bit to the reference's src/eos.f90 compiled where it lies
 *     (oracle/_ref/libeos_ref.so; tests/golden/eos.npz) and to its known-answer values (src/eos.f90:817,:820).
 *   - cdfmoc (+ -decomp, -full), cdfheatc (r4sum mode, + -full, -mxloption), cdfmhst (-vt / V,T,S / -Zdim / -full),
 *     cdfpsi (default, -V, -mean, -mask, -ref, -full, -open, --ssh-file, -lev), cdfvT (+ -vvl), cdfcurl (+ -T,
 *     -overf, level 0) : PINNED bit for bit to the reference's own STATEMENTS.  The programs cannot be built here
 *     (netcdf-fortran is absent), so oracle/ref_lift.py lifts their declaration blocks, ALLOCATEs and whole time
 *     loops by line range into harnesses with plain array arguments (only I/O statements are rewritten), compiled
 *     with amdflang -O1 -ffp-contract=off into oracle/_ref/liblift_ref.so; tests/golden/lift_*.npz hold the
 *     harness outputs and tests/test_lift.py compares.  That pins evaluation order, r4/r8 mixing and index ranges
 *     -- what a transcription can get wrong -- not cdfio, the option parsers or the file layouts.
 *   - cdfmocsig (+ -isodep, -eiv, -full, -ntr, TEOS-10), cdfmean (+ basins, -var, -S, -full), cdfsum (model / forcing
 *     form, -nowght -nomsk), cdfzonalsum (+ -pdeg), cdfzonalmean (+ -max), cdfvint (model / GSOP / OCCIPUT depths, means,
 *     -full), cdfvertmean, cdfvhst, cdfvtrp : PINNED the same way (tests/test_lift_next.py).
 *   - the level loops of cdfsig0 / cdfsigi (oracle_sig_tool) : PINNED the same way (tests/golden/lift_cdfsig.npz).
 *
 * Arrays are Fortran column-major (first index fastest), as in include/cdftools_hip.h.
 */
#ifndef CDF_ORACLE_H
#define CDF_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/cdfmoc.f90:352-388 for ONE record.
 *   e1v(npi,npj) e3v(npi,npj,npk) [already * vmask(level)] ibmask(nb,npi,npj) zv(npi,npj,npk)
 *   dmoc(nb,npj,npk) out, Sverdrup.  Levels 1..npk-1 of zv are read (jk=1,npk-1). */
void oracle_moc(int npi, int npj, int npk, int nb, const float *e1v, const float *e3v,
                const int16_t *ibmask, const float *zv, double *dmoc);
/* src/cdfmoc.f90:353,:360-365,:390-517 (-decomp) for ONE record; see the definition for the argument
 * list and the two reference quirks it keeps (dmoc_sh is never cleared; the border of zv). */
void oracle_moc_decomp(int npi, int npj, int npk, int nb, const float *e1v, const float *e3v,
                       const int16_t *ibmask, const float *zv3, const float *ztemp, const float *zsal,
                       const uint8_t *tmask, const uint8_t *umask, const float *e1u, const float *zcoef,
                       const float *gdept, int teos10, const double *dmoc, double *dmoc_sh,
                       double *dmoc_bt, double *dmoc_ag);
/* cdfmocsig, src/cdfmocsig.f90:368-470 for ONE record; see the definition for the arguments */
void oracle_mocsig(int npi, int npj, int npk, int nb, const float *zv3, const float *zveiv, const float *zt3,
                   const float *zs3, const float *e1v, const float *e3v, const float *e31d, const int16_t *ibmask,
                   const float *gdep, float zspv, float zspt, float zsps, float pref, int teos10, int nbins,
                   float sigmin, float sigstp, double *dmoc, double *depi);
/* src/eos.f90:802-887 (sigmai_dep), :1002-1087 (sigmai_dep2d), :601-632 (sigma0 = pref 0);
 * coefficients as set by eos_init :207-598 (teos10 = 0: EOS-80 polynomial, 1: TEOS-10).
 * ptem, psal: n r4 values; pref: one r4 depth (pref2d == NULL) or n per-point depths. */
void oracle_sigmai(const float *ptem, const float *psal, long n, float pref, const float *pref2d,
                   int teos10, double *sig);
/* src/eos.f90:889-1000 (sigmai_dep_80) and :1089-1201 (sigmai_dep2d_80): Jackett-McDougall 1994 */
void oracle_sigmai_jmd94(const float *ptem, const float *psal, long n, float pref, const float *pref2d,
                         double *sig);
void oracle_sigmantr(const float *ptem, const float *psal, long n, double *out);   /* src/eos.f90:634-687 */
/* the rest of the eos module's public list (cdf_oracle_eos2.c): which = 0 alpha, 1 beta (polynomial, src/eos.f90:1400-1691),
 * 2 alpha/beta, 3 beta (McDougall 1987, :1315-1398, :1693-1775); one depth, or a depth per point (the profile members) */
void oracle_eos_coef(int which, const float *ptem, const float *psal, long n, float pdep, const float *pdep2d,
                     int teos10, double *out);
/* spice_pol (flament = 0, :755-800) and spice_80 (flament = 1, :689-753) */
void oracle_spice(int flament, const float *ptem, const float *psal, long n, double *out);
/* eosbn2_pol (form 0, :1203-1244) and eosbn2_80 (form 1, :1246-1312); r4 result */
void oracle_bn2(int form, const float *t_up, const float *t_dn, const float *s_up, const float *s_dn, const float *e3w,
                long n, float pdep, int teos10, float *bn2);
/* callers src/cdfsig0.f90:146-152, src/cdfsigi.f90:131-137: r4 result = sigma * (S /= spval) */
void oracle_sig_tool(const float *ptem, const float *psal, long n, float pref, float spval, int teos10,
                     float *sig4);

/* cdf_oracle_tools.c -- see the comment at each definition for the reference lines followed */
void oracle_heatc(int npi, int npj, int npk, const float *e1t, const float *e2t, const float *e3t,
                  const float *e31d, const float *tmask, const float *temp, const float *mxl,
                  const float *gdepw, int mxlopt, int r4sum, double *sums);
void oracle_mhst(int npi, int npj, int npk, int nb, const float *e1v, const float *e3v, const float *e31d,
                 const float *vt, const float *vs, const float *zv, const float *zt, const float *zs,
                 const float *masks, int zdim, double *heat, double *salt);
void oracle_psi(int npi, int npj, int npk, const float *zu, const float *e2u, const float *e3u,
                const float *zv, const float *e1v, const float *e3v, const float *e31d, int iiref,
                int ijref, const float *fmask, float *psi4, double *dpsi8);
void oracle_vint_transport(int npi, int npj, int npk, const float *vel, const float *m2d, const float *e3,
                           const float *e31d, double *dtrp);
void oracle_vtrp_slope(int npi, int npj, const double *dtrpu, const double *dtrpv, const float *hdepw, const float *e1u,
                       const float *e2v, const float *tmask, double *along, double *cross);
void oracle_vhst_transport(int npi, int npj, int npk, const float *xt, const float *xs, const float *m2d,
                           const float *e3, const float *e31d, double *dheat, double *dsalt);
void oracle_mean(int npi, int npj, int npk, int nb, const float *e1, const float *e2, const float *e3,
                 const float *e31d, const float *zmask, const int16_t *ibmask, const float *zv, double *sums);
void oracle_mean_form(int npi, int npj, int npk, int nb, const float *e1, const float *e2, const float *e3,
                      const float *e31d, const float *zmask, const int16_t *ibmask, const float *zv, int form,
                      double *sums);
void oracle_zonal(int npi, int npj, int nlev, int nb, const float *e1, const float *e2, const float *zmaskvar,
                  const float *zmask, const float *zv, int mean_form, double *dzo, double *darea, float *rmax,
                  float *rmin);
void oracle_vint_cumul(int npi, int npj, int npk, int npko, const float *zt, const float *e3t, const float *e31d_full,
                       const float *tmask, const float *e31d, const float *gdepo, int lstd, float sclf, int mean,
                       float *out);
void oracle_vertmean(int npi, int npj, int nlev, const float *zv, const float *e3, const float *e31d, const float *zmask,
                     float gdep_first, float gdep_last, float rdep_up, float rdep_down, double *dmean);
void oracle_psi_open(int npi, int npj, int npk, const float *zu, const float *e2u, const float *e3u,
                     const float *zv, const float *e1v, const float *e3v, const float *e31d, int lmean,
                     int iiref, int ijref, int noref, const float *fmask, float *psi4, double *dpsi8);
void oracle_vt(int npi, int npj, int nfr, long fstride, const float *u, const float *v, const float *t,
               const float *s, const float *e3u, const float *e3v, double *vt, double *vs, double *ut,
               double *us, double *e3um, double *e3vm);
void oracle_curl(int npi, int npj, const float *un, const float *vn, const float *e1u, const float *e2v,
                 const float *e1f, const float *e2f, int lperio, int ltpoint, const double *ff,
                 double *rotn);

/* number of OpenMP threads the oracle's parallel loops use (1 when built without OpenMP) */
int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
