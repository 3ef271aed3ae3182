/*
 * cdf_oracle_eos2.c -- CPU restatement of the REST of the reference's eos module (TEST INFRASTRUCTURE ONLY; see
 * cdf_oracle.h): spice (both members), alpha, beta (polynomial and McDougall 1987 members), albet, eosbn2 (both
 * members).  sigma0 / sigmai / sigmantr are in cdf_oracle.c.
 *
 * PINNED bit for bit to the reference's src/eos.f90 compiled where it lies (oracle/_ref/libeos_ref.so through
 * eos_ref_driver.f90; tests/golden/eos_more.npz; tests/test_eos_more.py), and to the check value the source states
 * for spiciness (src/eos.f90:700: spice(15,33) = 0.544586321373410).
 *
 * What a transcription can get wrong here is the r4/r8 mixing, so it is spelled out at each line:
 *   - literals without a kind (0.5, 35.0, 0.025, 9.81, 0.255019e-07 ...) are r4 constants: rounded to r4 first, then
 *     widened where the other operand is r8;
 *   - r4 (op) r4 is evaluated in r4 (psal - 35.0, 0.5*(t_up + t_down), t_up - t_down, ptem*0.025).
 */
#include <math.h>
#include <stdlib.h>

#include "cdf_oracle.h"
#include "eos_coef.h"

static const double alp_c[2][CDFT_EOS_NK][CDFT_EOS_DNJ][CDFT_EOS_DNI] = CDFT_EOS_ALP_INIT;
static const double bet_c[2][CDFT_EOS_NK][CDFT_EOS_DNJ][CDFT_EOS_DNI] = CDFT_EOS_BET_INIT;
static const double spice_s[49] = CDFT_SPICE_POL_INIT;
static const double spice_b[6][5] = CDFT_SPICE_F02_INIT;

/* n_k(t,s) of alpha / beta: the same nesting as the density's r_k (src/eos.f90:1424-1441, :1476-1494):
 *   acc = (acc*t + Y*s) + c0, Y the Horner form in s; dln3 is a bare coefficient. */
static double d_nk(const double c[CDFT_EOS_DNJ][CDFT_EOS_DNI], int k, double t, double s) {
    const int jmax = CDFT_EOS_DJMAX(k);
    double acc = c[jmax][0];
    for (int j = jmax - 1; j >= 0; --j) {
        const int imax = jmax - j;
        double y = c[j][imax];
        for (int i = imax - 1; i >= 1; --i) y = y * s + c[j][i];
        acc = (acc * t + y * s) + c[j][0];
    }
    return acc;
}

/* alpha_0d / beta_0d (src/eos.f90:1505-1543, :1596-1641) on r8 arguments; the 2-D and profile members (:1452-1503,
 * :1545-1594, :1400-1450, :1643-1691) widen their r4 arguments and do the same arithmetic: ptem*r1_T0 is r4*r8,
 * ABS(psal + rdeltaS) is r4 + r8, and the test psal == 0 is on the unwidened value (same truth). */
static double ab_point(int beta, double ptem, double psal, double pdep, int teos10) {
    const int set = teos10 ? 1 : 0;
    const double rdeltas = teos10 ? CDFT_EOS_RDELTAS_T10 : CDFT_EOS_RDELTAS_80;
    const double r1_s0 = teos10 ? CDFT_EOS_R1_S0_T10 : CDFT_EOS_R1_S0_80;
    const double (*c)[CDFT_EOS_DNJ][CDFT_EOS_DNI] = beta ? bet_c[set] : alp_c[set];
    const double h = pdep * CDFT_EOS_R1_Z0;
    const double t = ptem * CDFT_EOS_R1_T0;
    const double s = sqrt(fabs(psal + rdeltas) * r1_s0);
    const double tm = (psal == 0.0) ? 0.0 : 1.0;
    const double n3 = d_nk(c[3], 3, t, s), n2 = d_nk(c[2], 2, t, s), n1 = d_nk(c[1], 1, t, s), n0 = d_nk(c[0], 0, t, s);
    const double n = ((n3 * h + n2) * h + n1) * h + n0;
    return beta ? n / s * tm : n * tm;                       /* :1448 dln / dls * dltm ; :1500 dln * dltm */
}

/* the ratio alpha/beta and beta of McDougall (1987) as the reference writes them three times each
 * (src/eos.f90:1279-1291 = :1338-1350 = :1381-1393 and :1293-1304 = :1718-1728 = :1760-1770); zt, zs, zh are r8,
 * every numeric constant an r4 literal.  Fortran's precedence and left-to-right order kept term by term. */
#define F(x) ((double)(x##f))
static double albet_mcd(double zt, double zs, double zh) {
    return ((((-F(0.255019e-07)) * zt + F(0.298357e-05)) * zt - F(0.203814e-03)) * zt + F(0.170907e-01)) * zt
           + F(0.665157e-01)
           + ((-F(0.678662e-05)) * zs - F(0.846960e-04) * zt + F(0.378110e-02)) * zs
           + (((-F(0.302285e-13)) * zh - F(0.251520e-11) * zs + F(0.512857e-12) * zt * zt) * zh
              - F(0.164759e-06) * zs
              + (F(0.791325e-08) * zt - F(0.933746e-06)) * zt
              + F(0.380374e-04)) * zh;
}
static double beta_mcd(double zt, double zs, double zh) {
    return (((-F(0.415613e-09)) * zt + F(0.555579e-07)) * zt - F(0.301985e-05)) * zt
           + F(0.785567e-03)
           + (F(0.515032e-08) * zs + F(0.788212e-08) * zt - F(0.356603e-06)) * zs
           + ((F(0.121551e-17) * zh - F(0.602281e-15) * zs - F(0.175379e-14) * zt + F(0.176621e-12)) * zh
              + F(0.408195e-10) * zs
              + ((-F(0.213127e-11)) * zt + F(0.192867e-09)) * zt
              - F(0.121555e-07)) * zh;
}

/* which: 0 alpha (polynomial), 1 beta (polynomial), 2 alpha/beta (McDougall 1987; "albet"), 3 beta (McDougall 1987).
 * pdep2d == NULL: one depth for all points (the 2-D members), else a depth per point (the profile members). */
void oracle_eos_coef(int which, const float *ptem, const float *psal, long n, float pdep, const float *pdep2d,
                     int teos10, double *out) {
    for (long p = 0; p < n; ++p) {
        const float dep = pdep2d ? pdep2d[p] : pdep;
        if (which <= 1) {
            out[p] = ab_point(which, (double)ptem[p], (double)psal[p], (double)dep, teos10);
        } else {
            const double zt = (double)ptem[p];
            const double zs = (double)(psal[p] - 35.0f);                        /* :1336 r4 - r4 */
            out[p] = which == 2 ? albet_mcd(zt, zs, (double)dep) : beta_mcd(zt, zs, (double)dep);
        }
    }
}

/* spice_pol (src/eos.f90:755-800, the GSW spiciness-0 fit; ld_teos10 only selects this member of the generic) and
 * spice_80 (:689-753, Flament 2002). */
void oracle_spice(int flament, const float *ptem, const float *psal, long n, double *out) {
    if (!flament) {
        const double *s = spice_s - 1;                       /* s[1]..s[49] as the source numbers them */
        for (long p = 0; p < n; ++p) {
            const double xs = sqrt(CDFT_SPICE_SFAC * (double)psal[p] + CDFT_SPICE_OFFSET);      /* :789 */
            const double ys = (double)(ptem[p] * 0.025f);                                       /* :790 r4 * r4 */
            double poly[7];
            for (int m = 0; m < 7; ++m) {                    /* sA + ys*(sB + ys*(... (sF + sG*ys))) */
                const double *c = s + 7 * m;
                poly[m] = c[1] + ys * (c[2] + ys * (c[3] + ys * (c[4] + ys * (c[5] + ys * (c[6] + c[7] * ys)))));
            }
            double acc = poly[6];                            /* :792-799: (poly_m) + xs*( ... ), innermost first */
            for (int m = 5; m >= 0; --m) acc = poly[m] + xs * acc;
            out[p] = acc;
        }
    } else {
        for (long p = 0; p < n; ++p) {
            const double salref = (double)psal[p] - 35.0;    /* :736 r4 - r8 */
            double spi = 0.0, tempt = 1.0;
            for (int i = 0; i < 6; ++i) {
                double salt = 1.0;
                for (int j = 0; j < 5; ++j) {
                    spi = spi + spice_b[i][j] * tempt * salt;   /* :743 */
                    salt = salt * salref;
                }
                tempt = tempt * (double)ptem[p];              /* :746 r8 * r4 */
            }
            out[p] = spi;
        }
    }
}

/* eosbn2_pol (src/eos.f90:1203-1244) and eosbn2_80 (:1246-1312): N^2 at the w level between two T levels.
 * form 0: polynomial alpha, beta of the EOS set; NOTE the reference hands alpha_0d / beta_0d the salinity ANOMALY
 * (s - 35) where they expect a salinity (:1231-1235) -- kept.  form 1: McDougall (1987).  Result stored in r4. */
void oracle_bn2(int form, const float *t_up, const float *t_dn, const float *s_up, const float *s_dn, const float *e3w,
                long n, float pdep, int teos10, float *bn2) {
    const double grav = (double)9.81f;                       /* REAL(KIND=8) :: dlgrav=9.81 */
    const double r1_rau0 = 1.0 / CDFT_EOS_RAU0;
    const double zh = (double)pdep;
    for (long p = 0; p < n; ++p) {
        const double zt = (double)(0.5f * (t_up[p] + t_dn[p]));
        const double zs = (double)(0.5f * (s_up[p] + s_dn[p]) - 35.0f);
        const double dt = (double)(t_up[p] - t_dn[p]), ds = (double)(s_up[p] - s_dn[p]);
        if (form == 0) {
            const double g = grav / (double)e3w[p] * r1_rau0;
            const double a = ab_point(0, zt, zs, zh, teos10), b = ab_point(1, zt, zs, zh, teos10);
            bn2[p] = (float)(g * (a * dt - b * ds));
        } else {
            const double g = grav / (double)e3w[p];
            bn2[p] = (float)(g * beta_mcd(zt, zs, zh) * (albet_mcd(zt, zs, zh) * dt - ds));
        }
    }
}
