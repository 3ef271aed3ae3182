// eos_dev.hpp -- device side of the polynomial equation of state (NEMO EOS-80 / TEOS-10), shared by
// eos.hip (sigma0 / sigmai entry points) and mocsig.hip (density binning of cdfmocsig).  Every
// translation unit that includes it gets its own copy of the coefficient tables in constant memory.
// Reference: src/eos.f90:802-887 (sigmai_dep), coefficients from eos_init :207-598.
#pragma once
#include "common.hpp"
#include "eos_coef.h"

namespace {

__constant__ double kC[2][CDFT_EOS_NK][CDFT_EOS_NJ][CDFT_EOS_NI] = CDFT_EOS_COEF_INIT;
__constant__ double kR[6] = CDFT_EOS_REF_INIT;

template <bool FAST>
__device__ __forceinline__ double muladd(double a, double b, double c) {
    return FAST ? __fma_rn(a, b, c) : __dadd_rn(__dmul_rn(a, b), c);
}

// sqrt of the scaled salinity, correctly rounded like __dsqrt_rn, without the range scaling the general routine wraps around
// its iteration (a compare, two selects and two ldexp per point that only matter below 2^-767): the argument here is
// |S + deltaS| * r1_s0 with S an r4 number, i.e. 0, at least 1e-7, at most 1e37, Inf or NaN.  The iteration is the
// library's own (v_rsq_f64, one joint refinement of g = sqrt and h = 1 / (2 sqrt), two corrections of g by its residual);
// 0 and Inf pass through.  A correctly rounded result is one number: the tests hold it to the oracle's sqrt bit for bit.
__device__ __forceinline__ double eos_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = __dmul_rn(x, y), h = __dmul_rn(y, 0.5);
    const double r = __fma_rn(-h, g, 0.5);
    g = __fma_rn(g, r, g);
    h = __fma_rn(h, r, h);
    double d = __fma_rn(-g, g, x);
    g = __fma_rn(d, h, g);
    d = __fma_rn(-g, g, x);
    g = __fma_rn(d, h, g);
    return (x == 0.0 || x == __builtin_inf()) ? x : g;
}

// r_k(t,s) with the reference's association (src/eos.f90:858-875): each t-level is written
//   ( ... )*dlt + (((c_n*dls + c_{n-1})*dls + ...) + c_1)*dls + c_0  ==  (acc*t + Y*s) + c_0
template <int SET, int K, bool FAST>
__device__ __forceinline__ double eos_rk(double t, double s) {
    constexpr int jmax = CDFT_EOS_JMAX(K);
    double acc = kC[SET][K][jmax][0];
#pragma unroll
    for (int j = jmax - 1; j >= 0; --j) {
        const int imax = jmax - j;
        double y = kC[SET][K][j][imax];
#pragma unroll
        for (int i = imax - 1; i >= 1; --i) y = muladd<FAST>(y, s, kC[SET][K][j][i]);
        if (FAST)
            acc = __fma_rn(acc, t, __fma_rn(y, s, kC[SET][K][j][0]));
        else
            acc = __dadd_rn(__dadd_rn(__dmul_rn(acc, t), __dmul_rn(y, s)), kC[SET][K][j][0]);
    }
    return acc;
}

// DEPTH: 0 = pref looked at here (a uniform branch), 1 = the caller knows pref == 0 (surface form), 2 = the caller knows
// pref != 0 -- the same arithmetic, without the branch's joins (moves and selects of r8 pairs around it)
template <int SET, bool FAST, int DEPTH = 0>
__device__ __forceinline__ double sigmai_point(float ptem, float psal, float pref) {
    constexpr double rdeltas = SET ? CDFT_EOS_RDELTAS_T10 : CDFT_EOS_RDELTAS_80;
    constexpr double r1_s0   = SET ? CDFT_EOS_R1_S0_T10 : CDFT_EOS_R1_S0_80;
    const double t  = __dmul_rn((double)ptem, CDFT_EOS_R1_T0);                      // :850
    const double s  = eos_sqrt(__dmul_rn(fabs(__dadd_rn((double)psal, rdeltas)), r1_s0));    // :851
    const double tm = (psal == 0.0f) ? 0.0 : 1.0;                                   // :852-856
    const double r0 = eos_rk<SET, 0, FAST>(t, s);
    double r = r0, ref = 0.0;
    // At the surface (sigma0: pref = 0) h = 0, so ((r3*h + r2)*h + r1)*h + r0 is r0 and the reference
    // profile is +0 for every finite T,S: the three depth-dependent polynomials are skipped (bit-identical).
    if (DEPTH == 2 || (DEPTH == 0 && pref != 0.0f)) {
        const double h = __dmul_rn((double)pref, CDFT_EOS_R1_Z0);                   // :842 / :1046
        ref = kR[5];                                                                // :845 / :1050
#pragma unroll
        for (int q = 4; q >= 0; --q) ref = muladd<FAST>(ref, h, kR[q]);
        ref = __dmul_rn(ref, h);
        const double r3 = eos_rk<SET, 3, FAST>(t, s);
        const double r2 = eos_rk<SET, 2, FAST>(t, s);
        const double r1 = eos_rk<SET, 1, FAST>(t, s);
        r = muladd<FAST>(muladd<FAST>(muladd<FAST>(r3, h, r2), h, r1), h, r0);     // :877
    }
    return __dmul_rn(__dadd_rn(__dadd_rn(r, ref), -CDFT_EOS_RAU0), tm);            // :880
}

// Neutral density of McDougall and Jackett (2005) as src/eos.f90:634-687 writes it: every product and sum in r8, in
// the written order (no contraction); rho - 1000, not masked.  rho(20,35) - 1000 = 24.59416751197 (:646).
__device__ __forceinline__ double sigmantr_point(float ptem, float psal) {
    const double t = (double)ptem, s = (double)psal, sr = __dsqrt_rn(fabs(s));
    auto mul = [](double a, double b) { return __dmul_rn(a, b); };
    auto add = [](double a, double b) { return __dadd_rn(a, b); };
    const double r1 = add(mul(add(mul(add(mul(-4.3159255086706703e-4, t), 8.1157118782170051e-2), t), 2.2280832068441331e-1), t),
                          1002.3063688892480);
    const double r2 = mul(add(add(mul(-1.7052298331414675e-7, s), -mul(3.1710675488863952e-3, t)), -1.0304537539692924e-4), s);
    const double r3 = add(mul(add(mul(add(mul(add(mul(-2.3850178558212048e-9, t), -1.6212552470310961e-7), t), 7.8717799560577725e-5), t),
                                  4.3907692647825900e-5), t), 1.0);
    const double r4 = mul(add(mul(add(mul(mul(-2.2744455733317707e-9, t), t), 6.0399864718597388e-6), t), -5.1268124398160734e-4), s);
    const double r5 = mul(mul(add(mul(mul(-1.3409379420216683e-9, t), t), -3.6138532339703262e-5), s), sr);
    return add(__ddiv_rn(add(r1, r2), add(add(r3, r4), r5)), -1000.0);
}

// the density cdfmocsig bins: sigmai (KIND 0 EOS-80, 1 TEOS-10) or the neutral density (KIND 2, -ntr)
template <int KIND, int DEPTH = 0>
__device__ __forceinline__ double mocsig_density(float t, float s, float pref) {
    if constexpr (KIND == 2) return sigmantr_point(t, s);
    else return sigmai_point<KIND, false, DEPTH>(t, s, pref);
}

}  // namespace
