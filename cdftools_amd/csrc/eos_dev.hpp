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

template <int SET, bool FAST>
__device__ __forceinline__ double sigmai_point(float ptem, float psal, float pref) {
    constexpr double rdeltas = SET ? CDFT_EOS_RDELTAS_T10 : CDFT_EOS_RDELTAS_80;
    constexpr double r1_s0   = SET ? CDFT_EOS_R1_S0_T10 : CDFT_EOS_R1_S0_80;
    const double t  = __dmul_rn((double)ptem, CDFT_EOS_R1_T0);                      // :850
    const double s  = __dsqrt_rn(__dmul_rn(fabs(__dadd_rn((double)psal, rdeltas)), r1_s0));  // :851
    const double tm = (psal == 0.0f) ? 0.0 : 1.0;                                   // :852-856
    const double r0 = eos_rk<SET, 0, FAST>(t, s);
    double r = r0, ref = 0.0;
    // At the surface (sigma0: pref = 0) h = 0, so ((r3*h + r2)*h + r1)*h + r0 is r0 and the reference
    // profile is +0 for every finite T,S: the three depth-dependent polynomials are skipped (bit-identical).
    if (pref != 0.0f) {
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

}  // namespace
