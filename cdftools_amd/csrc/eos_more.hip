// eos_more.hip -- the rest of the eos module's public list as pointwise kernels (C ABI section "eos, the other members"):
// thermal expansion alpha, haline contraction beta (polynomial of Roquet et al. 2015, src/eos.f90:1400-1691; the
// McDougall 1987 forms of beta and alpha/beta, :1315-1398, :1693-1775), spiciness (GSW spiciness-0 fit :755-800 and
// Flament 2002 :689-753) and the Brunt-Vaisala frequency between two T levels (eosbn2_pol :1203-1244, eosbn2_80
// :1246-1312).  sigma0 / sigmai / sigmantr are in eos.hip.
//
// Every expression keeps the reference's evaluation order and its r4/r8 mixing (literals without a kind are r4
// constants; r4 (op) r4 stays r4); the library is compiled with -ffp-contract=off, so the plain C++ expressions below
// are evaluated as written and the results are bit-identical to the reference's compiled module (tests/test_eos_more.py).
//
// 8-20 bytes of traffic per point for 60-250 r8 operations: these sit at or beyond the fp64-vector ridge of the chip
// (like sigmai); one lane per point, four points per lane when the arrays are 16-byte aligned.
#include "common.hpp"
#include <type_traits>
#include "eos_coef.h"

using namespace cdft;

namespace {

// constexpr, not __constant__: with every index known after unrolling the coefficients become literals of the instruction
// stream.  As __constant__ arrays they were fetched into ~180 scalar registers ahead of the loop, more than a wave has,
// and came back through v_readlane / v_writelane: 180 of the 402 vector instructions per point of eosbn2 (counters,
// profiles/r03/fp64_ops_pmc.json: 298 per point issued, 170 of them r8).
constexpr double kAlp[2][CDFT_EOS_NK][CDFT_EOS_DNJ][CDFT_EOS_DNI] = CDFT_EOS_ALP_INIT;
constexpr double kBet[2][CDFT_EOS_NK][CDFT_EOS_DNJ][CDFT_EOS_DNI] = CDFT_EOS_BET_INIT;
constexpr double kSpice[49] = CDFT_SPICE_POL_INIT;
constexpr double kF02[6][5] = CDFT_SPICE_F02_INIT;

// n_k(t,s): Horner in t around Horner in s, each t-level written ( .. )*t + ( .. )*s + c0 (src/eos.f90:1424-1441)
template <bool BETA, int SET, int K>
__device__ __forceinline__ double d_nk(double t, double s) {
    constexpr int jmax = CDFT_EOS_DJMAX(K);
    auto c = [](int j, int i) { return BETA ? kBet[SET][K][j][i] : kAlp[SET][K][j][i]; };
    double acc = c(jmax, 0);
#pragma unroll
    for (int j = jmax - 1; j >= 0; --j) {
        const int imax = jmax - j;
        double y = c(j, imax);
#pragma unroll
        for (int i = imax - 1; i >= 1; --i) y = y * s + c(j, i);
        acc = (acc * t + y * s) + c(j, 0);
    }
    return acc;
}

// alpha_0d / beta_0d (:1505-1543, :1596-1641); the array members widen their r4 arguments into the same arithmetic
template <bool BETA, int SET>
__device__ __forceinline__ double ab_point(double ptem, double psal, double pdep) {
    constexpr double rdeltas = SET ? CDFT_EOS_RDELTAS_T10 : CDFT_EOS_RDELTAS_80;
    constexpr double r1_s0   = SET ? CDFT_EOS_R1_S0_T10 : CDFT_EOS_R1_S0_80;
    const double h  = pdep * CDFT_EOS_R1_Z0;
    const double t  = ptem * CDFT_EOS_R1_T0;
    const double s  = __dsqrt_rn(fabs(psal + rdeltas) * r1_s0);
    const double tm = (psal == 0.0) ? 0.0 : 1.0;
    const double n3 = d_nk<BETA, SET, 3>(t, s), n2 = d_nk<BETA, SET, 2>(t, s);
    const double n1 = d_nk<BETA, SET, 1>(t, s), n0 = d_nk<BETA, SET, 0>(t, s);
    const double n  = ((n3 * h + n2) * h + n1) * h + n0;
    return BETA ? __ddiv_rn(n, s) * tm : n * tm;
}

// McDougall (1987): the constants are r4 literals in the source, widened where they meet an r8 operand
#define R4(x) ((double)(x##f))
__device__ __forceinline__ double albet_mcd(double zt, double zs, double zh) {     // :1338-1350
    return ((((-R4(0.255019e-07)) * zt + R4(0.298357e-05)) * zt - R4(0.203814e-03)) * zt + R4(0.170907e-01)) * zt
           + R4(0.665157e-01)
           + ((-R4(0.678662e-05)) * zs - R4(0.846960e-04) * zt + R4(0.378110e-02)) * zs
           + (((-R4(0.302285e-13)) * zh - R4(0.251520e-11) * zs + R4(0.512857e-12) * zt * zt) * zh
              - R4(0.164759e-06) * zs
              + (R4(0.791325e-08) * zt - R4(0.933746e-06)) * zt
              + R4(0.380374e-04)) * zh;
}
__device__ __forceinline__ double beta_mcd(double zt, double zs, double zh) {      // :1718-1728
    return (((-R4(0.415613e-09)) * zt + R4(0.555579e-07)) * zt - R4(0.301985e-05)) * zt
           + R4(0.785567e-03)
           + (R4(0.515032e-08) * zs + R4(0.788212e-08) * zt - R4(0.356603e-06)) * zs
           + ((R4(0.121551e-17) * zh - R4(0.602281e-15) * zs - R4(0.175379e-14) * zt + R4(0.176621e-12)) * zh
              + R4(0.408195e-10) * zs
              + ((-R4(0.213127e-11)) * zt + R4(0.192867e-09)) * zt
              - R4(0.121555e-07)) * zh;
}

__device__ __forceinline__ double spice_pol_point(float ptem, float psal) {       // :789-799
    const double xs = __dsqrt_rn(CDFT_SPICE_SFAC * (double)psal + CDFT_SPICE_OFFSET);
    const double ys = (double)(ptem * 0.025f);
    double acc = 0.0;
#pragma unroll
    for (int m = 6; m >= 0; --m) {
        const double *c = kSpice + 7 * m;
        const double pm = c[0] + ys * (c[1] + ys * (c[2] + ys * (c[3] + ys * (c[4] + ys * (c[5] + c[6] * ys)))));
        acc = (m == 6) ? pm : pm + xs * acc;
    }
    return acc;
}

__device__ __forceinline__ double spice_f02_point(float ptem, float psal) {       // :734-748
    const double salref = (double)psal - 35.0;
    double spi = 0.0, tempt = 1.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double salt = 1.0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            spi = spi + kF02[i][j] * tempt * salt;
            salt = salt * salref;
        }
        tempt = tempt * (double)ptem;
    }
    return spi;
}

template <int Q, int SET>
__device__ __forceinline__ double field_point(float t, float s, float dep) {
    if constexpr (Q == CDFT_EOSQ_ALPHA) return ab_point<false, SET>((double)t, (double)s, (double)dep);
    else if constexpr (Q == CDFT_EOSQ_BETA) return ab_point<true, SET>((double)t, (double)s, (double)dep);
    else if constexpr (Q == CDFT_EOSQ_ALBET_MCD87) return albet_mcd((double)t, (double)(s - 35.0f), (double)dep);
    else if constexpr (Q == CDFT_EOSQ_BETA_MCD87) return beta_mcd((double)t, (double)(s - 35.0f), (double)dep);
    else if constexpr (Q == CDFT_EOSQ_SPICE_POL) return spice_pol_point(t, s);
    else return spice_f02_point(t, s);
}

struct FieldArgs {
    const float *t, *s, *d;     // d: per-point depth or NULL
    double *out;
    int64_t n;
    float dep;
};

template <int Q, int SET, bool PDEP, int VEC>
__global__ void __launch_bounds__(256) eos_field_kernel(FieldArgs a) {
    const int64_t nvec = a.n / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        if constexpr (VEC == 4) {
            const f4a tt = ld_stream4(a.t + 4 * v), ss = ld_stream4(a.s + 4 * v);
            f4a dd = {a.dep, a.dep, a.dep, a.dep};
            if constexpr (PDEP) dd = ld_stream4(a.d + 4 * v);
            const double r0 = field_point<Q, SET>(tt.x, ss.x, dd.x), r1 = field_point<Q, SET>(tt.y, ss.y, dd.y);
            const double r2 = field_point<Q, SET>(tt.z, ss.z, dd.z), r3 = field_point<Q, SET>(tt.w, ss.w, dd.w);
            reinterpret_cast<double2 *>(a.out)[2 * v]     = make_double2(r0, r1);
            reinterpret_cast<double2 *>(a.out)[2 * v + 1] = make_double2(r2, r3);
        } else {
            a.out[v] = field_point<Q, SET>(a.t[v], a.s[v], PDEP ? a.d[v] : a.dep);
        }
    }
    if (VEC == 4 && blockIdx.x == 0 && threadIdx.x < (unsigned)(a.n - nvec * VEC)) {
        const int64_t q = nvec * VEC + threadIdx.x;
        a.out[q] = field_point<Q, SET>(a.t[q], a.s[q], PDEP ? a.d[q] : a.dep);
    }
}

// alpha_0d / beta_0d: one point with r8 arguments
template <bool BETA>
__global__ void eos_point_kernel(double t, double s, double dep, int teos, double *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *out = teos ? ab_point<BETA, 1>(t, s, dep) : ab_point<BETA, 0>(t, s, dep);
}

struct Bn2Args {
    const float *tu, *td, *su, *sd, *e3w;
    float *out;
    int64_t n;
    float dep;
};

// FORM 0: eosbn2_pol with coefficient set SET; note alpha_0d / beta_0d receive the salinity ANOMALY there (:1231-1235).
// FORM 1: eosbn2_80.
template <int FORM, int SET>
__device__ __forceinline__ float bn2_point(float tu, float td, float su, float sd, float e3w, double zh) {
    const double grav = (double)9.81f;
    const double zt = (double)(0.5f * (tu + td));
    const double zs = (double)(0.5f * (su + sd) - 35.0f);
    const double dt = (double)(tu - td), ds = (double)(su - sd);
    if constexpr (FORM == 0) {
        const double g = __ddiv_rn(grav, (double)e3w) * (1.0 / CDFT_EOS_RAU0);
        const double al = ab_point<false, SET>(zt, zs, zh), be = ab_point<true, SET>(zt, zs, zh);
        return (float)(g * (al * dt - be * ds));
    } else {
        const double g = __ddiv_rn(grav, (double)e3w);
        return (float)(g * beta_mcd(zt, zs, zh) * (albet_mcd(zt, zs, zh) * dt - ds));
    }
}

template <int FORM, int SET, int VEC>
__global__ void __launch_bounds__(256) eos_bn2_kernel(Bn2Args a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double zh = (double)a.dep;
    const int64_t nvec = a.n / VEC;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        if constexpr (VEC == 4) {               // four independent points per lane: their chains fill one another's latency
            const f4a tu = ld_stream4(a.tu + 4 * v), td = ld_stream4(a.td + 4 * v), su = ld_stream4(a.su + 4 * v);
            const f4a sd = ld_stream4(a.sd + 4 * v), e3 = ld_stream4(a.e3w + 4 * v);
            f4a r;
            r.x = bn2_point<FORM, SET>(tu.x, td.x, su.x, sd.x, e3.x, zh);
            r.y = bn2_point<FORM, SET>(tu.y, td.y, su.y, sd.y, e3.y, zh);
            r.z = bn2_point<FORM, SET>(tu.z, td.z, su.z, sd.z, e3.z, zh);
            r.w = bn2_point<FORM, SET>(tu.w, td.w, su.w, sd.w, e3.w, zh);
            *reinterpret_cast<f4a *>(a.out + 4 * v) = r;
        } else {
            a.out[v] = bn2_point<FORM, SET>(ld_stream(a.tu + v), ld_stream(a.td + v), ld_stream(a.su + v), ld_stream(a.sd + v),
                                            ld_stream(a.e3w + v), zh);
        }
    }
    if (VEC == 4 && blockIdx.x == 0 && threadIdx.x < (unsigned)(a.n - nvec * VEC)) {
        const int64_t q = nvec * VEC + threadIdx.x;
        a.out[q] = bn2_point<FORM, SET>(a.tu[q], a.td[q], a.su[q], a.sd[q], a.e3w[q], zh);
    }
}

inline unsigned grid_for(int64_t work) {
    int64_t blocks = (work + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

template <int Q, int SET>
int launch_field(const FieldArgs &a, hipStream_t st) {
    auto al = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec = al(a.t) && al(a.s) && al(a.d) && al(a.out);
    const unsigned g = grid_for(vec ? (a.n + 3) / 4 : a.n);
    if (a.d) {
        if (vec) hipLaunchKernelGGL((eos_field_kernel<Q, SET, true, 4>), dim3(g), dim3(256), 0, st, a);
        else     hipLaunchKernelGGL((eos_field_kernel<Q, SET, true, 1>), dim3(g), dim3(256), 0, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((eos_field_kernel<Q, SET, false, 4>), dim3(g), dim3(256), 0, st, a);
        else     hipLaunchKernelGGL((eos_field_kernel<Q, SET, false, 1>), dim3(g), dim3(256), 0, st, a);
    }
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

DevBuf g_in[5], g_out;      // staging for host arrays; single host thread, as the reference

}  // namespace

extern "C" {

int cdft_eos_field(int quantity, const float *ptem, const float *psal, const float *pdep_n, float pdep, int64_t n, int eos,
                   double *out, int in_space, int out_space, void *stream) {
    if (n < 0) return fail(CDFT_ERR_ARG, "cdft_eos_field: negative point count");
    if (quantity < CDFT_EOSQ_ALPHA || quantity > CDFT_EOSQ_SPICE_F02)
        return fail(CDFT_ERR_ARG, "cdft_eos_field: unknown quantity %d", quantity);
    if (quantity <= CDFT_EOSQ_BETA && eos != CDFT_EOS_80 && eos != CDFT_EOS_TEOS10)
        return fail(CDFT_ERR_ARG, "cdft_eos_field: alpha and beta exist for CDFT_EOS_80 and CDFT_EOS_TEOS10, not for %d", eos);
    if (n == 0) return CDFT_OK;
    if (!ptem || !psal || !out) return fail(CDFT_ERR_ARG, "cdft_eos_field: NULL argument");
    if (out_space != CDFT_MEM_HOST && out_space != CDFT_MEM_DEVICE)
        return fail(CDFT_ERR_ARG, "cdft_eos_field: bad memory-space tag %d", out_space);
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const void *dt, *ds, *dd = nullptr;
    CDFT_TRY(to_device(ptem, n * sizeof(float), in_space, g_in[0], st, &dt));
    CDFT_TRY(to_device(psal, n * sizeof(float), in_space, g_in[1], st, &ds));
    if (pdep_n && quantity <= CDFT_EOSQ_BETA_MCD87) CDFT_TRY(to_device(pdep_n, n * sizeof(float), in_space, g_in[2], st, &dd));
    void *dout = out;
    if (out_space == CDFT_MEM_HOST) {
        CDFT_TRY(g_out.reserve(n * sizeof(double) + 16));
        dout = g_out.ptr;
    }
    FieldArgs a{(const float *)dt, (const float *)ds, (const float *)dd, (double *)dout, n, pdep};
    int rc;
    const bool teos = eos == CDFT_EOS_TEOS10;
    switch (quantity) {
        case CDFT_EOSQ_ALPHA: rc = teos ? launch_field<CDFT_EOSQ_ALPHA, 1>(a, st) : launch_field<CDFT_EOSQ_ALPHA, 0>(a, st); break;
        case CDFT_EOSQ_BETA:  rc = teos ? launch_field<CDFT_EOSQ_BETA, 1>(a, st) : launch_field<CDFT_EOSQ_BETA, 0>(a, st); break;
        case CDFT_EOSQ_ALBET_MCD87: rc = launch_field<CDFT_EOSQ_ALBET_MCD87, 0>(a, st); break;
        case CDFT_EOSQ_BETA_MCD87:  rc = launch_field<CDFT_EOSQ_BETA_MCD87, 0>(a, st); break;
        case CDFT_EOSQ_SPICE_POL:   rc = launch_field<CDFT_EOSQ_SPICE_POL, 0>(a, st); break;
        default:                    rc = launch_field<CDFT_EOSQ_SPICE_F02, 0>(a, st); break;
    }
    CDFT_TRY(rc);
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(out, dout, n * sizeof(double), hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_eos_bn2(const float *ptem_up, const float *ptem_down, const float *psal_up, const float *psal_down, const float *pe3w,
                 float pdep, int64_t n, int eos, float *bn2, int in_space, int out_space, void *stream) {
    if (n < 0) return fail(CDFT_ERR_ARG, "cdft_eos_bn2: negative point count");
    if (eos != CDFT_EOS_80 && eos != CDFT_EOS_TEOS10 && eos != CDFT_EOS_MCD87)
        return fail(CDFT_ERR_ARG, "cdft_eos_bn2: eos must be CDFT_EOS_80, CDFT_EOS_TEOS10 or CDFT_EOS_MCD87, not %d", eos);
    if (n == 0) return CDFT_OK;
    if (!ptem_up || !ptem_down || !psal_up || !psal_down || !pe3w || !bn2) return fail(CDFT_ERR_ARG, "cdft_eos_bn2: NULL argument");
    if (out_space != CDFT_MEM_HOST && out_space != CDFT_MEM_DEVICE)
        return fail(CDFT_ERR_ARG, "cdft_eos_bn2: bad memory-space tag %d", out_space);
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const float *src[5] = {ptem_up, ptem_down, psal_up, psal_down, pe3w};
    const void *dev[5];
    for (int q = 0; q < 5; ++q) CDFT_TRY(to_device(src[q], n * sizeof(float), in_space, g_in[q], st, &dev[q]));
    void *dout = bn2;
    if (out_space == CDFT_MEM_HOST) {
        CDFT_TRY(g_out.reserve(n * sizeof(float) + 16));
        dout = g_out.ptr;
    }
    Bn2Args a{(const float *)dev[0], (const float *)dev[1], (const float *)dev[2], (const float *)dev[3], (const float *)dev[4],
              (float *)dout, n, pdep};
    bool vec = (reinterpret_cast<uintptr_t>(dout) & 15u) == 0;
    for (int q = 0; q < 5; ++q) vec = vec && (reinterpret_cast<uintptr_t>(dev[q]) & 15u) == 0;
    const unsigned g = grid_for(vec ? (n + 3) / 4 : n);
    auto launch = [&](auto form, auto set) {
        if (vec) hipLaunchKernelGGL((eos_bn2_kernel<decltype(form)::value, decltype(set)::value, 4>), dim3(g), dim3(256), 0, st, a);
        else     hipLaunchKernelGGL((eos_bn2_kernel<decltype(form)::value, decltype(set)::value, 1>), dim3(g), dim3(256), 0, st, a);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (eos == CDFT_EOS_MCD87)       launch(I1{}, I0{});
    else if (eos == CDFT_EOS_TEOS10) launch(I0{}, I1{});
    else                             launch(I0{}, I0{});
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(bn2, dout, n * sizeof(float), hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_eos_point(int quantity, double ptem, double psal, double pdep, int eos, double *out) {
    if (quantity != CDFT_EOSQ_ALPHA && quantity != CDFT_EOSQ_BETA)
        return fail(CDFT_ERR_ARG, "cdft_eos_point: quantity must be CDFT_EOSQ_ALPHA or CDFT_EOSQ_BETA, not %d", quantity);
    if (eos != CDFT_EOS_80 && eos != CDFT_EOS_TEOS10)
        return fail(CDFT_ERR_ARG, "cdft_eos_point: eos must be CDFT_EOS_80 or CDFT_EOS_TEOS10, not %d", eos);
    if (!out) return fail(CDFT_ERR_ARG, "cdft_eos_point: NULL argument");
    CDFT_TRY(require_device());
    CDFT_TRY(g_out.reserve(sizeof(double) + 16));
    if (quantity == CDFT_EOSQ_BETA)
        hipLaunchKernelGGL(eos_point_kernel<true>, dim3(1), dim3(64), 0, nullptr, ptem, psal, pdep, eos == CDFT_EOS_TEOS10, (double *)g_out.ptr);
    else
        hipLaunchKernelGGL(eos_point_kernel<false>, dim3(1), dim3(64), 0, nullptr, ptem, psal, pdep, eos == CDFT_EOS_TEOS10, (double *)g_out.ptr);
    CDFT_LAUNCH_CHECK();
    CDFT_HIP(hipMemcpy(out, g_out.ptr, sizeof(double), hipMemcpyDeviceToHost));
    return CDFT_OK;
}

int cdft_eos_more_release(void) {
    for (auto &b : g_in) b.release();
    g_out.release();
    return CDFT_OK;
}

}  // extern "C"
