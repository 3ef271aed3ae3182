// eos.hip -- pointwise equation of state: sigma0 / sigmai (C ABI section "eos").
// Reference: src/eos.f90:601-632 (sigma0), :802-887 (sigmai_dep), :1002-1087 (sigmai_dep2d),
// :889-1000 / :1089-1201 (Jackett-McDougall 1994 variants), coefficients from eos_init :207-598;
// callers src/cdfsig0.f90:146-152 and src/cdfsigi.f90:131-137.
//
// r8 arithmetic on widened r4 inputs, ~120 r8 operations per point for 12-16 bytes of traffic:
// without FMA contraction the kernel sits at the fp64-vector ridge, so the exact
// variant (same operation order and rounding as the reference, bit-identical results) is offered
// next to a contracted one (CDFT_EOS_FAST_FMA, a few ulp of r8, invisible after the r4 store).
#include "eos_dev.hpp"
#include <cstdlib>

using namespace cdft;

namespace {

template <int N>
__device__ __forceinline__ double horner(const double (&c)[N], double t) {
    double a = c[0];
#pragma unroll
    for (int i = 1; i < N; ++i) a = __dadd_rn(__dmul_rn(a, t), c[i]);
    return a;
}

// Jackett & McDougall 1994 (src/eos.f90:936-995); note SQRT(ABS(psal)) is an r4 expression (:931).
__device__ __forceinline__ double jmd94_point(float ptem, float psal, float pref) {
    constexpr double c_r1[] = {6.536332e-9, -1.120083e-6, 1.001685e-4, -9.095290e-3, 6.793952e-2, 999.842594};
    constexpr double c_r2[] = {5.3875e-9, -8.2467e-7, 7.6438e-5, -4.0899e-3, 0.824493};
    constexpr double c_r3[] = {-1.6546e-6, 1.0227e-4, -5.72466e-3};
    constexpr double c_e[]  = {-3.508914e-8, -1.248266e-8, -2.595994e-6};
    constexpr double c_bw[] = {1.296821e-6, -5.782165e-9, 1.045941e-4};
    constexpr double c_c[]  = {-7.267926e-5, 2.598241e-3, 0.1571896};
    constexpr double c_aw[] = {5.939910e-6, 2.512549e-3, -0.1028859, -4.721788};
    constexpr double c_b1[] = {-0.1909078, 7.390729, -55.87545};
    constexpr double c_a1[] = {2.326469e-3, 1.553190, -65.00517, 1044.077};
    constexpr double c_kw[] = {-1.361629e-4, -1.852732e-2, -30.41638, 2098.925, 190925.6};
    const double ref = (double)pref;
    // r4 square root, correctly rounded: an r8 sqrt rounded to r4 is exact for sqrt (53 >= 2*24+2)
    const double rs  = (double)(float)__dsqrt_rn((double)fabsf(psal));
    const double t = (double)ptem, s = (double)psal;
    const double r1 = horner(c_r1, t), r2 = horner(c_r2, t), r3 = horner(c_r3, t);
    const double rhop = __dadd_rn(__dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn(4.8314e-4, s), __dmul_rn(r3, rs)), r2), s), r1);
    const double e = horner(c_e, t), bw = horner(c_bw, t);
    const double b = __dadd_rn(bw, __dmul_rn(e, s));
    const double c = horner(c_c, t), aw = horner(c_aw, t);
    const double a = __dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(-2.042967e-2, rs), c), s), aw);
    const double b1 = horner(c_b1, t), a1 = horner(c_a1, t), kw = horner(c_kw, t);
    const double k0 = __dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(b1, rs), a1), s), kw);
    const double den = __dadd_rn(1.0, -__ddiv_rn(ref, __dadd_rn(k0, -__dmul_rn(ref, __dadd_rn(a, -__dmul_rn(ref, b))))));
    return __dadd_rn(__ddiv_rn(rhop, den), -1000.0);
}

enum Kind { kScalarRef = 0, kPointRef = 1, kTool = 2 };

struct EosArgs {
    const float *t, *s, *p2d;
    double *out8;
    float  *out4;
    int64_t n;
    float pref, spval;
    int no_land_shortcut;      // CDFT_EOS_NOLAND=1 (A/B and test hook): every point through the polynomials
};

template <int EOS, bool FAST, int DEPTH = 0>
__device__ __forceinline__ double eval_point(float t, float s, float p) {
    if (EOS == 2) return jmd94_point(t, s, p);
    if (EOS == 3) return sigmantr_point(t, s);                     // neutral density: no reference depth, no mask
    return sigmai_point<(EOS == 1 ? 1 : 0), FAST, DEPTH>(t, s, p);
}

// VEC = 4: every pointer 16-byte aligned, four points per lane per step (dwordx4 loads);
// VEC = 1: scalar fallback for unaligned arrays.  Grid-stride over vectors.
//
// Land.  NEMO writes T = S = 0 on land, and land comes in large pieces: a wave (256 consecutive points) that finds +0 in every
// T and S it loaded stores sigma(0, 0, pref) -- evaluated ONCE per lane before the loop by the same code, so the stored bits
// are those the full evaluation gives (for sigma0 that is -0.0: the reference's r * tmask with r = -0.16, src/eos.f90:630,
// and the tools' zsig * zmask, src/cdfsig0.f90:150) -- and moves on without touching the polynomials.  The branch is uniform
// over the wave (a ballot), so an ocean wave pays two OR / compare instructions per point.  A wave holding any other bit
// pattern -- an ocean point, a fill value, -0.0, NaN -- takes the full path.  Not for per-point reference depths (kPointRef).
// DEPTH 1 / 2: one reference depth for the launch, known to be 0 / not 0 (eos_dev.hpp: the same arithmetic without the joins
// of the branch on it); 0: looked at per point.
template <int EOS, bool FAST, int KIND, int VEC, int DEPTH>
__global__ void __launch_bounds__(256) eos_kernel(EosArgs a) {
    const int64_t nvec = a.n / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool same_mask = (EOS == 0 || EOS == 1) && KIND == kTool && a.spval == 0.f && !a.no_land_shortcut;   // (uniform; the hook keeps the written-out form too)
    double r_land = 0.0;
    float  o_land = 0.f;
    if (KIND != kPointRef && !a.no_land_shortcut) {
        r_land = eval_point<EOS, FAST, DEPTH>(0.f, 0.f, a.pref);
        o_land = (float)__dmul_rn(r_land, (double)((0.f == a.spval) ? 0.f : 1.f));
    }
    // the loads of the NEXT vector are issued before the current one is evaluated (3 % at eORCA12: the kernel is bound by
    // the vector pipeline -- counters: VALU busy 96 % of the time for sigma2, ~141 vector instructions per point of which 119
    // are r8 -- so a load that is not waited for is all there is to gain; writing the four points' Horner chains side by
    // side in the source changed nothing, the scheduler interleaves them by itself)
    f4a nt = {0.f, 0.f, 0.f, 0.f}, ns = nt, np_ = nt;
    float nt1 = 0.f, ns1 = 0.f, np1 = 0.f;
    const int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v0 < nvec) {
        if (VEC == 4) {
            nt = ld_stream4(a.t + 4 * v0); ns = ld_stream4(a.s + 4 * v0);
            if (KIND == kPointRef) np_ = ld_stream4(a.p2d + 4 * v0);
        } else {
            nt1 = a.t[v0]; ns1 = a.s[v0];
            if (KIND == kPointRef) np1 = a.p2d[v0];
        }
    }
    for (int64_t v = v0; v < nvec; v += stride) {
        float t[VEC], s[VEC], p[VEC];
        if (VEC == 4) {
            t[0] = nt.x; t[1] = nt.y; t[2] = nt.z; t[3] = nt.w;
            s[0] = ns.x; s[1] = ns.y; s[2] = ns.z; s[3] = ns.w;
            if (KIND == kPointRef) { p[0] = np_.x; p[1] = np_.y; p[2] = np_.z; p[3] = np_.w; }
        } else {
            t[0] = nt1; s[0] = ns1;
            if (KIND == kPointRef) p[0] = np1;
        }
        if (v + stride < nvec) {
            const int64_t w = v + stride;
            if (VEC == 4) {
                nt = ld_stream4(a.t + 4 * w); ns = ld_stream4(a.s + 4 * w);
                if (KIND == kPointRef) np_ = ld_stream4(a.p2d + 4 * w);
            } else {
                nt1 = a.t[w]; ns1 = a.s[w];
                if (KIND == kPointRef) np1 = a.p2d[w];
            }
        }
        if (KIND != kPointRef && !a.no_land_shortcut) {
            uint32_t bits = 0;
#pragma unroll
            for (int c = 0; c < VEC; ++c) bits |= __float_as_uint(t[c]) | __float_as_uint(s[c]);
            if (!__any(bits != 0u)) {                    // every lane of the wave: T = S = +0 on all its points
                if (KIND == kTool) {
                    if (VEC == 4) reinterpret_cast<float4 *>(a.out4)[v] = make_float4(o_land, o_land, o_land, o_land);
                    else a.out4[v] = o_land;
                } else if (VEC == 4) {
                    reinterpret_cast<double2 *>(a.out8)[2 * v]     = make_double2(r_land, r_land);
                    reinterpret_cast<double2 *>(a.out8)[2 * v + 1] = make_double2(r_land, r_land);
                } else {
                    a.out8[v] = r_land;
                }
                continue;
            }
        }
        double r[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c)
            r[c] = eval_point<EOS, FAST, DEPTH>(t[c], s[c], KIND == kPointRef ? p[c] : a.pref);
        if (KIND == kTool) {
            float o[VEC];
            if (same_mask) {
                // spval = 0 (the usual missing value of S): the tool's mask (S == spval ? 0 : 1, cdfsig0.f90:148-150) is the 0 / 1
                // sigmai has already multiplied by (S == 0 ? 0 : 1, eos.f90:852-856, :880).  (x * m) * m = x * m bit for bit for
                // m = 1 (exact) and m = 0 (the zero, or the NaN of a non-finite x, keeps its sign through the second product):
                // the second product, its compare, select and conversion are left out -- 4 of ~86 instructions per point
#pragma unroll
                for (int c = 0; c < VEC; ++c) o[c] = (float)r[c];
            } else {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {   // cdfsig0.f90:148-150: r8 * r4 mask, stored into r4
                    const float zmask = (s[c] == a.spval) ? 0.f : 1.f;
                    o[c] = (float)__dmul_rn(r[c], (double)zmask);
                }
            }
            if (VEC == 4) { f4a ov = {o[0], o[1], o[2], o[3]}; __builtin_nontemporal_store(ov, reinterpret_cast<f4a *>(a.out4) + v); }
            else a.out4[v] = o[0];
        } else {
            if (VEC == 4) {
                reinterpret_cast<double2 *>(a.out8)[2 * v]     = make_double2(r[0], r[1]);
                reinterpret_cast<double2 *>(a.out8)[2 * v + 1] = make_double2(r[2], r[3]);
            } else {
                a.out8[v] = r[0];
            }
        }
    }
    // tail of a vectorised launch: < 4 points, one lane each
    if (VEC == 4 && blockIdx.x == 0 && threadIdx.x < (unsigned)(a.n - nvec * VEC)) {
        const int64_t q = nvec * VEC + threadIdx.x;
        const double r = eval_point<EOS, FAST, DEPTH>(a.t[q], a.s[q], KIND == kPointRef ? a.p2d[q] : a.pref);
        if (KIND == kTool) {
            const float zmask = (a.s[q] == a.spval) ? 0.f : 1.f;
            a.out4[q] = (float)__dmul_rn(r, (double)zmask);
        } else {
            a.out8[q] = r;
        }
    }
}

template <int EOS, bool FAST, int KIND>
int launch_kind(const EosArgs &a, hipStream_t st) {
    if (a.n == 0) return CDFT_OK;
    auto al = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec = al(a.t) && al(a.s) && al(a.p2d) && al(a.out8) && al(a.out4);
    const int threads = 256;
    const int64_t work = vec ? (a.n + 3) / 4 : a.n;
    int64_t blocks = (work + threads - 1) / threads;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    // the surface form as its own kernel (sigma0: 94 -> 90 vector instructions per point, 3.85 -> 3.6 ms at eORCA12); the deep
    // form keeps the uniform branch: specialised too it came out at 146 instructions per point instead of 143 (measured)
    if (vec && KIND != kPointRef && (EOS == 0 || EOS == 1) && a.pref == 0.f) {
        hipLaunchKernelGGL((eos_kernel<EOS, FAST, KIND, 4, 1>), dim3((unsigned)blocks), dim3(threads), 0, st, a);
    } else if (vec) {
        hipLaunchKernelGGL((eos_kernel<EOS, FAST, KIND, 4, 0>), dim3((unsigned)blocks), dim3(threads), 0, st, a);
    } else {
        hipLaunchKernelGGL((eos_kernel<EOS, FAST, KIND, 1, 0>), dim3((unsigned)blocks), dim3(threads), 0, st, a);
    }
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

template <int KIND>
int launch_eos(const EosArgs &a, int eos, int flags, hipStream_t st) {
    const bool fast = (flags & CDFT_EOS_FAST_FMA) != 0;
    switch (eos) {
        case CDFT_EOS_80:     return fast ? launch_kind<0, true, KIND>(a, st) : launch_kind<0, false, KIND>(a, st);
        case CDFT_EOS_TEOS10: return fast ? launch_kind<1, true, KIND>(a, st) : launch_kind<1, false, KIND>(a, st);
        case CDFT_EOS_JMD94:  return launch_kind<2, false, KIND>(a, st);
        case CDFT_EOS_NTR:    return launch_kind<3, false, KIND>(a, st);
        default: return fail(CDFT_ERR_ARG, "unknown equation of state %d", eos);
    }
}

// staging for host arrays (the Fortran tools pass getvar slabs); single host thread, as the reference
DevBuf g_t, g_s, g_p, g_o;

int run(int kind, const float *ptem, const float *psal, const float *pref2d, int64_t n, float pref,
        float spval, int eos, int flags, void *out, size_t out_elem, int in_space, int out_space,
        void *stream) {
    if (n < 0) return fail(CDFT_ERR_ARG, "cdft_eos: negative point count");
    if (n == 0) return CDFT_OK;                                   // empty slab: nothing to do
    if (!ptem || !psal || !out || (kind == kPointRef && !pref2d))
        return fail(CDFT_ERR_ARG, "cdft_eos: NULL argument");
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const void *dt, *ds, *dp = nullptr;
    CDFT_TRY(to_device(ptem, n * sizeof(float), in_space, g_t, st, &dt));
    CDFT_TRY(to_device(psal, n * sizeof(float), in_space, g_s, st, &ds));
    if (kind == kPointRef) CDFT_TRY(to_device(pref2d, n * sizeof(float), in_space, g_p, st, &dp));
    void *dout = out;
    if (out_space == CDFT_MEM_HOST || out_space == CDFT_MEM_PINNED) {
        CDFT_TRY(g_o.reserve(n * out_elem + 16));
        dout = g_o.ptr;
    } else if (out_space != CDFT_MEM_DEVICE) {
        return fail(CDFT_ERR_ARG, "cdft_eos: bad memory-space tag %d", out_space);
    }
    EosArgs a;
    a.t = (const float *)dt; a.s = (const float *)ds; a.p2d = (const float *)dp;
    a.out8 = kind == kTool ? nullptr : (double *)dout;
    a.out4 = kind == kTool ? (float *)dout : nullptr;
    a.n = n; a.pref = pref; a.spval = spval;
    { const char *e = getenv("CDFT_EOS_NOLAND"); a.no_land_shortcut = (e && e[0] == '1') ? 1 : 0; }
    int rc;
    if (kind == kScalarRef)     rc = launch_eos<kScalarRef>(a, eos, flags, st);
    else if (kind == kPointRef) rc = launch_eos<kPointRef>(a, eos, flags, st);
    else                        rc = launch_eos<kTool>(a, eos, flags, st);
    CDFT_TRY(rc);
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(out, dout, n * out_elem, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    } else if (out_space == CDFT_MEM_PINNED) {                    // on its way when the call returns (stream order keeps the
        CDFT_HIP(hipMemcpyAsync(out, dout, n * out_elem, hipMemcpyDeviceToHost, st));   // staging buffer safe from the next call)
    }
    return CDFT_OK;
}

}  // namespace

extern "C" {

int cdft_eos_sigmai(const float *ptem, const float *psal, int64_t n, float pref, int eos, int flags,
                    double *sig, int in_space, int out_space, void *stream) {
    TraceRange trace_("cdft_eos_sigmai");
    return run(kScalarRef, ptem, psal, nullptr, n, pref, 0.f, eos, flags, sig, sizeof(double), in_space,
               out_space, stream);
}

int cdft_eos_sigmai2d(const float *ptem, const float *psal, const float *pref2d, int64_t n, int eos,
                      int flags, double *sig, int in_space, int out_space, void *stream) {
    return run(kPointRef, ptem, psal, pref2d, n, 0.f, 0.f, eos, flags, sig, sizeof(double), in_space,
               out_space, stream);
}

int cdft_eos_sig_tool(const float *ptem, const float *psal, int64_t n, float pref, float spval, int eos,
                      int flags, float *sig4, int in_space, int out_space, void *stream) {
    TraceRange trace_("cdft_eos_sig_tool");
    return run(kTool, ptem, psal, nullptr, n, pref, spval, eos, flags, sig4, sizeof(float), in_space,
               out_space, stream);
}

int cdft_eos_release(void) {
    g_t.release(); g_s.release(); g_p.release(); g_o.release();
    return CDFT_OK;
}

}  // extern "C"
