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
#include <cstdlib>
#include "eos_dev.hpp"

using namespace cdft;

namespace {

DevBuf g_mi[8], g_mo[2];

struct SigArgs {
    const float *zv, *zeiv, *zt, *zs, *e1v, *e3v, *e31d, *gdep;
    const int16_t *ib;                 // (nb,npi,npj)
    int npi, npj, npk, nb, nbins, j0, j1;
    float zspv, zspt, zsps, pref, sigmin, sigstp;
    double rsigstp;                    // 1 / sigstp in r8 (class kernel: the r4 division by sigstp as an r8 product)
    double *dmoc, *depi;               // (nb,nbins,npj)
    int no_land_shortcut;              // CDFT_MOCSIG_NOLAND=1: the equation of state on every cell (A/B and test hook)
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
                const double dens = mocsig_density<TEOS>(t, s, a.pref);                    // :395
                const float zttmp = (float)__dmul_rn(dens, (double)itm);                   // :398
                int ibin = (int)__fdiv_rn(__fsub_rn(zttmp, a.sigmin), a.sigstp);           // :400
                ibin = min(max(ibin, 1), a.nbins) - 1;                                     // :401-402
                const double term = -(double)__fmul_rn(v, zarea);                          // :417
                const double dd = ISODEP ? (double)__fmul_rn(__fmul_rn(gd, (float)itm), zarea) : 0.0;   // :424
                const double dw = ISODEP ? (double)__fmul_rn((float)itm, zarea) : 0.0;                   // :425
                for (int b = 0; b < a.nb; ++b) {
                    const int16_t m = a.ib[p * a.nb + b];
                    const double fm = (double)m, add = __dmul_rn(term, fm);
                    if (m == 0 && add == 0.0) continue;                                    // an exact zero (0*NaN is not: kept)
                    (void)__hip_atomic_fetch_add(&h[b * a.nbins + ibin], add, __ATOMIC_RELAXED,
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

// ---------------------------------------------------------------------------------------------------------------
// Class kernel (0/1 basin masks with at most kSC bit patterns, -eiv as a fifth stream, -full's e3t_1d in place of e3v, -isodep as two more histograms; rows of 4..5116 cells).
// The loop above spends a scalar load per operand and cell, re-reads e1v and the basin masks at every level and
// issues one LDS atomic per cell AND basin.  Here a thread owns NV 16-byte vectors of the row for all levels: e1v
// and the class of each cell's basin bit pattern (see cdfmoc's class kernel, moc_zonal.hpp) stay in registers,
// v / T / S / e3v arrive as nontemporal 16-byte loads, and a cell's term goes to ONE (class, bin) accumulator of the
// wave's LDS histogram.  Neighbouring cells mostly fall into the same density bin: the four cells of a vector are
// merged in registers first (left to right), so real fields issue one atomic per vector, and the conflicts of a
// whole wave adding into one bin stay hidden behind the equation of state.  A term that is an exact zero (land:
// v = 0) is not added -- it could not change the sum -- which keeps the land cells out of bin 1 (:401) altogether.
// Basin b = the classes whose pattern has bit b, + 0 * (the other classes): the NaN of the reference's 0*NaN, or 0.
constexpr int kSC = 8;

struct SigClassArgs {
    const uint8_t *cls;            // (npi,npj) class of the cell's basin bit pattern
    uint8_t pat[kSC];
    int nclass;
};

// OPT: 0 = v, T, S, e3v and nothing else (the plain kernel: every option below costs it 1-6 % even as a uniform branch, measured);
//      1 = -eiv and / or -full (uniform branches on a.zeiv, a.e3v);  2 = the same with -isodep;
//      3 = the plain kernel when the three missing values zspv, zspt, zsps are all 0 (what NEMO files carry) and the density is
//          sigmai: "v == 0 ? 0 : v" and "S == 0 ? 0 : S" only turn -0.0 into +0.0, which changes nothing downstream (a zero term
//          is never added; sigmai takes |S + dS| and tests S == 0), and the mask (S == 0 ? 0 : 1) is the one sigmai has
//          already multiplied by -- (x*m)*m = x*m bit for bit for m in {0,1}: eight vector instructions per cell fewer
template <int NV, int TEOS, int T, int DEPTH, int OPT>
__global__ void __launch_bounds__(T) mocsig_class_kernel(SigArgs a, SigClassArgs ca) {
    // [nwaves][nq][nclass][nbins], then the basin sums raw [nb][nbins]; with -isodep (nq = 3: the transport, depth * weight
    // and weight sums of :435, :444-445) raw [nq][nb][nbins] is formed in the place of wave 1's histogram, after the waves
    // have been added into wave 0's: see sig_class_shmem()
    extern __shared__ double hist[];
    constexpr bool EXTRA = OPT == 1 || OPT == 2, ISODEP = OPT == 2, Z0 = OPT == 3;
    constexpr int nwaves = T / 64, nq = ISODEP ? 3 : 1;
    const int jj = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hq = ca.nclass * a.nbins, hsz = nq * hq;
    double *h = hist + (int64_t)wave * hsz, *raw = hist + (ISODEP ? (int64_t)hsz : (int64_t)nwaves * hsz);
    for (int t = lane; t < hsz; t += 64) h[t] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t nij = (int64_t)a.npi * a.npj, r2 = (int64_t)jj * a.npi;
    if (jj >= a.j0 && jj <= a.j1) {                     // rows ij1..ij2 (:405-407); the others stay 0
        const int nfull = a.npi >> 2;
        int i0[NV];
        float e1[NV][4];
        int hoff[NV][4];                                // class * nbins, or -1 for a cell this lane does not own
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int v = threadIdx.x + r * T;
            i0[r] = min(4 * v, a.npi - 4);              // vectors 0..nfull-1 start at 4v, one more is anchored at npi-4
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = i0[r] + c;
                const bool own = (v < nfull || (v == nfull && i >= 4 * nfull)) && i >= 1 && i < a.npi - 1;   // ji = 2..npiglo-1
                e1[r][c]   = a.e1v[r2 + i];
                hoff[r][c] = own ? (int)ca.cls[r2 + i] * a.nbins : -1;
            }
        }
        // one vector (v, T, S, e3v: four 16-byte loads) is in flight while the previous one goes through the equation of
        // state: 32 registers of loads instead of 16*NV, which keeps the kernel at three waves per SIMD (it is bound by the
        // r8 issue rate and the latency of its Horner chains, not by HBM: more waves, not more bytes in flight)
        f4u cv, ct, cs, ce, cei = {0.f, 0.f, 0.f, 0.f};         // (cei: the eddy-induced velocity of -eiv, a.zeiv != NULL: uniform)
        {
            const int64_t e = r2 + i0[0];
            cv = ld_stream4u(a.zv + e); ct = ld_stream4u(a.zt + e); cs = ld_stream4u(a.zs + e);
            if (!EXTRA || a.e3v) ce = ld_stream4u(a.e3v + e);
            else { const float z = a.e31d[0]; ce = f4u{z, z, z, z}; }          // -full: the level's thickness (:389), uniform
            if (EXTRA && a.zeiv) cei = ld_stream4u(a.zeiv + e);
        }
        for (int k = 0; k < a.npk - 1; ++k) {
            const int64_t e = (int64_t)k * nij + r2;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                f4u nv_ = cv, nt_ = ct, ns_ = cs, ne_ = ce, nei_ = cei;
                if (r + 1 < NV || k + 1 < a.npk - 1) {
                    const int64_t en = (r + 1 < NV ? e : e + nij) + i0[r + 1 < NV ? r + 1 : 0];
                    nv_ = ld_stream4u(a.zv + en); nt_ = ld_stream4u(a.zt + en); ns_ = ld_stream4u(a.zs + en);
                    if (!EXTRA || a.e3v) ne_ = ld_stream4u(a.e3v + en);
                    else if (r + 1 == NV) { const float z = a.e31d[k + 1]; ne_ = f4u{z, z, z, z}; }
                    if (EXTRA && a.zeiv) nei_ = ld_stream4u(a.zeiv + en);
                }
                const float v4[4] = {cv.x, cv.y, cv.z, cv.w}, t4[4] = {ct.x, ct.y, ct.z, ct.w};
                const float s4[4] = {cs.x, cs.y, cs.z, cs.w}, e4[4] = {ce.x, ce.y, ce.z, ce.w};
                int key[4];
                double val[4], vdd[4], vdw[4];                  // (vdd, vdw: -isodep only)
                float vz[4];
                const float gd = ISODEP ? a.gdep[k] : 0.f;
                // The term of every cell first (:374, :389, :417): -(v * (e1v * e3v)) in r4, widened.
                bool live = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    vz[c] = Z0 ? v4[c] : ((v4[c] == a.zspv) ? 0.f : v4[c]);                    // :374
                    if (EXTRA && a.zeiv) vz[c] = __fadd_rn(vz[c], c == 0 ? cei.x : c == 1 ? cei.y : c == 2 ? cei.z : cei.w);   // :375-378
                    const float zarea = __fmul_rn(e1[r][c], e4[c]);                            // :389
                    val[c] = -(double)__fmul_rn(vz[c], zarea);                                 // :417
                    if (ISODEP) {                                                              // :392-393 with :382-383, :424-425
                        const float fitm = ((s4[c] == a.zsps ? 0.f : s4[c]) == a.zsps) ? 0.f : 1.f;
                        vdd[c] = (double)__fmul_rn(__fmul_rn(gd, fitm), zarea);
                        vdw[c] = (double)__fmul_rn(fitm, zarea);
                        live |= (hoff[r][c] >= 0) && (val[c] != 0.0 || vdd[c] != 0.0 || vdw[c] != 0.0);
                    } else {
                        live |= (hoff[r][c] >= 0) && (val[c] != 0.0);                          // (NaN != 0: kept)
                    }
                }
                // Land.  A term that is exactly zero (v = 0 or e3v = 0: land, and it comes in large pieces) is never added
                // (below), so its density class is never looked at: a wave whose 256 cells all have such a term skips the
                // equation of state altogether -- ~115 r8 operations per cell.  Uniform over the wave (a ballot); a NaN
                // term is not zero and takes the full path.  CDFT_MOCSIG_NOLAND=1: A/B and test hook.
                if (!a.no_land_shortcut && !__any(live)) { cv = nv_; ct = nt_; cs = ns_; ce = ne_; cei = nei_; continue; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float t = t4[c], s = s4[c];
                    if (t == a.zspt) t = 0.f;                                                  // :382
                    if (!Z0 && s == a.zsps) s = 0.f;                                           // :383
                    const int itm = (s == a.zsps) ? 0 : 1;                                     // :392-393
                    const double dens = mocsig_density<TEOS, DEPTH>(t, s, a.pref);             // :395
                    const float zttmp = Z0 ? (float)dens : (float)__dmul_rn(dens, (double)itm);   // :398
                    // :400  (zttmp - sigmin) / sigstp in r4, correctly rounded: the quotient of two r4 numbers is never closer
                    // than 2^-49 (relative) to the middle between two r4 numbers unless it is one, so the r8 product with the r8
                    // reciprocal of sigstp (error < 2^-52) rounds to the same r4 as the division -- three instructions, not ten
                    const float zq = (float)__dmul_rn((double)__fsub_rn(zttmp, a.sigmin), a.rsigstp);
                    int ibin = (int)zq;
                    ibin = min(max(ibin, 1), a.nbins) - 1;                                     // :401-402
                    // (written as one short-circuit expression: with the test of val[c] hoisted into a variable the compiler
                    // emits 8 % more instructions for this loop and spills more scalars -- measured, 6.9 -> 7.5 ms)
                    key[c] = (hoff[r][c] >= 0 && (val[c] != 0.0 || (ISODEP && (vdd[c] != 0.0 || vdw[c] != 0.0)))) ? hoff[r][c] + ibin : -1;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (key[c] >= 0 && key[c] == key[c + 1]) {
                        val[c + 1] = __dadd_rn(val[c + 1], val[c]);
                        if (ISODEP) { vdd[c + 1] = __dadd_rn(vdd[c + 1], vdd[c]); vdw[c + 1] = __dadd_rn(vdw[c + 1], vdw[c]); }
                        key[c] = -1;
                    }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (key[c] >= 0) {                  // (an exact zero is not added: it could not change a sum that starts at +0)
                        if (!ISODEP || val[c] != 0.0)
                            (void)__hip_atomic_fetch_add(&h[key[c]], val[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // :435
                        if (ISODEP && vdd[c] != 0.0)
                            (void)__hip_atomic_fetch_add(&h[hq + key[c]], vdd[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // :444
                        if (ISODEP && vdw[c] != 0.0)
                            (void)__hip_atomic_fetch_add(&h[2 * hq + key[c]], vdw[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // :445
                    }
                cv = nv_; ct = nt_; cs = ns_; ce = ne_; cei = nei_;
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < hsz; t += T) {                // waves added in wave order
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += hist[(int64_t)w * hsz + t];
        hist[t] = s;                                            // wave 0's slot t is only read by this thread
    }
    __syncthreads();
    const int rq = a.nb * a.nbins;
    for (int t = threadIdx.x; t < nq * rq; t += T) {           // class sums -> basin sums
        const int q = t / rq, u = t - q * rq, b = u / a.nbins, n = u - b * a.nbins;
        double s = 0.0, z = 0.0;
        for (int c = 0; c < ca.nclass; ++c) {
            const double v = hist[q * hq + c * a.nbins + n];
            if ((ca.pat[c] >> b) & 1) s += v;
            else                      z += v;
        }
        raw[t] = s + __dmul_rn(z, 0.0);
    }
    __syncthreads();
    if ((int)threadIdx.x < a.nb) {                              // :463-467: from the highest density down
        const int b = threadIdx.x;
        const double *hm = raw + b * a.nbins;
        double acc = __ddiv_rn(hm[a.nbins - 1], (double)1.e6f);
        a.dmoc[((int64_t)jj * a.nbins + (a.nbins - 1)) * a.nb + b] = acc;
        for (int n = a.nbins - 2; n >= 0; --n) {
            acc = __dadd_rn(acc, __ddiv_rn(hm[n], (double)1.e6f));
            a.dmoc[((int64_t)jj * a.nbins + n) * a.nb + b] = acc;
        }
    }
    if (ISODEP) {                                               // :454-460
        for (int t = threadIdx.x; t < rq; t += T) {
            const int b = t / a.nbins, n = t - b * a.nbins;
            const double d = raw[rq + t], w = raw[2 * rq + t];
            a.depi[((int64_t)jj * a.nbins + n) * a.nb + b] = (w != 0.0) ? __ddiv_rn(d, w) : 99999.0;
        }
    }
}

// bits(i,j) = basin membership bits; seen[1+v] = 1 for every pattern v present; seen[0] raised by a value outside {0,1}
__global__ void sig_bits_kernel(const int16_t *__restrict__ ib, int nb, int64_t nij, uint8_t *__restrict__ bits, int *__restrict__ seen) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    uint32_t v = 0;
    bool bad = false;
    for (int b = 0; b < nb; ++b) {
        const int16_t m = ib[p * nb + b];
        bad |= (m != 0 && m != 1);
        v |= (m != 0 ? 1u : 0u) << b;
    }
    bits[p] = (uint8_t)v;
    if (bad) atomicOr(seen, 1);
    if (!seen[1 + v]) seen[1 + v] = 1;                   // benign race: every writer stores 1
}
__global__ void sig_cls_kernel(uint8_t *__restrict__ bits, int64_t nij, const uint8_t *__restrict__ lut) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nij) bits[p] = lut[bits[p]];                 // in place: pattern -> class
}

constexpr int kSigT = 256;

// LDS of the class kernel: the waves' histograms, or wave 0's + the basin sums formed in the place of wave 1's
size_t sig_class_shmem(int nq, int nclass, int nb, int nbins) {
    const size_t hsz = (size_t)nq * nclass * nbins * 8, rawsz = (size_t)nq * nb * nbins * 8;
    return nq == 1 ? (size_t)(kSigT / 64) * hsz + rawsz : std::max((size_t)(kSigT / 64) * hsz, hsz + rawsz);
}

template <int NV, int KIND, int DEPTH, int OPT>
int launch_sig_class_opt(const SigArgs &a, const SigClassArgs &ca, size_t shmem, hipStream_t st) {
    if (shmem > 64 * 1024)
        CDFT_TRY(dyn_lds_once((const void *)mocsig_class_kernel<NV, KIND, kSigT, DEPTH, OPT>, (int)shmem));
    hipLaunchKernelGGL((mocsig_class_kernel<NV, KIND, kSigT, DEPTH, OPT>), dim3(a.npj), dim3(kSigT), shmem, st, a, ca);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

template <int NV, int KIND, int DEPTH>
int launch_sig_class_depth(const SigArgs &a, const SigClassArgs &ca, size_t shmem, hipStream_t st) {
    if (a.depi) return launch_sig_class_opt<NV, KIND, DEPTH, 2>(a, ca, shmem, st);
    if (a.zeiv || !a.e3v) return launch_sig_class_opt<NV, KIND, DEPTH, 1>(a, ca, shmem, st);
    if constexpr (KIND != 2) {                                    // (sigmantr applies no mask of its own)
        if (a.zspv == 0.f && a.zspt == 0.f && a.zsps == 0.f && !getenv("CDFT_MOCSIG_SPVAL_GENERAL"))     // (A/B and test hook)
            return launch_sig_class_opt<NV, KIND, DEPTH, 3>(a, ca, shmem, st);
    }
    return launch_sig_class_opt<NV, KIND, DEPTH, 0>(a, ca, shmem, st);
}

template <int NV, int KIND>
int launch_sig_class_kind(const SigArgs &a, const SigClassArgs &ca, size_t shmem, hipStream_t st) {
    if (KIND == 2) return launch_sig_class_depth<NV, KIND, 0>(a, ca, shmem, st);            // (the neutral density has no depth)
    if (getenv("CDFT_MOCSIG_DEPTH0")) return launch_sig_class_depth<NV, KIND, 0>(a, ca, shmem, st);      // (A/B hook: the uniform branch)
    return a.pref == 0.f ? launch_sig_class_depth<NV, KIND, 1>(a, ca, shmem, st) : launch_sig_class_depth<NV, KIND, 2>(a, ca, shmem, st);
}

template <int NV>
int launch_sig_class(const SigArgs &a, const SigClassArgs &ca, int eos, size_t shmem, hipStream_t st) {
    if (eos == CDFT_EOS_TEOS10) return launch_sig_class_kind<NV, 1>(a, ca, shmem, st);
    if (eos == CDFT_EOS_NTR)    return launch_sig_class_kind<NV, 2>(a, ca, shmem, st);
    return launch_sig_class_kind<NV, 0>(a, ca, shmem, st);
}

}  // namespace

extern "C" {

int cdft_mocsig_compute(const float *zv, const float *zveiv, const float *zt, const float *zs, const float *e1v,
                        const float *e3v, const float *e31d, const int16_t *ibmask, const float *gdep, int npiglo,
                        int npjglo, int npk, int nbasins, float zspv, float zspt, float zsps, float pref, int eos,
                        int nbins, float sigmin, float sigstp, int in_space, double *dmoc, double *depi, int out_space,
                        void *stream) {
    TraceRange trace_("cdft_mocsig_compute");
    if (!zv || !zt || !zs || !e1v || (!e3v && !e31d) || !ibmask || !dmoc) return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: NULL argument");
    if (npiglo < 1 || npjglo < 1 || npk < 2 || nbasins < 1 || nbasins > 8 || nbins < 1 || nbins > 2048 || !(sigstp > 0.f))
        return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: bad sizes or binning (%d x %d x %d, %d basins, %d bins)", npiglo, npjglo,
                    npk, nbasins, nbins);
    if (depi && !gdep) return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: -isodep needs gdep");
    if (eos != CDFT_EOS_80 && eos != CDFT_EOS_TEOS10 && eos != CDFT_EOS_NTR)
        return fail(CDFT_ERR_ARG, "cdft_mocsig_compute: eos must be CDFT_EOS_80, CDFT_EOS_TEOS10 or CDFT_EOS_NTR");
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
    a.rsigstp = 1.0 / (double)sigstp;
    a.dmoc = om; a.depi = od;
    { const char *e = getenv("CDFT_MOCSIG_NOLAND"); a.no_land_shortcut = (e && e[0] == '1') ? 1 : 0; }
    // class kernel when it applies (see above); CDFT_MOCSIG_NOCLASS=1 keeps the general kernel (A/B and test hook)
    bool done = false;
    const int nvec = npiglo / 4 + 1, nv = (nvec + kSigT - 1) / kSigT;
    if (npiglo >= 4 && nv <= 5 && !getenv("CDFT_MOCSIG_NOCLASS")) {
        static DevBuf sbits, sseen;
        CDFT_TRY(sbits.reserve(nij + 64 + 256));
        CDFT_TRY(sseen.reserve(257 * sizeof(int)));
        uint8_t *bits = (uint8_t *)sbits.ptr;
        int *seen = (int *)sseen.ptr;
        CDFT_HIP(hipMemsetAsync(seen, 0, 257 * sizeof(int), st));
        hipLaunchKernelGGL(sig_bits_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, (const int16_t *)dib, nbasins,
                           (int64_t)nij, bits, seen);
        CDFT_LAUNCH_CHECK();
        int hseen[257];
        CDFT_HIP(hipMemcpyAsync(hseen, seen, sizeof(hseen), hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
        SigClassArgs ca;
        ca.nclass = 0;
        uint8_t lut[256];
        for (int v = 0; v < 256; ++v) {
            lut[v] = 0;
            if (hseen[1 + v]) {
                if (ca.nclass < kSC) { ca.pat[ca.nclass] = (uint8_t)v; lut[v] = (uint8_t)ca.nclass; }
                ++ca.nclass;
            }
        }
        const size_t shmem = sig_class_shmem(depi ? 3 : 1, std::min(ca.nclass, kSC), nbasins, nbins);
        if (!hseen[0] && ca.nclass <= kSC && shmem <= 150 * 1024) {
            uint8_t *dlut = bits + nij + 64;
            CDFT_HIP(hipMemcpyAsync(dlut, lut, 256, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(sig_cls_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, bits, (int64_t)nij, dlut);
            CDFT_LAUNCH_CHECK();
            CDFT_HIP(hipStreamSynchronize(st));                  // lut[] is a stack array
            ca.cls = bits;
            if (nv <= 2) CDFT_TRY(launch_sig_class<2>(a, ca, eos, shmem, st));
            else         CDFT_TRY(launch_sig_class<5>(a, ca, eos, shmem, st));
            done = true;
        }
    }
    if (!done) {
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
    } else if (eos == CDFT_EOS_NTR) {
        if (depi) hipLaunchKernelGGL((mocsig_row_kernel<2, true>), grid, block, shmem, st, a);
        else      hipLaunchKernelGGL((mocsig_row_kernel<2, false>), grid, block, shmem, st, a);
    } else {
        if (depi) hipLaunchKernelGGL((mocsig_row_kernel<0, true>), grid, block, shmem, st, a);
        else      hipLaunchKernelGGL((mocsig_row_kernel<0, false>), grid, block, shmem, st, a);
    }
    CDFT_LAUNCH_CHECK();
    }
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
