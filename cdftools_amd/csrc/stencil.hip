// stencil.hip -- the two C-grid stencil tools: cdfvT and cdfcurl (C ABI sections "cdfvT", "cdfcurl").
//
//   cdfvT   src/cdfvT.f90:160-259   T,S interpolated to U/V points (2-point means along i / j),
//                                   times u / v, accumulated over every time frame, per cell
//   cdfcurl src/cdfcurl.f90:262-304 curl of (u,v) at F points from the 4 surrounding U/V points
//
// Both are pointwise after the stencil: every product is formed in r4 exactly as the reference
// writes it and (cdfvT) accumulated over frames in the reference's order, so results are
// BIT-IDENTICAL to the reference loop.
//
// Layout of the work: one workgroup = one 1024-cell tile of one (jj,jk) row; lanes own 4 consecutive
// cells (one 16-byte load per field; rows are only 4-byte aligned in general, so the loads are
// declared with 4-byte alignment).  The i+1 neighbour of a lane's last cell belongs to the next
// lane: the first cell of every lane is staged through LDS (plus one halo cell per tile).  The j+1
// row is a second coalesced load; the block -> tile map gives every XCD a contiguous band of rows so
// that the row read as "j+1" is still in that XCD's L2 when it is read again as "j".
#include "common.hpp"
#include "shard.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

using namespace cdft;

namespace {

constexpr int kThreads = 256;
constexpr int kTile    = kThreads * 4;

typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

// nv valid cells (1..4) starting at p; cells beyond the row are returned as 0 and never touched
__device__ __forceinline__ void load4(const float *p, int nv, float (&o)[4]) {
    if (nv == 4) {
        const f4u v = *reinterpret_cast<const f4u *>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = (c < nv) ? p[c] : 0.f;
    }
}

// the same for a field no other row reads again (nontemporal, common.hpp)
__device__ __forceinline__ void load4s(const float *p, int nv, float (&o)[4]) {
    if (nv == 4) {
        const f4u v = ld_stream4u(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = (c < nv) ? p[c] : 0.f;
    }
}

// results are written once and never read back by these kernels: full vectors, nontemporal
__device__ __forceinline__ void store4(float *p, int nv, const double (&v)[4]) {
    if (nv == 4) {
        const f4u q = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<f4u *>(p));
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < nv) p[c] = (float)v[c];
    }
}
__device__ __forceinline__ void store4(double *p, int nv, const double (&v)[4]) {
    if (nv == 4) {
        const d2u q0 = {v[0], v[1]}, q1 = {v[2], v[3]};
        __builtin_nontemporal_store(q0, reinterpret_cast<d2u *>(p));
        __builtin_nontemporal_store(q1, reinterpret_cast<d2u *>(p) + 1);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < nv) p[c] = v[c];
    }
}

// XCD-aware block -> work-item map: blocks b and b+8 share an XCD (round-robin dispatch), so XCD x
// takes the contiguous band [x*per, (x+1)*per) and sweeps it in order.  Speed only, never correctness.
__device__ __forceinline__ int64_t xcd_band_item(int64_t total) {
    const int64_t b = blockIdx.x, per = (total + 7) / 8;
    return (b & 7) * per + (b >> 3);
}

// ------------------------------------------------------------------------------------------ cdfvT
struct VtArgs {
    const float *u, *v, *t, *s, *e3u, *e3v;   // frame 0 of (npi,npj,npk) fields
    int64_t fstride;                          // elements between frames
    int nfr, npi, npj, npk, ntx, tw;          // tw: cells of a tile (a multiple of 4, <= kTile)
    void *vt, *vs, *ut, *us, *e3um, *e3vm;    // (npi,npj,npk) r8 or r4
};

template <bool VVL, typename OUT>
__global__ void __launch_bounds__(kThreads) vt_kernel(VtArgs a) {
    __shared__ float lt[2][kThreads + 1], ls[2][kThreads + 1];
    const int64_t nrows = (int64_t)a.npj * a.npk, total = nrows * a.ntx;
    const int64_t q = xcd_band_item(total);
    if (q >= total) return;                                      // whole block: no barrier is skipped
    const int64_t L = q / a.ntx;
    const int tx = (int)(q - L * a.ntx), tid = threadIdx.x;
    const int jj = (int)(L % a.npj);
    const int i0 = tx * a.tw + tid * 4;
    const int nv = min(min(4, a.tw - tid * 4), a.npi - i0);       // <= 0: lane past the tile or the row end
    const bool north = jj < a.npj - 1;
    const int64_t cell = L * a.npi + i0;
    const int ih = tx * a.tw + a.tw;                              // halo column of this tile
    const int nthr = a.tw / 4;                                    // lanes that hold cells (tw is a multiple of 4); lane nthr - 1 looks at the halo

    double aut[4] = {0, 0, 0, 0}, avt[4] = {0, 0, 0, 0}, aus[4] = {0, 0, 0, 0}, avs[4] = {0, 0, 0, 0};
    double ce3u[4] = {0, 0, 0, 0}, ce3v[4] = {0, 0, 0, 0};
    // Software pipeline over the frames (as in the cdfmoc kernel): the loads of frame f+1 are in
    // flight while frame f is interpolated and accumulated.
    struct Frame { float t[4], s[4], tn[4], sn[4], u[4], v[4], e3u[4], e3v[4], ht, hs; };
    auto fetch = [&](Frame &r, int f) {
        const int64_t e = (int64_t)f * a.fstride + cell;
#pragma unroll
        for (int c = 0; c < 4; ++c) { r.t[c] = r.s[c] = r.tn[c] = r.sn[c] = r.u[c] = r.v[c] = r.e3u[c] = r.e3v[c] = 0.f; }
        r.ht = r.hs = 0.f;
        if (nv > 0) {
            load4(a.t + e, nv, r.t); load4(a.s + e, nv, r.s); load4s(a.u + e, nv, r.u); load4s(a.v + e, nv, r.v);
            if (north) { load4(a.t + e + a.npi, nv, r.tn); load4(a.s + e + a.npi, nv, r.sn); }
            if (VVL) { load4s(a.e3u + e, nv, r.e3u); load4s(a.e3v + e, nv, r.e3v); }
        }
        if (tid == 0 && ih < a.npi) {                             // halo cell of the tile
            const int64_t eh = (int64_t)f * a.fstride + L * a.npi + ih;
            r.ht = a.t[eh]; r.hs = a.s[eh];
        }
    };
    Frame cur, nxt;
    fetch(cur, 0);
    for (int f = 0; f < a.nfr; ++f) {
        if (f + 1 < a.nfr) fetch(nxt, f + 1);
        const float (&t)[4] = cur.t, (&s)[4] = cur.s, (&tn)[4] = cur.tn, (&sn)[4] = cur.sn;
        const float (&u)[4] = cur.u, (&v)[4] = cur.v, (&e3u)[4] = cur.e3u, (&e3v)[4] = cur.e3v;
        const int buf = f & 1;                                    // double-buffered: one barrier per frame
        if (tid < nthr) { lt[buf][tid] = t[0]; ls[buf][tid] = s[0]; }          // (lanes past the tile hold nothing)
        if (tid == 0) { lt[buf][nthr] = cur.ht; ls[buf][nthr] = cur.hs; }
        __syncthreads();
        const float tnx = lt[buf][tid + 1], snx = ls[buf][tid + 1];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // zworku / zworkv are set for ji < npiglo, jj < npjglo only, 0 elsewhere (src/cdfvT.f90:201-206)
            const bool inner = north && (i0 + c < a.npi - 1);
            const float te = (c < 3) ? t[c < 3 ? c + 1 : 3] : tnx, se = (c < 3) ? s[c < 3 ? c + 1 : 3] : snx;
            const float wtu = inner ? __fmul_rn(0.5f, __fadd_rn(t[c], te)) : 0.f;
            const float wtv = inner ? __fmul_rn(0.5f, __fadd_rn(t[c], tn[c])) : 0.f;
            const float wsu = inner ? __fmul_rn(0.5f, __fadd_rn(s[c], se)) : 0.f;
            const float wsv = inner ? __fmul_rn(0.5f, __fadd_rn(s[c], sn[c])) : 0.f;
            if (VVL) {                                            // :208-211, :228-229
                aut[c] = __dadd_rn(aut[c], (double)__fmul_rn(__fmul_rn(wtu, e3u[c]), u[c]));
                avt[c] = __dadd_rn(avt[c], (double)__fmul_rn(__fmul_rn(wtv, e3v[c]), v[c]));
                aus[c] = __dadd_rn(aus[c], (double)__fmul_rn(__fmul_rn(wsu, e3u[c]), u[c]));
                avs[c] = __dadd_rn(avs[c], (double)__fmul_rn(__fmul_rn(wsv, e3v[c]), v[c]));
                ce3u[c] = __dadd_rn(ce3u[c], (double)e3u[c]);
                ce3v[c] = __dadd_rn(ce3v[c], (double)e3v[c]);
            } else {                                              // :214-215, :231-232
                aut[c] = __dadd_rn(aut[c], (double)__fmul_rn(wtu, u[c]));
                avt[c] = __dadd_rn(avt[c], (double)__fmul_rn(wtv, v[c]));
                aus[c] = __dadd_rn(aus[c], (double)__fmul_rn(wsu, u[c]));
                avs[c] = __dadd_rn(avs[c], (double)__fmul_rn(wsv, v[c]));
            }
        }
        cur = nxt;
    }
    if (nv <= 0) return;
    const double dn = (double)a.nfr;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (VVL) {                                                // :240-245
            avt[c] = __ddiv_rn(avt[c], ce3v[c]); avs[c] = __ddiv_rn(avs[c], ce3v[c]);
            aut[c] = __ddiv_rn(aut[c], ce3u[c]); aus[c] = __ddiv_rn(aus[c], ce3u[c]);
            ce3u[c] = __ddiv_rn(ce3u[c], dn); ce3v[c] = __ddiv_rn(ce3v[c], dn);
        } else {                                                  // :247-250
            avt[c] = __ddiv_rn(avt[c], dn); avs[c] = __ddiv_rn(avs[c], dn);
            aut[c] = __ddiv_rn(aut[c], dn); aus[c] = __ddiv_rn(aus[c], dn);
        }
    }
    store4((OUT *)a.vt + cell, nv, avt); store4((OUT *)a.vs + cell, nv, avs);
    store4((OUT *)a.ut + cell, nv, aut); store4((OUT *)a.us + cell, nv, aus);
    if (VVL) { store4((OUT *)a.e3um + cell, nv, ce3u); store4((OUT *)a.e3vm + cell, nv, ce3v); }
}

// ------------------------------------------------------------------------------------------ cdfcurl
struct CurlArgs {
    const float *un, *vn;                     // (npi,npj,nlev)
    const float *e1u, *e2v, *e1f, *e2f;       // (npi,npj)
    const float *den;                         // (npi,npj) e1f*e2f in r4 (the tile kernel reads one plane instead of two)
    const double *ff;                         // (npi,npj) or null
    int npi, npj, nlev, ntx, lperio;
    int band;                                 // curl_tile_kernel: rows of a band (0: the whole eighth of an XCD), see there
    int fast;                                 // curl_tile_kernel: the branch-free level loop for inner tiles (CDFT_CURL_FAST=0: A/B hook)
    void *out;                                // (npi,npj,nlev) r8 or r4
    float *work;                              // F-point curl for the T-point pass: the r4 value the reference widens into dl_rotn (:282), widened when read
    uint8_t *fmk;                             // ... and its fmask (0 / 1), so that the T-point pass does not form it again from u and v
};

// one F point from scalar loads (periodic column, T-point pass); un/vn point at the level
__device__ __forceinline__ float curl_at(const CurlArgs &a, const float *un, const float *vn, int i, int j,
                                         float *fmask) {
    const int64_t p = (int64_t)j * a.npi + i, pe = p + 1, pn = p + a.npi;
    float f = __fmul_rn(__fmul_rn(__fmul_rn(un[p], un[pn]), vn[p]), vn[pe]);      // src/cdfcurl.f90:270
    f = (f != 0.f) ? 1.f : 0.f;
    if (fmask) *fmask = f;
    const float num = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a.e2v[pe], vn[pe]), -__fmul_rn(a.e2v[p], vn[p])),
                                          -__fmul_rn(a.e1u[pn], un[pn])), __fmul_rn(a.e1u[p], un[p]));
    return __fdiv_rn(__fmul_rn(num, f), __fmul_rn(a.e1f[p], a.e2f[p]));           // :276-282
}

// One workgroup = one 1024-cell tile of row jj for ALL levels: the six metric vectors are loaded once
// into registers and every level streams past them (12 algorithmic bytes per cell-level).
template <typename OUT, bool KEEP>
__global__ void __launch_bounds__(kThreads) curl_kernel(CurlArgs a) {
    __shared__ float lv[2][kThreads + 1], le[kThreads + 1];
    const int64_t total = (int64_t)a.npj * a.ntx;
    const int64_t q = xcd_band_item(total);
    if (q >= total) return;                                       // whole block
    const int jj = (int)(q / a.ntx), tx = (int)(q - (int64_t)jj * a.ntx), tid = threadIdx.x;
    const int i0 = tx * kTile + tid * 4, nv = min(4, a.npi - i0);
    const int ih = tx * kTile + kTile;                            // halo column of the tile
    const bool north = jj < a.npj - 1, has_halo = ih < a.npi;
    const int64_t nij = (int64_t)a.npi * a.npj, p2 = (int64_t)jj * a.npi + i0;
    float e1u[4] = {0, 0, 0, 0}, e1un[4] = {0, 0, 0, 0}, e2v[4] = {0, 0, 0, 0}, e1f[4] = {1, 1, 1, 1}, e2f[4] = {1, 1, 1, 1};
    double ffv[4] = {0, 0, 0, 0};
    if (nv > 0) {
        load4(a.e1u + p2, nv, e1u); load4(a.e2v + p2, nv, e2v); load4(a.e1f + p2, nv, e1f); load4(a.e2f + p2, nv, e2f);
        if (north) load4(a.e1u + p2 + a.npi, nv, e1un);
        if (!KEEP && a.ff)
            for (int c = 0; c < nv; ++c) ffv[c] = a.ff[p2 + c];
    }
    le[tid] = e2v[0];
    if (tid == 0) le[kThreads] = has_halo ? a.e2v[(int64_t)jj * a.npi + ih] : 0.f;
    __syncthreads();
    const float enx = le[tid + 1];
    float den[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) den[c] = __fmul_rn(e1f[c], e2f[c]);
    // software pipeline over the levels: level lev+1 is in flight while level lev is differenced
    struct Level { float u[4], un1[4], v[4], hv; };
    auto fetch = [&](Level &r, int lev) {
        const float *un = a.un + (int64_t)lev * nij, *vn = a.vn + (int64_t)lev * nij;
#pragma unroll
        for (int c = 0; c < 4; ++c) { r.u[c] = r.un1[c] = r.v[c] = 0.f; }
        r.hv = 0.f;
        if (nv > 0) {
            load4(un + p2, nv, r.u); load4s(vn + p2, nv, r.v);
            if (north) load4(un + p2 + a.npi, nv, r.un1);
        }
        if (tid == 0 && has_halo) r.hv = vn[(int64_t)jj * a.npi + ih];
    };
    Level cur, nxt;
    fetch(cur, 0);
    for (int lev = 0; lev < a.nlev; ++lev) {
        const float *un = a.un + (int64_t)lev * nij, *vn = a.vn + (int64_t)lev * nij;
        if (lev + 1 < a.nlev) fetch(nxt, lev + 1);
        float u[4], un1[4], v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { u[c] = cur.u[c]; un1[c] = cur.un1[c]; v[c] = cur.v[c]; }
        const float hv = cur.hv;
        cur = nxt;
        const int buf = lev & 1;                                  // double-buffered: one barrier per level
        lv[buf][tid] = v[0];
        if (tid == 0) lv[buf][kThreads] = hv;
        __syncthreads();
        if (nv <= 0) continue;
        const float vnx = lv[buf][tid + 1];
        double r[4];
        uint8_t fk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + c;
            float val = 0.f;                                          // drotn(:,:) = 0.d0, :275
            if (c < nv && north) {
                if (i < a.npi - 1) {
                    const float ve = (c < 3) ? v[c < 3 ? c + 1 : 3] : vnx, ee = (c < 3) ? e2v[c < 3 ? c + 1 : 3] : enx;
                    float f = __fmul_rn(__fmul_rn(__fmul_rn(u[c], un1[c]), v[c]), ve);
                    f = (f != 0.f) ? 1.f : 0.f;
                    fk[c] = (uint8_t)f;
                    const float num = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(ee, ve), -__fmul_rn(e2v[c], v[c])),
                                                          -__fmul_rn(e1un[c], un1[c])), __fmul_rn(e1u[c], u[c]));
                    val = __fdiv_rn(__fmul_rn(num, f), den[c]);
                } else if (a.lperio && a.npi > 2) {
                    val = curl_at(a, un, vn, 1, jj, nullptr);         // drotn(npiglo,:) = drotn(2,:), :285
                }
            }
            r[c] = (double)val;
            if (KEEP && c < nv) { a.work[(int64_t)lev * nij + p2 + c] = val; a.fmk[(int64_t)lev * nij + p2 + c] = fk[c]; }
            if (!KEEP && a.ff && ffv[c] != 0.0) r[c] = __ddiv_rn(r[c], ffv[c]);   // -overf, :299-301
        }
        if (!KEEP) store4((OUT *)a.out + (int64_t)lev * nij + p2, nv, r);
    }
}

// The kernel that runs (npiglo >= 4).  A tile is `tw` columns of one row, 4 per lane, for a range of levels; tw and
// the workgroup size are chosen so that nearly every lane has cells (1024-column tiles left 30 % of the lanes idle
// at npiglo = 1442).  Waves share nothing: the east neighbour of a lane's last cell is the next lane's first one
// (a ds_bpermute; the last lane of a wave or tile loads it), the last vector of the row is anchored at npiglo-4 (its
// repeated cells compute and store what their owner does).  Workgroups are handed out in (level chunk, row, tile)
// order, so the ones running together sweep neighbouring rows through the same levels: that keeps the u(j+1) row in
// L2 for the row above and the DRAM accesses in long runs (a persistent grid with equal shares of (tile, level)
// units, each workgroup somewhere else in the levels, ran 50 % slower).  The levels are cut into `lc` chunks only
// when one chunk would leave the chip half empty in its last round (eORCA025: 1.2 rounds).
#ifndef CDFT_CURL_PF
#define CDFT_CURL_PF 2
#endif
constexpr int kCurlPF = CDFT_CURL_PF;                             // level buffers in registers (kCurlPF-1 in flight)
#ifndef CDFT_CURL_FPF
#define CDFT_CURL_FPF 2
#endif
constexpr int kCurlFPF = CDFT_CURL_FPF;                           // the same for the inner tiles' loop
#ifndef CDFT_CURL_WPE
#define CDFT_CURL_WPE 6
#endif

template <typename OUT, bool KEEP>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(CDFT_CURL_WPE, 8))) curl_tile_kernel(CurlArgs a, int tw, int lc) {
    const int lane = threadIdx.x & 63;
    const int64_t tiles = (int64_t)a.ntx * a.npj, total = tiles * lc;
    int chunk, jj, tx;
    if (a.band > 0) {
        // Row bands (A/B hook CDFT_CURL_BAND): XCD x owns the x-th eighth of the rows and works through it band by band --
        // a band = a.band rows, all their level chunks (chunk-major), then the next band -- so that the metric planes of a
        // band (3 x 17 KB per row) are asked for again while they are still in L2 / Infinity Cache, with short level chunks.
        const int x = blockIdx.x & 7;
        const int64_t s = blockIdx.x >> 3;
        const int j0 = (int)((int64_t)x * a.npj / 8), nrows = (int)((int64_t)(x + 1) * a.npj / 8) - j0;
        const int64_t per_band = (int64_t)a.band * a.ntx * lc;
        const int b = (int)(s / per_band);
        const int r0 = b * a.band, rb = min(a.band, nrows - r0);                 // rows of this band
        if (rb <= 0) return;
        const int64_t l = s - (int64_t)b * per_band;
        if (l >= (int64_t)rb * a.ntx * lc) return;
        chunk = (int)(l / ((int64_t)rb * a.ntx));
        const int64_t t = l - (int64_t)chunk * rb * a.ntx;
        jj = j0 + r0 + (int)(t / a.ntx);
        tx = (int)(t - (int64_t)(t / a.ntx) * a.ntx);
    } else {
        const int64_t job = xcd_band_item(total);
        if (job >= total) return;                                 // whole block
        chunk = (int)(job / tiles);
        const int64_t tile = job - chunk * tiles;
        jj = (int)(tile / a.ntx); tx = (int)(tile - (int64_t)jj * a.ntx);
    }
    const int64_t nij = (int64_t)a.npi * a.npj;
    {
        const int lev0 = (int)((int64_t)chunk * a.nlev / lc), lev1 = (int)((int64_t)(chunk + 1) * a.nlev / lc);
#define CURL_TILE_FAST 1
#include "curl_tile_body.inc"
#undef CURL_TILE_FAST
    }
}

// The persistent order (CDFT_CURL_PERSIST; a kernel of its own so that the orders above keep their registers): the grid is what the
// chip holds at once, XCD x owns the x-th eighth of the rows, and workgroup w of an XCD's W takes the tiles w, w + W, w + 2W, ...
// of that eighth in (row, tile) order, each for ALL levels: the metrics are read once per tile, every workgroup of the chip starts
// its tile at level 0 at the same moment and they walk the levels side by side -- a level of W consecutive tiles is one
// contiguous run per XCD and field.
template <typename OUT, bool KEEP>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(6, 8))) curl_tile_persist_kernel(CurlArgs a, int tw) {
    const int lane = threadIdx.x & 63;
    const int64_t nij = (int64_t)a.npi * a.npj;
    const int W = (int)(gridDim.x >> 3), px = blockIdx.x & 7;
    const int pj0 = (int)((int64_t)px * a.npj / 8), ntiles = ((int)((int64_t)(px + 1) * a.npj / 8) - pj0) * a.ntx;
    const int lev0 = 0, lev1 = a.nlev;
    for (int pt = (int)(blockIdx.x >> 3); pt < ntiles; pt += W) {
        const int jj = pj0 + pt / a.ntx, tx = pt % a.ntx;
#include "curl_tile_body.inc"
    }
}

// Level chunks: the fewest that fill the last round of workgroups (slots = what the chip holds at once); a chunk
// reloads the tile's metrics, 20 bytes per cell against 12 per cell and level.
inline int curl_level_chunks(int64_t tiles, int nlev, int slots) {
    int best = 1;
    double best_cost = 1e30;
    for (int lc = 1; lc <= std::min(nlev, 8); ++lc) {
        const double rounds = (double)tiles * lc / slots;
        const double cost = std::ceil(rounds) / rounds * (1.0 + 20.0 * lc / (12.0 * nlev));
        if (cost < best_cost - 0.02) { best_cost = cost; best = lc; }
    }
    return best;
}

// Tiles per row, tile width (a multiple of 4) and workgroup size (a multiple of 64, <= kThreads) that leave the
// fewest lanes without cells; fewer, wider tiles on a tie.
struct CurlShape { int ntx, tw, threads; };
inline CurlShape curl_shape(int npi) {
    CurlShape best{1, 4, 64};
    double best_eff = -1.0;
    const int n0 = (npi + kTile - 1) / kTile;
    for (int ntx = n0; ntx <= n0 + 4; ++ntx) {
        const int tw = (((npi + ntx - 1) / ntx + 3) / 4) * 4, threads = ((tw / 4 + 63) / 64) * 64;
        if (threads > kThreads) continue;
        const double eff = (double)npi / ((double)ntx * threads * 4);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = CurlShape{ntx, tw, threads}; }
    }
    return best;
}

// Workgroups of `kernel` the device holds at once (the persistent grid), a multiple of 8 for the XCD band map.
template <typename K>
int resident_workgroups(K kernel, int threads) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return 2048;
    }
    return ((prop.multiProcessorCount * per_cu + 7) / 8) * 8;
}

// -T: 4-point mean of the F-point curl to T points with the product of the 4 fmask (:286-297).
// fmask(npiglo,:) and fmask(:,npjglo) are never assigned by the reference; taken as 0 (the F-point pass writes 0 there).
// A lane takes four cells of a row: the F-point values and masks of its row and the row to the south, columns i0-1 .. i0+3,
// are 2 x (an r4 value, a 16-byte vector; a byte, a 4-byte word) -- one thread per cell with its four masks formed again from
// u and v and the F-point values kept in r8 was 21 loads per cell and 16 bytes more: 1.0 ms at eORCA025.
template <typename OUT>
__global__ void __launch_bounds__(256) curl_tpoint_kernel(CurlArgs a) {
    const int64_t nij = (int64_t)a.npi * a.npj;
    const int i0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x), j = blockIdx.y, lev = blockIdx.z;
    if (i0 >= a.npi) return;
    const float *w = a.work + (int64_t)lev * nij;
    const uint8_t *fk = a.fmk + (int64_t)lev * nij;
    OUT *out = (OUT *)a.out + (int64_t)lev * nij + (int64_t)j * a.npi;
    auto tval = [&](int ii, int jj) -> double {                  // dl_rotn(ii,jj), ii >= 1, jj >= 1: one cell from scalar loads
        const int64_t q = (int64_t)jj * a.npi + ii;
        const float zm = __fmul_rn(__fmul_rn(__fmul_rn((float)fk[q], (float)fk[q - a.npi]), (float)fk[q - 1]), (float)fk[q - a.npi - 1]);
        const double s4 = __dadd_rn(__dadd_rn(__dadd_rn((double)w[q], (double)w[q - a.npi]), (double)w[q - 1]), (double)w[q - a.npi - 1]);
        return __dmul_rn(__dmul_rn(0.25, s4), (double)zm);
    };
    double r[4] = {0.0, 0.0, 0.0, 0.0};
    const int nv = min(4, a.npi - i0);
    if (j >= 1) {
        if (nv == 4 && i0 >= 1) {                                 // the usual lane: columns i0-1 .. i0+3 of rows j and j-1
            double wn[5], ws[5];
            float fn[5], fs[5];
            const int64_t q = (int64_t)j * a.npi + i0;
            typedef uint32_t u32u __attribute__((aligned(1)));
            wn[0] = (double)w[q - 1]; ws[0] = (double)w[q - a.npi - 1];
            const f4u n4 = *reinterpret_cast<const f4u *>(w + q), s4v = *reinterpret_cast<const f4u *>(w + q - a.npi);
            wn[1] = (double)n4.x; wn[2] = (double)n4.y; wn[3] = (double)n4.z; wn[4] = (double)n4.w;
            ws[1] = (double)s4v.x; ws[2] = (double)s4v.y; ws[3] = (double)s4v.z; ws[4] = (double)s4v.w;
            const uint32_t bn = *reinterpret_cast<const u32u *>(fk + q), bs = *reinterpret_cast<const u32u *>(fk + q - a.npi);
            fn[0] = (float)fk[q - 1]; fs[0] = (float)fk[q - a.npi - 1];
#pragma unroll
            for (int c = 0; c < 4; ++c) { fn[c + 1] = (float)((bn >> (8 * c)) & 0xffu); fs[c + 1] = (float)((bs >> (8 * c)) & 0xffu); }
#pragma unroll
            for (int c = 0; c < 4; ++c) {                         // the reference's order: (ii,jj), (ii,jj-1), (ii-1,jj), (ii-1,jj-1)
                const float zm = __fmul_rn(__fmul_rn(__fmul_rn(fn[c + 1], fs[c + 1]), fn[c]), fs[c]);
                const double s4 = __dadd_rn(__dadd_rn(__dadd_rn(wn[c + 1], ws[c + 1]), wn[c]), ws[c]);
                r[c] = __dmul_rn(__dmul_rn(0.25, s4), (double)zm);
            }
        } else {                                                  // the first lane of a row, the last one of a row that is not a multiple of 4
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = i0 + c;
                if (c >= nv) continue;
                if (i >= 1) r[c] = tval(i, j);
                else if (a.lperio && a.npi > 1) r[c] = tval(a.npi - 1, j);    // dl_rotn(1,:) = dl_rotn(npiglo,:)
            }
        }
    }
    if (a.ff) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < nv) { const double f = a.ff[(int64_t)j * a.npi + i0 + c]; if (f != 0.0) r[c] = __ddiv_rn(r[c], f); }
    }
    store4(out + i0, nv, r);
}

// e1f*e2f in r4, the divisor of src/cdfcurl.f90:276-282, once per mesh
__global__ void curl_den_kernel(const float *__restrict__ e1f, const float *__restrict__ e2f, float *__restrict__ den, int64_t n) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) den[p] = __fmul_rn(e1f[p], e2f[p]);
}

DevBuf g_in[6], g_out[6];

}  // namespace

namespace cdft { namespace shard {
int vt_compute(int ndev, const int *devices, const float *u, const float *v, const float *t, const float *s, const float *e3u,
               const float *e3v, int64_t fstride, int nfr, int npi, int npj, int npk, void *vt, void *vs, void *ut, void *us,
               void *e3um, void *e3vm, int out_r4);
} }

extern "C" {

int cdft_vt_compute_sharded(const float *u, const float *v, const float *t, const float *s, const float *e3u,
                            const float *e3v, int64_t fstride, int nframes, int npiglo, int npjglo, int npk,
                            void *vt, void *vs, void *ut, void *us, void *e3um, void *e3vm, int out_is_r4, int ndev,
                            const int *devices) {
    TraceRange trace_("cdft_vt_compute_sharded");
    if (!u || !v || !t || !s || !vt || !vs || !ut || !us)
        return fail(CDFT_ERR_ARG, "cdft_vt_compute_sharded: NULL argument");
    if (e3u && (!e3v || !e3um || !e3vm)) return fail(CDFT_ERR_ARG, "cdft_vt_compute_sharded: -vvl needs e3u, e3v, e3um, e3vm");
    if (npiglo < 1 || npjglo < 1 || npk < 1 || nframes < 1 || fstride < (int64_t)npiglo * npjglo * npk)
        return fail(CDFT_ERR_ARG, "cdft_vt_compute_sharded: bad sizes (%d,%d,%d) x %d frames", npiglo, npjglo, npk, nframes);
    return cdft::shard::vt_compute(ndev, devices, u, v, t, s, e3u, e3v, fstride, nframes, npiglo, npjglo, npk, vt, vs, ut, us,
                                   e3um, e3vm, out_is_r4);
}

int cdft_vt_compute(const float *u, const float *v, const float *t, const float *s, const float *e3u,
                    const float *e3v, int64_t fstride, int nframes, int npiglo, int npjglo, int npk, int in_space,
                    void *vt, void *vs, void *ut, void *us, void *e3um, void *e3vm, int out_is_r4, int out_space,
                    void *stream) {
    TraceRange trace_("cdft_vt_compute");
    if (!u || !v || !t || !s || !vt || !vs || !ut || !us)
        return fail(CDFT_ERR_ARG, "cdft_vt_compute: NULL argument");
    const bool vvl = e3u != nullptr;
    if (vvl && (!e3v || !e3um || !e3vm)) return fail(CDFT_ERR_ARG, "cdft_vt_compute: -vvl needs e3u, e3v, e3um, e3vm");
    const int64_t n3 = (int64_t)npiglo * npjglo * npk;
    if (npiglo < 1 || npjglo < 1 || npk < 1 || nframes < 1 || fstride < n3)
        return fail(CDFT_ERR_ARG, "cdft_vt_compute: bad sizes (%d,%d,%d) x %d frames, stride %lld", npiglo, npjglo, npk,
                    nframes, (long long)fstride);
    CDFT_TRY(require_device());
    hipStream_t st = as_stream(stream);
    const size_t inb = ((size_t)(nframes - 1) * fstride + n3) * 4, esz = out_is_r4 ? 4 : 8;
    const float *src[6] = {u, v, t, s, e3u, e3v};
    const void *din[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int q = 0; q < (vvl ? 6 : 4); ++q) CDFT_TRY(to_device(src[q], inb, in_space, g_in[q], st, &din[q]));
    void *hout[6] = {vt, vs, ut, us, e3um, e3vm}, *dout[6];
    for (int q = 0; q < 6; ++q) {
        dout[q] = hout[q];
        if ((out_space == CDFT_MEM_HOST || out_space == CDFT_MEM_PINNED) && q < (vvl ? 6 : 4)) {
            CDFT_TRY(g_out[q].reserve(n3 * esz));
            dout[q] = g_out[q].ptr;
        }
    }
    VtArgs a;
    a.u = (const float *)din[0]; a.v = (const float *)din[1]; a.t = (const float *)din[2]; a.s = (const float *)din[3];
    a.e3u = (const float *)din[4]; a.e3v = (const float *)din[5]; a.fstride = fstride; a.nfr = nframes;
    a.npi = npiglo; a.npj = npjglo; a.npk = npk;
    // tiles of equal width chosen so that few lanes hang over the row's end (cdfcurl's rule: a row of 4322 cells in six tiles
    // of 724 on 192 lanes, 94 % of the lanes at work, instead of four of 1024 and one of 226, 84 %); CDFT_VT_TW: A/B hook
    CurlShape sh = curl_shape(npiglo);
    {   // ... where that buys something: at 4322 cells (84 -> 94 % of the lanes) the two were measured equal, at 1442 (70 -> 94 %) 7 % apart
        const int n0 = (npiglo + kTile - 1) / kTile;
        const double eff0 = (double)npiglo / ((double)n0 * kTile), eff1 = (double)npiglo / ((double)sh.ntx * sh.threads * 4);
        if (eff1 - eff0 < 0.1) sh = CurlShape{n0, kTile, kThreads};
    }
    if (const char *etw = getenv("CDFT_VT_TW")) {
        const int tw = std::min(kTile, atoi(etw) / 4 * 4);
        if (tw >= 4) sh = CurlShape{(npiglo + tw - 1) / tw, tw, ((tw / 4 + 63) / 64) * 64};      // (0: the default)
    }
    a.ntx = sh.ntx; a.tw = sh.tw;
    a.vt = dout[0]; a.vs = dout[1]; a.ut = dout[2]; a.us = dout[3]; a.e3um = dout[4]; a.e3vm = dout[5];
    const int64_t total = (int64_t)npjglo * npk * a.ntx;
    const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(sh.threads);
    if (vvl) {
        if (out_is_r4) hipLaunchKernelGGL((vt_kernel<true, float>), grid, block, 0, st, a);
        else           hipLaunchKernelGGL((vt_kernel<true, double>), grid, block, 0, st, a);
    } else {
        if (out_is_r4) hipLaunchKernelGGL((vt_kernel<false, float>), grid, block, 0, st, a);
        else           hipLaunchKernelGGL((vt_kernel<false, double>), grid, block, 0, st, a);
    }
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST || out_space == CDFT_MEM_PINNED) {
        for (int q = 0; q < (vvl ? 6 : 4); ++q)
            CDFT_HIP(hipMemcpyAsync(hout[q], dout[q], n3 * esz, hipMemcpyDeviceToHost, st));
        if (out_space == CDFT_MEM_HOST) CDFT_HIP(hipStreamSynchronize(st));      // (page-locked: on their way when the call returns)
    }
    return CDFT_OK;
}

int cdft_stencil_release(void) {
    for (DevBuf &b : g_in) b.release();
    for (DevBuf &b : g_out) b.release();
    return CDFT_OK;
}

}  // extern "C"

struct cdft_curl_plan {
    int npi = 0, npj = 0, lperio = 0;
    float *e1u = nullptr, *e2v = nullptr, *e1f = nullptr, *e2f = nullptr, *den = nullptr;
    double *ff = nullptr;
    bool mesh_set = false;
    DevBuf su, sv, sout, work;
};

extern "C" {

int cdft_curl_create(cdft_curl_plan **plan, int npiglo, int npjglo) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_curl_create: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1) return fail(CDFT_ERR_ARG, "cdft_curl_create: bad sizes %d x %d", npiglo, npjglo);
    CDFT_TRY(require_device());
    cdft_curl_plan *p = new cdft_curl_plan;
    p->npi = npiglo; p->npj = npjglo;
    *plan = p;
    return CDFT_OK;
}

int cdft_curl_set_mesh(cdft_curl_plan *p, const float *e1u, const float *e2v, const float *e1f, const float *e2f,
                       const double *ff, int lperio, int memspace, void *stream) {
    if (!p || !e1u || !e2v || !e1f || !e2f) return fail(CDFT_ERR_ARG, "cdft_curl_set_mesh: NULL argument");
    hipStream_t st = as_stream(stream);
    const size_t nij = (size_t)p->npi * p->npj;
    const hipMemcpyKind kind = memspace == CDFT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    float **dst[4] = {&p->e1u, &p->e2v, &p->e1f, &p->e2f};
    const float *src[4] = {e1u, e2v, e1f, e2f};
    for (int q = 0; q < 4; ++q) {
        if (!*dst[q] && hipMalloc((void **)dst[q], nij * 4 + 64) != hipSuccess) {
            (void)hipGetLastError();
            return fail(CDFT_ERR_ALLOC, "cdft_curl_set_mesh: device allocation failed");
        }
        CDFT_HIP(hipMemcpyAsync(*dst[q], src[q], nij * 4, kind, st));
    }
    if (ff) {
        if (!p->ff && hipMalloc((void **)&p->ff, nij * 8 + 64) != hipSuccess) {
            (void)hipGetLastError();
            return fail(CDFT_ERR_ALLOC, "cdft_curl_set_mesh: device allocation failed");
        }
        CDFT_HIP(hipMemcpyAsync(p->ff, ff, nij * 8, kind, st));
    } else if (p->ff) {
        (void)hipFree(p->ff);
        p->ff = nullptr;
    }
    if (!p->den && hipMalloc((void **)&p->den, nij * 4 + 64) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ALLOC, "cdft_curl_set_mesh: device allocation failed");
    }
    hipLaunchKernelGGL(curl_den_kernel, dim3((unsigned)((nij + 255) / 256)), dim3(256), 0, st, p->e1f, p->e2f, p->den, (int64_t)nij);
    CDFT_LAUNCH_CHECK();
    CDFT_HIP(hipStreamSynchronize(st));
    p->lperio = lperio ? 1 : 0;
    p->mesh_set = true;
    return CDFT_OK;
}

int cdft_curl_compute(cdft_curl_plan *p, const float *un, const float *vn, int nlev, int in_space, int ltpoint,
                      void *out, int out_is_r8, int out_space, void *stream) {
    TraceRange trace_("cdft_curl_compute");
    if (!p || !un || !vn || !out) return fail(CDFT_ERR_ARG, "cdft_curl_compute: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_curl_compute: cdft_curl_set_mesh has not succeeded");
    if (nlev < 1) return fail(CDFT_ERR_ARG, "cdft_curl_compute: nlev = %d", nlev);
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj, n3 = nij * nlev;
    const size_t esz = out_is_r8 ? 8 : 4;
    const void *du, *dv;
    CDFT_TRY(to_device(un, n3 * 4, in_space, p->su, st, &du));
    CDFT_TRY(to_device(vn, n3 * 4, in_space, p->sv, st, &dv));
    void *dout = out;
    if (out_space == CDFT_MEM_HOST || out_space == CDFT_MEM_PINNED) {
        CDFT_TRY(p->sout.reserve(n3 * esz));
        dout = p->sout.ptr;
    }
    CurlArgs a;
    a.un = (const float *)du; a.vn = (const float *)dv; a.e1u = p->e1u; a.e2v = p->e2v; a.e1f = p->e1f; a.e2f = p->e2f; a.den = p->den;
    a.ff = p->ff; a.npi = p->npi; a.npj = p->npj; a.nlev = nlev;
    a.lperio = p->lperio; a.out = dout; a.work = nullptr; a.fmk = nullptr; a.band = 0; a.fast = 1;
    if (ltpoint) {
        CDFT_TRY(p->work.reserve(n3 * 5));
        a.work = (float *)p->work.ptr;
        a.fmk = (uint8_t *)p->work.ptr + n3 * 4;
    }
    const char *env = getenv("CDFT_CURL_ROWTILE");                // the one-row-tile kernel (tests compare the two)
    if (p->npi >= 4 && !(env && env[0] == '1')) {
        CurlShape sh = curl_shape(p->npi);
        if (const char *etw = getenv("CDFT_CURL_TW")) {           // A/B hook: tile width, a multiple of 4 in 4..1024
            const int tw = std::max(4, std::min(kTile, atoi(etw) / 4 * 4));
            sh = CurlShape{(p->npi + tw - 1) / tw, tw, ((tw / 4 + 63) / 64) * 64};
        }
        a.ntx = sh.ntx;
        const int64_t tiles = (int64_t)a.ntx * p->npj;
        const int variant = ltpoint ? 0 : out_is_r8 ? 1 : 2;
        typedef void (*CurlKernel)(CurlArgs, int, int);
        static const CurlKernel table[3] = {curl_tile_kernel<double, true>, curl_tile_kernel<double, false>, curl_tile_kernel<float, false>};
        static int resident[3][kThreads / 64];                    // per kernel variant and workgroup size; 0 = not asked yet
        int &res = resident[variant][sh.threads / 64 - 1];
        if (res == 0) res = resident_workgroups(table[variant], sh.threads);
        // Row bands (round 4): each XCD works through its eighth of the rows in bands of 128, a band's level chunks one after
        // the other, a chunk sized so that a (band, chunk) moves about 33 MB of u, v and result -- 5 levels at eORCA12, 15 at
        // eORCA025 (measured over bands of 16..256 rows x 1..25 chunks at both grids, profiles/r04/curl_band_*.txt: 3.05 -> 2.82 ms
        // at eORCA12, 0.338 -> 0.334 at eORCA025).  Grids with fewer than 128 rows per XCD keep the plain (chunk, row, tile) order.
        // CDFT_CURL_BAND (0: off) and CDFT_CURL_CHUNKS are A/B hooks.
        const int rows_x = (p->npj + 7) / 8;
        a.band = rows_x >= 128 ? 128 : 0;
        // the branch-free level loop of the inner tiles: 2.72 against 2.82-2.91 ms at eORCA12 (five tiles of six), nothing at
        // eORCA025 where one tile of two is inner (0.332 against 0.327 ms): on from three tiles per row (CDFT_CURL_FAST=0 / 1: A/B hook)
        { const char *ef = getenv("CDFT_CURL_FAST"); a.fast = ef ? (ef[0] == '0' ? 0 : 1) : (sh.ntx >= 3 ? 1 : 0); }
        if (const char *eb = getenv("CDFT_CURL_BAND")) a.band = std::max(0, atoi(eb));
        int lc = curl_level_chunks(tiles, nlev, res);
        if (a.band > 0) {
            const double per_level = (double)a.band * p->npi * 12.0;
            const int levs = std::max(1, (int)(33.0e6 / per_level + 0.5));
            lc = std::max(1, std::min(nlev, (nlev + levs - 1) / levs));
        }
        if (const char *elc = getenv("CDFT_CURL_CHUNKS")) lc = std::max(1, std::min(nlev, atoi(elc)));
        dim3 grid((unsigned)(((tiles * lc + 7) / 8) * 8)), block(sh.threads);
        if (const char *ep = getenv("CDFT_CURL_PERSIST")) {       // A/B hook: the persistent order, n workgroups per CU (0: what fits)
            const int percu = atoi(ep);
            if (percu >= 0 && ep[0] != '-') {
                int ncu = 256;
                hipDeviceProp_t prop;
                int dev = 0;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
                a.band = -1;
                lc = 1;
                grid = dim3((unsigned)(percu > 0 ? ((ncu * percu + 7) / 8) * 8 : res));
            }
        }
        if (a.band > 0) {
            const int nb = (rows_x + a.band - 1) / a.band;
            grid = dim3((unsigned)(8 * (int64_t)nb * a.band * a.ntx * lc));
        }
        if (a.band < 0) {
            typedef void (*CurlPersist)(CurlArgs, int);
            static const CurlPersist ptable[3] = {curl_tile_persist_kernel<double, true>, curl_tile_persist_kernel<double, false>,
                                                  curl_tile_persist_kernel<float, false>};
            hipLaunchKernelGGL(ptable[variant], grid, block, 0, st, a, sh.tw);
        } else
        hipLaunchKernelGGL(table[variant], grid, block, 0, st, a, sh.tw, lc);
    } else {
        a.ntx = (p->npi + kTile - 1) / kTile;
        const int64_t total = (int64_t)p->npj * a.ntx;
        const dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(kThreads);
        if (ltpoint)        hipLaunchKernelGGL((curl_kernel<double, true>), grid, block, 0, st, a);
        else if (out_is_r8) hipLaunchKernelGGL((curl_kernel<double, false>), grid, block, 0, st, a);
        else                hipLaunchKernelGGL((curl_kernel<float, false>), grid, block, 0, st, a);
    }
    if (ltpoint) {
        CDFT_LAUNCH_CHECK();
        const dim3 g2((unsigned)((p->npi + 1023) / 1024), (unsigned)p->npj, (unsigned)nlev);
        if (out_is_r8) hipLaunchKernelGGL((curl_tpoint_kernel<double>), g2, dim3(256), 0, st, a);
        else           hipLaunchKernelGGL((curl_tpoint_kernel<float>), g2, dim3(256), 0, st, a);
    }
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(out, dout, n3 * esz, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    } else if (out_space == CDFT_MEM_PINNED) {                    // on its way when the call returns; the next call's kernel
        CDFT_HIP(hipMemcpyAsync(out, dout, n3 * esz, hipMemcpyDeviceToHost, st));   // follows it in stream order
    }
    return CDFT_OK;
}

int cdft_curl_destroy(cdft_curl_plan *p) {
    if (!p) return CDFT_OK;
    void *ptrs[] = {p->e1u, p->e2v, p->e1f, p->e2f, p->den, p->ff};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    p->su.release(); p->sv.release(); p->sout.release(); p->work.release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"
