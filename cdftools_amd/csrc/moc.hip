// moc.hip -- cdfmoc plan: mesh preparation, zonal-kernel dispatch, bottom-to-surface cumsum.
// C ABI: include/cdftools_hip.h ("cdfmoc").  Reference loops: src/cdfmoc.f90:325-336 (masks),
// :343-348,:576-596 (e3v), :355-379 (zonal sum), :382-388 (vertical cumulated sum).
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "moc_zonal.hpp"
#include "shard.hpp"

using namespace cdft;
using namespace cdft::moc;

namespace cdft { namespace shard {
int moc_parts(ShardSet &sh);
int moc_set_mesh(ShardSet &sh, const float *e1v, const float *e3v, const int8_t *vmask, const int16_t *ibmask);
int moc_compute(ShardSet &sh, const float *zv, int64_t rec_stride, int npt, double *dmoc);
int moc_stage(ShardSet &sh, const float *zv, int64_t rec_stride, int npt);
int moc_run(ShardSet &sh, int npt, double *dmoc);
} }

struct cdft_moc_plan {
    cdft::ShardSet *sh = nullptr;   // set: a shell over one plan per row block (shard.hpp); the fields below are unused
    int npi = 0, npj = 0, npk = 0, nbasins = 0;
    float   *w    = nullptr;   // (npi,npj,npk) r4 weight e1v*e3v
    uint8_t *bits = nullptr;   // (npi,npj) basin bits
    uint8_t *cls  = nullptr;   // (npi,npj) class of the cell's bit pattern (class mode)
    int     *flag = nullptr;   // device: [0] error flag (mask value check), [1..256] pattern present
    uint8_t pat[kMaxClass] = {0};
    int     nclass = 0;        // distinct patterns (the empty one included), 0 when there are more than kMaxClass
    bool mesh_set = false;
    int  tune_waves = 0, tune_nv = 0, tune_mode = 0;
    DevBuf st_e1v, st_e3v, st_mask, st_zv, st_out;
    DevBuf seg_out;            // class kernel, rows cut into segments: the segments' sums (nseg x the size of the result)
    DevBuf st_vm;
    bool last_aligned = true;  // alignment of the records of the last launch (cdft_moc_describe)
    bool last_pool = false;    // the last launch read the pool (staged records)   // staging for host arrays
    // -decomp (cdft_moc_decomp_*): the mesh it needs beyond w / bits, and its work arrays
    bool dec_set = false;
    DevBuf d_e3v, d_e1u, d_zcoef, d_tmask, d_umask;  // e3v (masked) 3-D, e1u / zcoef 2-D, masks 3-D bytes
    std::vector<float> gdept;
    DevBuf d_t, d_s, d_zvgeo, d_sig8, d_dvbt, d_hdep, d_dvgeo, d_raw, d_o4;   // record inputs, work, results
    int64_t pref3d_n = 0;
    DevBuf d_pref3d;               // -decomp: gdept(jk) at every point of the levels 1 .. npk-1 (the reference depths of one sigmai launch)
    bool profiling = false;
    std::vector<hipEvent_t> ev;                      // (start, stop) pairs, one per profiled launch
    size_t ev_used = 0;
    // records handed over once by cdft_moc_stage: re-laid out in `pool` (below); with CDFT_MOC_POOL=0 (A/B hook) kept as they
    // came (in HBM: the caller's own array, or st_zv for a host array)
    const float *staged = nullptr;
    int64_t staged_stride = 0;
    int staged_npt = 0;
    // The pool: the plan's own layout of staged records.  Row (jj,jk) of the domain owns ONE contiguous run of (1+npt) slots
    // of pool_P r4 (pool_P = npi rounded up to 4): its weights, then its npt records -- what the workgroup of that row reads,
    // front to back, instead of 1+npt runs 4.7 GB apart (eORCA12).  Measured (tools/layout_probe.py, profiles/r04): 8.7-9.0 ms
    // per launch wherever the allocator put the pool, against 9.1-9.7 ms for the dense arrays depending on where they lay.
    DevBuf pool;
    int64_t pool_P = 0;        // slot length (elements)
    int pool_npt = 0;          // records the pool is laid out for (0: nothing staged in it)
    bool pool_w = false;       // the weights of the current mesh are in the pool
    // The stream route (cdft_moc_stream_*): the domain goes through HBM a WINDOW of levels at a time -- the window's rows laid out
    // like the pool's (a row's metric / weights, then its npt records, in one run) -- and nothing 3-D is resident: `win` is the
    // window, `win_vm` the window's vmask bytes (dense), `e1v2d` the 2-D e1v; the raw sums of every window land in st_out.
    DevBuf win, win_vm, e1v2d;
    int win_npt = 0, win_nlev = 0;
    int64_t win_P = 0;
    bool basins_set = false;   // bits / cls / e1v2d are there (cdft_moc_set_basins, or cdft_moc_set_mesh*)
    int64_t last_slot = 0;     // slot length of the layout the last launch read (cdft_moc_describe)
    int open_npt = 0;          // cdft_moc_stage_open ... _close: records being written into the pool by the caller's own copies
    int64_t n_launch = 0, n_launch_rec = 0;   // zonal launches so far and the records they covered (cdft_moc_stats)
};

static int moc_compute_on(cdft_moc_plan *p, const float *d_zv, int64_t rec_stride, int npt, double *dmoc, int dmoc_space,
                          hipStream_t st, int64_t z_row = 0, const float *w = nullptr, int64_t w_row = 0);
static int basins_from(cdft_moc_plan *p, const int16_t *d_msk, hipStream_t st);

namespace {

// w(i,j,k) = e1v(i,j) * e3v(i,j,k), rounded to r4 like the reference's first product
// (src/cdfmoc.f90:374: e1v(ji,jj)*e3v(ji,jj,jk) is evaluated first, in r4).
// With vm (the level's vmask as the reference's INTEGER*2 ivmask, here a byte): e3v is the raw metric and
// get_e3v's  e3v * ivmask  (:581-593) is formed here, in r4, before the product with e1v.
// Rows of w are w_row elements apart (npi: dense).
__global__ void moc_weight_kernel(const float *__restrict__ e1v, const float *__restrict__ e3v,
                                  const int8_t *__restrict__ vm, float *__restrict__ w, int64_t nij, int64_t n, int npi,
                                  int64_t w_row) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const float e3 = vm ? __fmul_rn(e3v[e], (float)vm[e]) : e3v[e];
        const int64_t row = e / npi;
        w[row * w_row + (e - row * npi)] = __fmul_rn(e1v[e % nij], e3);
    }
}

// bits(i,j) = sum_b (ibmask(b,i,j) != 0) << b ; any value outside {0,1} raises the flag.
__global__ void moc_bits_kernel(const int16_t *__restrict__ ibmask, uint8_t *__restrict__ bits,
                                int nb, int64_t nij, int *flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nij; e += stride) {
        uint32_t v = 0;
        bool bad = false;
        for (int b = 0; b < nb; ++b) {
            const int16_t mk = ibmask[e * nb + b];
            bad |= (mk != 0 && mk != 1);
            v |= (mk != 0 ? 1u : 0u) << b;
        }
        bits[e] = (uint8_t)v;
        if (bad) atomicOr(flag, 1);
        if (!flag[1 + v]) flag[1 + v] = 1;               // benign race: every writer stores 1
    }
}

// cls(i,j) = class index of the cell's basin-bit pattern (lut)
__global__ void moc_class_kernel(const uint8_t *__restrict__ bits, uint8_t *__restrict__ cls, int64_t nij,
                                 const uint8_t *__restrict__ lut) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nij; e += stride) cls[e] = lut[bits[e]];
}

// src/cdfmoc.f90:382-388: dmoc(:,jj,jk) = dmoc(:,jj,jk+1) + dmoc(:,jj,jk)/1.d6, jk = npk-1..1;
// level npk is the zero the reference leaves from its initialisation (:352).
__global__ void moc_cumsum_kernel(double *__restrict__ dmoc, int nbj, int npk, int npt) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nbj * npt) return;
    const int64_t t = idx / nbj, q = idx - t * nbj;
    double *base = dmoc + t * (int64_t)npk * nbj + q;
    double below = 0.0;
    base[(int64_t)(npk - 1) * nbj] = 0.0;
    // Sixteen levels at a time: their loads first, then the chain, then the stores (round 6).  Level by level -- load, divide,
    // add, store, and the next load only then, the stores being to the array the loads read -- a thread made 74 memory round
    // trips: at eORCA025 (87 MB moved) the kernel was those round trips, a cdfmoc call 34 us over its zonal kernel, 22 now; at
    // eORCA12 it is its 260 MB (43 us before and after).  The same operations in the same order.
    constexpr int B = 16;
    for (int j0 = npk - 2; j0 >= 0; j0 -= B) {
        double v[B];
#pragma unroll
        for (int i = 0; i < B; ++i) v[i] = base[(int64_t)max(j0 - i, 0) * nbj];
#pragma unroll
        for (int i = 0; i < B; ++i)
            if (j0 - i >= 0) {                                    // (uniform)
                below = below + v[i] / 1.0e6;
                v[i] = below;
            }
#pragma unroll
        for (int i = 0; i < B; ++i)
            if (j0 - i >= 0) base[(int64_t)(j0 - i) * nbj] = v[i];
    }
}


// ------------------------------------------------------------------------------------ cdfmoc -decomp
// src/cdfmoc.f90:360-365 then :397-401: dvbt = sum_k e3v*zv*1.d0 (r4 product, r8 sum, k = 1..npk-1 in
// order), hdep = sum_k e3v (r4), dvbt = dvbt / hdep where hdep /= 0, else 0.
__global__ void dec_vmean_kernel(const float *__restrict__ e3v, const float *__restrict__ zv, int64_t nij, int npk,
                                 double *__restrict__ dvbt, float *__restrict__ hdep) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    double d = 0.0;
    float h = 0.f;
    for (int k = 0; k < npk - 1; ++k) {
        const float e = e3v[(int64_t)k * nij + p];
        d = __dadd_rn(d, (double)__fmul_rn(e, zv[(int64_t)k * nij + p]));
        h = __fadd_rn(h, e);
    }
    hdep[p] = h;
    dvbt[p] = (h != 0.f) ? __ddiv_rn(d, (double)h) : 0.0;
}

// :480-484: the same division for the mean of the geostrophic velocity accumulated by dec_shear_kernel
__global__ void dec_divide_kernel(double *__restrict__ dvbt, const float *__restrict__ hdep, int64_t nij) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nij) return;
    const float h = hdep[p];
    dvbt[p] = (h != 0.f) ? __ddiv_rn(dvbt[p], (double)h) : 0.0;
}

// :403-414 and :489-502: raw(b,jj,jk) = - sum_ji (e1v*e3v*ibmask(b)) * dvbt(ji,jj): the r4 weight times
// an r8 field that does not depend on the level.  One workgroup per (jj,jk) row, basins selected by
// bit; level npk (index npk-1) is written as 0.  Fixed reduction order.
template <int NB>
__global__ void __launch_bounds__(256) dec_zonal_r8_kernel(const float *__restrict__ w, const uint8_t *__restrict__ bits,
                                                            const double *__restrict__ field, int npi, int npj, int npk,
                                                            int nb, double *__restrict__ raw, int64_t w_row) {
    __shared__ double scratch[NB][4];
    const int row = blockIdx.x, jk = row / npj, jj = row - jk * npj;
    double acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.0;
    if (jk < npk - 1) {
        const int64_t r2 = (int64_t)jj * npi, r3 = (int64_t)row * w_row;
        for (int i = threadIdx.x; i < npi; i += 256) {
            const double t = __dmul_rn((double)w[r3 + i], field[r2 + i]);
            const double tz = __dmul_rn(t, 0.0);          // ibmask = 0: a zero, or the NaN of 0*NaN / 0*Inf
            const uint32_t bb = bits[r2 + i];
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] += ((bb >> b) & 1u) ? t : tz;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        double s = acc[b];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) scratch[b][wave] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) {
        const int b = threadIdx.x;
        const double s = ((scratch[b][0] + scratch[b][1]) + scratch[b][2]) + scratch[b][3];
        raw[((int64_t)jk * npj + jj) * nb + b] = -s;
    }
}

// :439-466 for ONE level jk (launched from the bottom level up, in order).  sig8 = sigmai(T,S,gdept(jk))
// of the level (r8); zsig0 = REAL(sig8 * itmask) is formed on the fly for the six points of the stencil.
// dvgeo is updated in place (the reference's iup/ido pair collapses: each cell only reads its own
// previous value); zvgeo(:,:,jk) receives the r4 geostrophic velocity, the border keeps the velocity of
// level npk-1 (the reference never assigns the border of zv, see the oracle); dvbt += e3v*zv*1.d0.
__global__ void dec_shear_kernel(const double *__restrict__ sig8, const uint8_t *__restrict__ tm,
                                 const uint8_t *__restrict__ um, const float *__restrict__ e1u,
                                 const float *__restrict__ zcoef, const uint8_t *__restrict__ bits,
                                 const float *__restrict__ e3v, const float *__restrict__ zvlast, int npi, int npj,
                                 double *__restrict__ dvgeo, float *__restrict__ zvgeo, double *__restrict__ dvbt) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)npi * npj) return;
    const int jj = (int)(p / npi), ji = (int)(p - (int64_t)jj * npi);
    float zv;
    if (ji >= 1 && ji < npi - 1 && jj >= 1 && jj < npj - 1) {
        auto sg = [&](int64_t q) { return (float)__dmul_rn(sig8[q], (double)tm[q]); };        // :443
        const int64_t pn = p + npi;
        const int isum = max(1, (int)um[pn - 1] + (int)um[pn] + (int)um[p - 1] + (int)um[p]);
        const float zmsv = __fdiv_rn(1.f, (float)isum);                                          // :449
        float g = __fdiv_rn(__fmul_rn(__fsub_rn(sg(pn), sg(pn - 1)), (float)um[pn - 1]), e1u[pn - 1]);
        g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(sg(pn + 1), sg(pn)), (float)um[pn]), e1u[pn]));
        g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(sg(p), sg(p - 1)), (float)um[p - 1]), e1u[p - 1]));
        g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(sg(p + 1), sg(p)), (float)um[p]), e1u[p]));   // :450-453
        double inc = __dmul_rn((double)zcoef[p], (double)g);
        inc = __dmul_rn(inc, (double)zmsv);
        inc = __dmul_rn(inc, (double)(bits[p] & 1u));                                            // ibmask(npglo,ji,jj)
        inc = __dmul_rn(inc, (double)e3v[p]);
        const double old = dvgeo[p], now = __dadd_rn(old, inc);                                  // :457
        dvgeo[p] = now;
        zv = (float)__dmul_rn(0.5, __dadd_rn(now, old));                                         // :459
    } else {
        zv = zvlast[p];
    }
    zvgeo[p] = zv;
    dvbt[p] = __dadd_rn(dvbt[p], (double)__fmul_rn(e3v[p], zv));                                 // :466
}

// gdept(jk) over level jk, jk = 0 .. nlev-1 (the per-point reference depths of cdft_eos_sigmai2d)
__global__ void dec_pref_kernel(const float *__restrict__ gdept, int64_t nij, float *__restrict__ pref3d) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nij) pref3d[(int64_t)blockIdx.y * nij + p] = gdept[blockIdx.y];
}

// dec_shear_kernel for ALL levels in one launch: a thread walks its column from level npk-1 up (jk = npk-2 .. 0) with dvgeo
// and the depth sum dvbt in registers; sig8 is sigmai of the levels 0 .. npk-2, formed by ONE launch with per-point reference
// depths.  Per cell and level the statements and their order are those of the kernel above (74 launches of it and 74 of the
// EOS on a 1.7 M-cell plane were 4.3 ms per record at eORCA025).
__global__ void __launch_bounds__(256) dec_shear_all_kernel(const double *__restrict__ sig8, const uint8_t *__restrict__ tm,
                                                            const uint8_t *__restrict__ um, const float *__restrict__ e1u,
                                                            const float *__restrict__ zcoef, const uint8_t *__restrict__ bits,
                                                            const float *__restrict__ e3v, const float *__restrict__ zvlast, int npi, int npj,
                                                            int npk, double *__restrict__ dvgeo_out, float *__restrict__ zvgeo,
                                                            double *__restrict__ dvbt_out) {
    const int ji = blockIdx.x * blockDim.x + threadIdx.x, jj = blockIdx.y;
    if (ji >= npi) return;
    const int64_t nij = (int64_t)npi * npj, p = (int64_t)jj * npi + ji, pn = p + npi;
    const bool inner = ji >= 1 && ji < npi - 1 && jj >= 1 && jj < npj - 1;
    const float zl = zvlast[p];
    float e_nw = 1.f, e_n = 1.f, e_w = 1.f, e_c = 1.f;
    double coef = 0.0, bit0 = 0.0;
    if (inner) { e_nw = e1u[pn - 1]; e_n = e1u[pn]; e_w = e1u[p - 1]; e_c = e1u[p]; coef = (double)zcoef[p]; bit0 = (double)(bits[p] & 1u); }
    double dvgeo = 0.0, dvbt = 0.0;
    for (int jk = npk - 2; jk >= 0; --jk) {
        const int64_t o = (int64_t)jk * nij;
        const double *sg8 = sig8 + o;
        const uint8_t *tmk = tm + o, *umk = um + o;
        const float e3 = e3v[o + p];
        float zv;
        if (inner) {
            auto sg = [&](int64_t q) { return (float)__dmul_rn(sg8[q], (double)tmk[q]); };        // :443
            const int isum = max(1, (int)umk[pn - 1] + (int)umk[pn] + (int)umk[p - 1] + (int)umk[p]);
            const float zmsv = __fdiv_rn(1.f, (float)isum);                                          // :449
            float g = __fdiv_rn(__fmul_rn(__fsub_rn(sg(pn), sg(pn - 1)), (float)umk[pn - 1]), e_nw);
            g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(sg(pn + 1), sg(pn)), (float)umk[pn]), e_n));
            g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(sg(p), sg(p - 1)), (float)umk[p - 1]), e_w));
            g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(sg(p + 1), sg(p)), (float)umk[p]), e_c));   // :450-453
            double inc = __dmul_rn(coef, (double)g);
            inc = __dmul_rn(inc, (double)zmsv);
            inc = __dmul_rn(inc, bit0);                                                              // ibmask(npglo,ji,jj)
            inc = __dmul_rn(inc, (double)e3);
            const double old = dvgeo, now = __dadd_rn(old, inc);                                     // :457
            dvgeo = now;
            zv = (float)__dmul_rn(0.5, __dadd_rn(now, old));                                         // :459
        } else {
            zv = zl;
        }
        zvgeo[o + p] = zv;
        dvbt = __dadd_rn(dvbt, (double)__fmul_rn(e3, zv));                                           // :466
    }
    dvgeo_out[p] = dvgeo;
    dvbt_out[p] = dvbt;
}

// The same walk with the stencil's neighbours taken from the neighbouring LANES instead of loaded (round 5).  In the kernel
// above a cell and level is 18 vector memory instructions -- six r8 and six byte loads of the neighbours' density and tmask,
// four byte loads of umask, e3v, the store -- for 18 algorithmic bytes: the address units bound it (requesting a level ahead
// changed nothing, profiles/r05).  Here a lane loads sigma and tmask of ITS column on rows jj and jj+1 and umask there, forms
// the two r4 values REAL(sig8 * itmask) (:443) and hands them and its umask bytes east and west through the lanes: 8 memory
// instructions.  A wave covers 62 columns; its lanes 0 and 63 stand on the columns next to them and only give.  Every
// statement on a cell is the one above, in its order.
constexpr int kShearCols = 62;
__global__ void __launch_bounds__(256) dec_shear_share_kernel(const double *__restrict__ sig8, const uint8_t *__restrict__ tm,
                                                              const uint8_t *__restrict__ um, const float *__restrict__ e1u,
                                                              const float *__restrict__ zcoef, const uint8_t *__restrict__ bits,
                                                              const float *__restrict__ e3v, const float *__restrict__ zvlast, int npi, int npj,
                                                              int npk, double *__restrict__ dvgeo_out, float *__restrict__ zvgeo,
                                                              double *__restrict__ dvbt_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jj = blockIdx.y;
    const int c0 = (blockIdx.x * 4 + wave) * kShearCols;           // the wave's first column
    if (c0 >= npi) return;                                           // (whole wave: the lanes only talk inside a wave)
    const int ji = c0 - 1 + lane;                                    // this lane's column; lanes 0 and 63: the neighbours'
    const int jic = min(max(ji, 0), npi - 1);                        // (an address inside the row for the lanes that stand outside it)
    const bool owns = lane >= 1 && lane <= kShearCols && ji < npi;   // the lane writes its column
    const bool rowin = jj >= 1 && jj < npj - 1;                      // (uniform) the row has cells of the stencil
    const bool inner = rowin && ji >= 1 && ji < npi - 1;
    const int64_t nij = (int64_t)npi * npj, p = (int64_t)jj * npi + jic, pn = rowin ? p + npi : p;
    const float zl = zvlast[p];
    float e_nw = 1.f, e_n = 1.f, e_w = 1.f, e_c = 1.f;
    double coef = 0.0, bit0 = 0.0;
    if (inner) { e_nw = e1u[pn - 1]; e_n = e1u[pn]; e_w = e1u[p - 1]; e_c = e1u[p]; coef = (double)zcoef[p]; bit0 = (double)(bits[p] & 1u); }
    double dvgeo = 0.0, dvbt = 0.0;
    for (int jk = npk - 2; jk >= 0; --jk) {
        const int64_t o = (int64_t)jk * nij;
        const float e3 = e3v[o + p];
        float zv = zl;
        if (rowin) {                                                 // (uniform: every lane of the block takes it, the lanes talk)
            const float zc = (float)__dmul_rn(sig8[o + p], (double)tm[o + p]);       // :443, this column on row jj ...
            const float zn = (float)__dmul_rn(sig8[o + pn], (double)tm[o + pn]);     // ... and on row jj+1
            const int uc = (int)um[o + p], un = (int)um[o + pn];
            const float zc_w = __shfl_up(zc, 1), zc_e = __shfl_down(zc, 1), zn_w = __shfl_up(zn, 1), zn_e = __shfl_down(zn, 1);
            const int uc_w = __shfl_up(uc, 1), un_w = __shfl_up(un, 1);
            if (inner) {
                const int isum = max(1, un_w + un + uc_w + uc);
                const float zmsv = __fdiv_rn(1.f, (float)isum);                                      // :449
                float g = __fdiv_rn(__fmul_rn(__fsub_rn(zn, zn_w), (float)un_w), e_nw);
                g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(zn_e, zn), (float)un), e_n));
                g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(zc, zc_w), (float)uc_w), e_w));
                g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fsub_rn(zc_e, zc), (float)uc), e_c));       // :450-453
                double inc = __dmul_rn(coef, (double)g);
                inc = __dmul_rn(inc, (double)zmsv);
                inc = __dmul_rn(inc, bit0);                                                          // ibmask(npglo,ji,jj)
                inc = __dmul_rn(inc, (double)e3);
                const double old = dvgeo, now = __dadd_rn(old, inc);                                 // :457
                dvgeo = now;
                zv = (float)__dmul_rn(0.5, __dadd_rn(now, old));                                     // :459
            }
        }
        if (owns) zvgeo[o + p] = zv;
        dvbt = __dadd_rn(dvbt, (double)__fmul_rn(e3, zv));                                           // :466
    }
    if (owns) { dvgeo_out[p] = dvgeo; dvbt_out[p] = dvbt; }
}

// :416-419, :505-516 per (basin,jj) column: cumulated sums of the barotropic and shear parts and the
// ageostrophic remainder.  sh is IN/OUT (the reference accumulates onto what the array holds, :471).
__global__ void dec_finish_kernel(const double *__restrict__ tot, const double *__restrict__ raw_bt,
                                  const double *__restrict__ raw_sh, const double *__restrict__ raw_btw,
                                  int nbj, int npk, double *__restrict__ sh, double *__restrict__ bt,
                                  double *__restrict__ ag) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nbj) return;
    const int64_t top = (int64_t)(npk - 1) * nbj + q;
    double b = 0.0;                                          // dmoc_bt(:,:,npk) stays at its initial 0
    double s = __dadd_rn(sh[top], -raw_btw[top]);            // :505 at level npk (dmoc_btw is 0 there)
    bt[top] = b; sh[top] = s;
    ag[top] = __dadd_rn(__dadd_rn(tot[top], -s), -b);
    for (int jk = npk - 2; jk >= 0; --jk) {
        const int64_t e = (int64_t)jk * nbj + q;
        b = __dadd_rn(b, __ddiv_rn(raw_bt[e], 1.0e6));                                           // :418
        const double r = __dadd_rn(__dadd_rn(sh[e], raw_sh[e]), -raw_btw[e]);                    // :471, :505
        s = __dadd_rn(s, __ddiv_rn(r, 1.0e6));                                                   // :509
        bt[e] = b; sh[e] = s;
        ag[e] = __dadd_rn(__dadd_rn(tot[e], -s), -b);                                            // :516
    }
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// pick waves per row and float4 vectors per lane for a row of npi cells
// class mode: *nwaves waves x *nv vectors make a SEGMENT of the row (one workgroup); the row has ceil(nvec / segment) of them
int row_segments(const cdft_moc_plan *p, int nwaves, int nv, int mode) {
    const int nvec = (p->npi + 6) / 4;
    return mode == kModeClass ? (nvec + kWave * nwaves * nv - 1) / (kWave * nwaves * nv) : 1;
}

void choose_shape(const cdft_moc_plan *p, bool aligned, int *nwaves, int *nv, int *mode) {
    const int nvec = (p->npi + 6) / 4;             // worst-case aligned vectors covering a row
    const bool class_ok = p->nclass > 0 && p->nclass <= kMaxClass;
    int md = p->tune_mode ? p->tune_mode : ((p->nbasins > 1 && class_ok) ? kModeClass : kModeSelect);
    if (md == kModeClass && (p->nbasins == 1 || !class_ok)) md = kModeSelect;
    // records whose base or stride is not a multiple of 16 bytes (eORCA025: 130 537 050 cells per record): the class
    // kernel reads them with 16-byte loads at r4 alignment, whatever the number of basins; the select kernel would
    // fall back to scalar loads
    if (!aligned) md = (class_ok && p->npi >= 4 && p->tune_mode != kModeSelect) ? kModeClass : kModeSelect;
    // rows longer than one workgroup of the select / fma kernel covers (16 waves x 6, 5 or 2 vectors; 4 for unaligned
    // records: 16k .. 24k cells): only the class kernel cuts a row into segments -- it takes them whatever the number of
    // basins, as long as the masks have at most kMaxClass bit patterns
    {
        const int vcap = !aligned ? 4 : (md == kModeFma ? 2 : (p->nbasins > 1 ? kBigNV - 1 : kBigNV));
        if (md != kModeClass && p->tune_mode == 0 && nvec > kWave * kMaxWaves * vcap && class_ok && p->npi >= 4) md = kModeClass;
    }
    int k, v;
    if (!aligned && md != kModeClass) {
        md = kModeSelect;
        v  = 4;
        k  = ceil_div(nvec, kWave * v);
    } else if (p->tune_waves > 0 || p->tune_nv > 0) {
        if (p->tune_waves > 0 && p->tune_nv > 0) { k = p->tune_waves; v = p->tune_nv; }
        else if (p->tune_waves > 0) { k = p->tune_waves; v = ceil_div(nvec, kWave * k); }
        else { v = p->tune_nv; k = ceil_div(nvec, kWave * v); }
    } else {
        if (md == kModeClass) {
            // vectors per lane by workgroup size (256 / 512 / 1024 threads leave 512 / 256 / 128 VGPRs per lane); records
            // that are not 16-byte aligned cost more registers per vector (measured: no spill with these)
            const int v_small = aligned ? 9 : 6;
            k = ceil_div(nvec, kWave * v_small);                   // 9: measured best at npiglo = 4322
            // rows no 256-thread workgroup covers (> 9.2k cells, eORCA36): segments of kSegWaves waves x 9 vectors, the
            // shape that streams best, one workgroup each (6 waves x 9 in one 512-thread workgroup: 70 % of the HBM
            // peak at npiglo = 12960, segments: see DESIGN 4.1)
            if (k > kSmallWaves) {
                // ... with the vectors per lane (9, 10 or 8; 6 or 5 for unaligned records) that leave the fewest idle
                // lanes in the last segment: 12960 cells = 3 x 2 x 64 x 9 at 94 %, 10000 cells = 2 x 2 x 64 x 10 at 98 %
                const int cand_a[3] = {9, 10, 8}, cand_u[2] = {6, 5};
                const int *cand = aligned ? cand_a : cand_u;
                const int ncand = aligned ? 3 : 2;
                double best = 0.0;
                v = cand[0];
                for (int c = 0; c < ncand; ++c) {
                    const int cap = kWave * kSegWaves * cand[c], ns = ceil_div(nvec, cap);
                    const double fill = (double)nvec / ((double)ns * cap);
                    if (fill > best + 1e-9) { best = fill; v = cand[c]; }
                }
                *nwaves = kSegWaves; *nv = v; *mode = md;
                return;
            }
        } else {
            int vmax = (md == kModeFma) ? 4 : 8;
            k = ceil_div(nvec, kWave * vmax);
            if (k > kSmallWaves) {          // long rows: 1024-thread workgroups, fewer registers
                // (several basin accumulators leave room for one vector less than a single one does)
                vmax = (md == kModeFma) ? 2 : (p->nbasins > 1 ? kBigNV - 1 : kBigNV);
                k = ceil_div(nvec, kWave * vmax);
            }
        }
        v = ceil_div(nvec, kWave * k);
    }
    *nwaves = k; *nv = v; *mode = md;
}

}  // namespace

// the zonal kernel of one call: raw -(zonal sums) of npt records into d_out (nbasins,npj,npk,npt),
// the class kernel's workgroup -> row map (moc_zonal.hpp): on unless CDFT_MOC_XCDMAP=0 or CDFT_XCD_BANDS=0 (A/B hooks) or the rows are few
static int zonal_xcd_map(int npj) {
    const char *e = getenv("CDFT_MOC_XCDMAP");
    return (npj >= 64 && !(e && e[0] == '0') && xcd_bands_on()) ? 1 : 0;
}

// levels 1..npk-1 (level npk is left to the caller)
// w_row / z_row: elements between consecutive (jj,jk) rows of the weights / of a record (npi: dense arrays)
// nlev >= 0: a window of nlev levels (w / d_zv are the window's, d_out is shifted to its first level by the caller)
static int zonal_launch(cdft_moc_plan *p, const float *w, int64_t w_row, const float *d_zv, int64_t rec_stride, int64_t z_row, int npt,
                        double *d_out, hipStream_t st, bool profile, int nlev = -1) {
    ZonalArgs a;
    a.w = w; a.bits = p->bits; a.cls = p->cls; a.nclass = p->nclass;
    for (int c = 0; c < kMaxClass; ++c) a.pat[c] = p->pat[c];
    a.zv = d_zv; a.rec_stride = rec_stride; a.npt = npt;
    a.w_row = w_row; a.z_row = z_row;
    a.out = d_out; a.npi = p->npi; a.npj = p->npj; a.npk = p->npk; a.nbasins = p->nbasins;
    a.nlev = nlev >= 0 ? nlev : p->npk - 1;
    a.rows_per_wg = 1;            // (2, 4, 8, 16 measured at eORCA025 and eORCA12: no gain, the kernel is not launch-bound)
    a.xcd_map = zonal_xcd_map(p->npj);
    // vectors aligned to the arrays: every record 16-byte aligned, and a row begins the same number of cells past a
    // 16-byte boundary in w and in the records
    const bool aligned = ((uintptr_t)d_zv % 16 == 0) && (rec_stride % 4 == 0) && ((z_row - w_row) % 4 == 0) &&
                         ((uintptr_t)w % 16 == 0);
    a.aligned = aligned ? 1 : 0;
    int k, v, md;
    choose_shape(p, aligned, &k, &v, &md);
    p->last_aligned = aligned;
    p->last_pool = (w != p->w);
    p->last_slot = rec_stride;
    if (k > kMaxWaves || v > kMaxNV)
        return fail(CDFT_ERR_ARG, "cdft_moc_compute: a row of %d cells needs %d waves x %d vectors of the %s kernel (rows this long "
                    "take basin masks with at most %d bit patterns: the class kernel)", p->npi, k, v, md == kModeFma ? "fma" : "select",
                    kMaxClass);
    if (md != kModeClass && k * v * kWave < (p->npi + 6) / 4)
        return fail(CDFT_ERR_ARG, "cdft_moc_compute: %d waves x %d vectors do not cover a row of %d (only the class kernel cuts rows)", k, v,
                    p->npi);
    a.nseg = row_segments(p, k, v, md);
    if (a.nseg > 1 || a.rows_per_wg != 1 || md != kModeClass) a.xcd_map = 0;
    a.seg_out = nullptr;
    a.seg_stride = (int64_t)p->nbasins * p->npj * p->npk * npt;
    if (a.nseg > 1) {
        CDFT_TRY(p->seg_out.reserve((size_t)a.nseg * a.seg_stride * sizeof(double)));
        a.seg_out = (double *)p->seg_out.ptr;
    }
    hipEvent_t ev1 = nullptr;
    if (profile && p->profiling) {
        if (p->ev_used + 2 > p->ev.size()) {
            hipEvent_t e0, e1;
            CDFT_HIP(hipEventCreate(&e0));
            CDFT_HIP(hipEventCreate(&e1));
            p->ev.push_back(e0);
            p->ev.push_back(e1);
        }
        CDFT_HIP(hipEventRecord(p->ev[p->ev_used], st));
        ev1 = p->ev[p->ev_used + 1];
        p->ev_used += 2;
    }
    if (md == kModeClass)     CDFT_TRY(launch_zonal_class(a, k, v, st));
    else if (p->nbasins == 1) CDFT_TRY(launch_zonal_nb1(a, k, v, aligned, md, st));
    else if (p->nbasins <= 5) CDFT_TRY(launch_zonal_nb5(a, k, v, aligned, md, st));
    else                      CDFT_TRY(launch_zonal_nb8(a, k, v, aligned, md, st));
    if (ev1) CDFT_HIP(hipEventRecord(ev1, st));
    p->n_launch += 1; p->n_launch_rec += npt;

    return CDFT_OK;
}

extern "C" {

int cdft_moc_create(cdft_moc_plan **plan, int npiglo, int npjglo, int npk, int nbasins) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_moc_create: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 2)
        return fail(CDFT_ERR_ARG, "cdft_moc_create: bad sizes npiglo=%d npjglo=%d npk=%d (npk >= 2)",
                    npiglo, npjglo, npk);
    if (nbasins < 1 || nbasins > 8)
        return fail(CDFT_ERR_ARG, "cdft_moc_create: nbasins=%d outside 1..8", nbasins);
    if (npiglo > (1 << 24))                         // (rows of any length: the class kernel cuts them into segments)
        return fail(CDFT_ERR_ARG, "cdft_moc_create: npiglo=%d exceeds the supported row length %d", npiglo, 1 << 24);
    if ((int64_t)npjglo * (npk - 1) > 0x7fffffffLL)
        return fail(CDFT_ERR_ARG, "cdft_moc_create: npjglo*(npk-1) exceeds the grid limit");
    CDFT_TRY(require_device());
    cdft_moc_plan *p = new cdft_moc_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk; p->nbasins = nbasins;
    const size_t n3 = (size_t)npiglo * npjglo * npk;
    // (the 3-D weights come with the first cdft_moc_set_mesh*: a program on the stream route never holds them)
    (void)n3;
    if (hipMalloc((void **)&p->bits, (size_t)npiglo * npjglo + 64) != hipSuccess ||
        hipMalloc((void **)&p->cls, (size_t)npiglo * npjglo + 64 + 256) != hipSuccess ||
        hipMalloc((void **)&p->flag, 257 * sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        cdft_moc_destroy(p);
        return fail(CDFT_ERR_ALLOC, "cdft_moc_create: device allocation of %zu bytes failed",
                    (size_t)npiglo * npjglo * 2);
    }
    *plan = p;
    return CDFT_OK;
}

int cdft_moc_create_sharded(cdft_moc_plan **plan, int npiglo, int npjglo, int npk, int nbasins, int ndev, const int *devices) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_moc_create_sharded: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 2 || nbasins < 1 || nbasins > 8)
        return fail(CDFT_ERR_ARG, "cdft_moc_create_sharded: bad sizes %d x %d x %d, %d basins", npiglo, npjglo, npk, nbasins);
    cdft_moc_plan *p = new cdft_moc_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk; p->nbasins = nbasins;
    int rc = shard::make(&p->sh, npiglo, npjglo, npk, nbasins, ndev, devices, 0, 6);
    if (rc == CDFT_OK) rc = shard::moc_parts(*p->sh);
    if (rc != CDFT_OK) { cdft_moc_destroy(p); return rc; }
    *plan = p;
    return CDFT_OK;
}

int cdft_moc_set_mesh(cdft_moc_plan *p, const float *e1v, const float *e3v, const int16_t *ibmask,
                      int memspace, void *stream) {
    TraceRange trace_("cdft_moc_set_mesh");
    return cdft_moc_set_mesh_vmask(p, e1v, e3v, nullptr, ibmask, memspace, stream);
}

int cdft_moc_set_mesh_vmask(cdft_moc_plan *p, const float *e1v, const float *e3v, const int8_t *vmask,
                            const int16_t *ibmask, int memspace, void *stream) {
    if (!p || !e1v || !e3v || !ibmask) return fail(CDFT_ERR_ARG, "cdft_moc_set_mesh: NULL argument");
    if (p->sh) {
        if (memspace == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));   // the blocks use their own streams
        CDFT_TRY(shard::moc_set_mesh(*p->sh, e1v, e3v, vmask, ibmask));
        p->mesh_set = true;
        return CDFT_OK;
    }
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj, n3 = nij * p->npk;
    if (!p->w && hipMalloc((void **)&p->w, n3 * sizeof(float) + 64) != hipSuccess) {
        (void)hipGetLastError();
        p->w = nullptr;
        return fail(CDFT_ERR_ALLOC, "cdft_moc_set_mesh: device allocation of %zu bytes (the weights) failed", (size_t)n3 * sizeof(float));
    }
    const void *d_e1v, *d_e3v, *d_msk, *d_vm = nullptr;
    CDFT_TRY(to_device(e1v, nij * sizeof(float), memspace, p->st_e1v, st, &d_e1v));
    CDFT_TRY(to_device(e3v, n3 * sizeof(float), memspace, p->st_e3v, st, &d_e3v));
    if (vmask) CDFT_TRY(to_device(vmask, (size_t)n3, memspace, p->st_vm, st, &d_vm));
    CDFT_TRY(to_device(ibmask, nij * p->nbasins * sizeof(int16_t), memspace, p->st_mask, st, &d_msk));
    CDFT_HIP(hipMemsetAsync(p->w + n3, 0, 64, st));
    p->pool_w = false;                                    // (the pool's copy of the weights is the old mesh's)
    const int threads = 256;
    int blocks = (int)((n3 + threads - 1) / threads);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(moc_weight_kernel, dim3(blocks), dim3(threads), 0, st, (const float *)d_e1v,
                       (const float *)d_e3v, (const int8_t *)d_vm, p->w, nij, n3, p->npi, (int64_t)p->npi);
    CDFT_LAUNCH_CHECK();
    CDFT_TRY(p->e1v2d.reserve((size_t)nij * 4));                       // (kept for the stream route, 62 MB at eORCA12)
    CDFT_HIP(hipMemcpyAsync(p->e1v2d.ptr, d_e1v, (size_t)nij * 4, hipMemcpyDeviceToDevice, st));
    const int rc_b = basins_from(p, (const int16_t *)d_msk, st);
    // the staged copies of host arrays are no longer needed (e3v is as large as the weight)
    p->st_e1v.release(); p->st_e3v.release(); p->st_mask.release(); p->st_vm.release();
    p->mesh_set = (rc_b == CDFT_OK);
    return rc_b;
}

// bits, classes and their table from ibmask (nbasins,npi,npj) i2 in HBM; p->basins_set says whether it held 0/1 only
static int basins_from(cdft_moc_plan *p, const int16_t *d_msk, hipStream_t st) {
    const int64_t nij = (int64_t)p->npi * p->npj;
    const int threads = 256;
    int blocks = (int)((nij + threads - 1) / threads);
    if (blocks > 256 * 16) blocks = 256 * 16;
    p->basins_set = false;
    CDFT_HIP(hipMemsetAsync(p->flag, 0, 257 * sizeof(int), st));
    hipLaunchKernelGGL(moc_bits_kernel, dim3(blocks), dim3(threads), 0, st, d_msk, p->bits, p->nbasins, nij, p->flag);
    CDFT_LAUNCH_CHECK();
    int flags[257];
    CDFT_HIP(hipMemcpyAsync(flags, p->flag, sizeof(flags), hipMemcpyDeviceToHost, st));
    CDFT_HIP(hipStreamSynchronize(st));
    const int flag = flags[0];
    // class mode: the distinct bit patterns, in increasing order (the empty pattern included: its cells
    // still contribute the reference's 0*zv terms, see moc_zonal.hpp)
    uint8_t lut[256];
    p->nclass = 0;
    for (int v = 0; v < 256; ++v) lut[v] = 0xFF;
    for (int v = 0; v < 256; ++v)
        if (flags[1 + v]) {
            if (p->nclass < kMaxClass) { p->pat[p->nclass] = (uint8_t)v; lut[v] = (uint8_t)p->nclass; }
            ++p->nclass;
        }
    if (p->nclass > kMaxClass) p->nclass = 0;
    if (p->nclass > 0 && !flag) {
        uint8_t *dlut = p->cls + nij + 64;                       // the table lives behind the map
        CDFT_HIP(hipMemcpyAsync(dlut, lut, 256, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(moc_class_kernel, dim3(blocks), dim3(threads), 0, st, p->bits, p->cls, nij, dlut);
        CDFT_LAUNCH_CHECK();
        CDFT_HIP(hipStreamSynchronize(st));                      // lut[] is a stack array
    }
    if (flag) return fail(CDFT_ERR_ARG, "cdft_moc_set_mesh: ibmask holds values other than 0/1");
    p->basins_set = true;
    return CDFT_OK;
}

extern "C" int cdft_moc_set_basins(cdft_moc_plan *p, const float *e1v, const int16_t *ibmask, int memspace, void *stream) {
    TraceRange trace_("cdft_moc_set_basins");
    if (!p || !e1v || !ibmask) return fail(CDFT_ERR_ARG, "cdft_moc_set_basins: NULL argument");
    if (p->sh) return fail(CDFT_ERR_STATE, "cdft_moc_set_basins: single-device plans only (the stream route)");
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj;
    const void *d_e1v, *d_msk;
    CDFT_TRY(to_device(e1v, nij * sizeof(float), memspace, p->st_e1v, st, &d_e1v));
    CDFT_TRY(to_device(ibmask, nij * p->nbasins * sizeof(int16_t), memspace, p->st_mask, st, &d_msk));
    CDFT_TRY(p->e1v2d.reserve((size_t)nij * 4));
    CDFT_HIP(hipMemcpyAsync(p->e1v2d.ptr, d_e1v, (size_t)nij * 4, hipMemcpyDeviceToDevice, st));
    const int rc = basins_from(p, (const int16_t *)d_msk, st);
    p->st_e1v.release(); p->st_mask.release();
    // a rejected mask has already overwritten bits / cls / pat: a mesh set before must not let a later compute run on them
    if (rc != CDFT_OK) p->mesh_set = false;
    return rc;
}

int cdft_moc_set_tuning(cdft_moc_plan *p, int waves_per_row, int vec_per_lane, int mode) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_moc_set_tuning: NULL plan");
    if (p->sh) {
        for (void *q : p->sh->part) CDFT_TRY(cdft_moc_set_tuning((cdft_moc_plan *)q, waves_per_row, vec_per_lane, mode));
        return CDFT_OK;
    }
    if (waves_per_row < 0 || waves_per_row > kMaxWaves || vec_per_lane < 0 || vec_per_lane > kMaxNV ||
        mode < 0 || mode > 3)
        return fail(CDFT_ERR_ARG, "cdft_moc_set_tuning: out of range (%d,%d,%d)", waves_per_row,
                    vec_per_lane, mode);
    if (waves_per_row > 0 && vec_per_lane > 0 && mode != kModeClass && !(mode == 0 && p->nbasins > 1 && p->nclass > 0) &&
        waves_per_row * vec_per_lane * kWave < (p->npi + 6) / 4)                    // (the class kernel cuts the row into segments)
        return fail(CDFT_ERR_ARG, "cdft_moc_set_tuning: %d waves x %d vectors do not cover a row of %d",
                    waves_per_row, vec_per_lane, p->npi);
    p->tune_waves = waves_per_row; p->tune_nv = vec_per_lane; p->tune_mode = mode;
    return CDFT_OK;
}

int cdft_moc_describe(cdft_moc_plan *p, char *buf, size_t buflen) {
    if (!p || !buf) return fail(CDFT_ERR_ARG, "cdft_moc_describe: NULL argument");
    if (p->sh) return cdft_moc_describe((cdft_moc_plan *)p->sh->part[0], buf, buflen);
    int k, v, md;
    choose_shape(p, p->last_aligned, &k, &v, &md);       // the shape of the last launch (16-byte aligned records before any)
    const int nb = p->nbasins == 1 ? 1 : (p->nbasins <= 5 ? 5 : 8);
    const int nseg = row_segments(p, k, v, md);
    if (md == kModeClass && nseg > 1)
        snprintf(buf, buflen, "moc_zonal_class_kernel<NV=%d,KC=%d,ALIGNED=%d> row_segments=%d waves_per_segment=%d basin_patterns=%d", v,
                 kMaxClass, p->last_aligned ? 1 : 0, nseg, k, p->nclass);
    else if (md == kModeClass)
        snprintf(buf, buflen, "moc_zonal_class_kernel<NV=%d,KC=%d,ALIGNED=%d> waves_per_row=%d basin_patterns=%d xcd_bands=%d", v, kMaxClass,
                 p->last_aligned ? 1 : 0, k, p->nclass, zonal_xcd_map(p->npj) ? 1 : 0);
    else
        snprintf(buf, buflen, "moc_zonal_kernel<NV=%d,NB=%d,ALIGNED=%d,MODE=%s> waves_per_row=%d", v, nb, p->last_aligned ? 1 : 0,
                 md == kModeFma ? "fma" : "select", k);
    if (p->last_pool) {                                   // the layout the launch read: dense arrays (nothing said) or the pool
        const size_t n = strlen(buf);
        if (n + 1 < buflen) snprintf(buf + n, buflen - n, " layout=pool(slot=%lld)", (long long)p->last_slot);
    }
    return CDFT_OK;
}

int cdft_moc_compute(cdft_moc_plan *p, const float *zv, int64_t rec_stride, int npt, int zv_space,
                     double *dmoc, int dmoc_space, void *stream) {
    TraceRange trace_("cdft_moc_compute");
    if (!p || !zv || !dmoc) return fail(CDFT_ERR_ARG, "cdft_moc_compute: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_moc_compute: cdft_moc_set_mesh has not succeeded");
    const int64_t n3 = (int64_t)p->npi * p->npj * p->npk;
    if (npt < 1 || rec_stride < n3)
        return fail(CDFT_ERR_ARG, "cdft_moc_compute: npt=%d rec_stride=%lld (need npt>=1, stride>=%lld)",
                    npt, (long long)rec_stride, (long long)n3);
    if (p->sh) {                                                                 // any memory spaces; returns when done
        if (zv_space == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
        return shard::moc_compute(*p->sh, zv, rec_stride, npt, dmoc);   // (dense arrays per block: the blocks' pools are not touched)
    }
    hipStream_t st = as_stream(stream);
    const void *d_zv;
    CDFT_TRY(to_device(zv, ((size_t)(npt - 1) * rec_stride + n3) * sizeof(float), zv_space, p->st_zv,
                       st, &d_zv));
    if (p->staged) { p->staged = nullptr; p->staged_npt = 0; }     // (records kept as they came, CDFT_MOC_POOL=0: st_zv may just have been overwritten)
    return moc_compute_on(p, (const float *)d_zv, rec_stride, npt, dmoc, dmoc_space, st);
}

// zonal sums + cumulated sums of npt records that are in HBM, result to host or HBM
static int moc_compute_on(cdft_moc_plan *p, const float *d_zv, int64_t rec_stride, int npt, double *dmoc, int dmoc_space,
                          hipStream_t st, int64_t z_row, const float *w, int64_t w_row) {
    const size_t out_bytes = (size_t)p->nbasins * p->npj * p->npk * npt * sizeof(double);
    double *d_out = dmoc;
    if (dmoc_space == CDFT_MEM_HOST || dmoc_space == CDFT_MEM_PINNED) {
        CDFT_TRY(p->st_out.reserve(out_bytes));
        d_out = (double *)p->st_out.ptr;
    } else if (dmoc_space != CDFT_MEM_DEVICE) {
        return fail(CDFT_ERR_ARG, "cdft_moc_compute: bad memory-space tag %d", dmoc_space);
    }

    CDFT_TRY(zonal_launch(p, w ? w : p->w, w ? w_row : (int64_t)p->npi, d_zv, rec_stride, z_row > 0 ? z_row : (int64_t)p->npi, npt, d_out, st, true));

    const int nbj = p->nbasins * p->npj;
    const int64_t nthr = (int64_t)nbj * npt;
    hipLaunchKernelGGL(moc_cumsum_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, d_out,
                       nbj, p->npk, npt);
    CDFT_LAUNCH_CHECK();
    if (dmoc_space != CDFT_MEM_DEVICE) {
        CDFT_HIP(hipMemcpyAsync(dmoc, d_out, out_bytes, hipMemcpyDeviceToHost, st));
        if (dmoc_space == CDFT_MEM_HOST) CDFT_HIP(hipStreamSynchronize(st));     // (pinned: the caller synchronises)
    }
    return CDFT_OK;
}

extern "C" {

// CDFT_MOC_POOL=0: cdft_moc_stage keeps the records as they came (A/B hook, read at every call)
static bool pool_wanted() { const char *e = getenv("CDFT_MOC_POOL"); return !(e && e[0] == '0'); }

// rows of npi r4, src_row elements apart, into rows dst_row apart (16 KB rows at eORCA12: a workgroup per row)
__global__ void __launch_bounds__(256) moc_pool_put_kernel(const float *__restrict__ src, int64_t src_row, float *__restrict__ dst,
                                                           int64_t dst_row, int npi, int64_t nrows) {
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const float *s = src + r * src_row;
        float *d = dst + r * dst_row;
        for (int i = threadIdx.x; i < npi; i += 256) d[i] = s[i];
    }
}

// The (jj,jk) rows of one field -- nrows of them, src_row elements apart -- into slot `slot` of the pool's rows first_row...
// A source in this device's HBM is read by a kernel; anything else (host memory, another device's HBM) by a strided copy.
static int pool_put(cdft_moc_plan *p, int slot, int64_t first_row, const float *src, int64_t src_row, int64_t nrows, hipStream_t st) {
    const int64_t R = (int64_t)(p->pool_npt + 1) * p->pool_P;
    float *dst = (float *)p->pool.ptr + first_row * R + (int64_t)slot * p->pool_P;
    hipPointerAttribute_t at;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeDevice && at.device == cur) {
        const unsigned grid = (unsigned)(nrows < 65536 ? nrows : 65536);
        hipLaunchKernelGGL(moc_pool_put_kernel, dim3(grid), dim3(256), 0, st, src, src_row, dst, R, p->npi, nrows);
        CDFT_LAUNCH_CHECK();
        return CDFT_OK;
    }
    (void)hipGetLastError();
    CDFT_HIP(hipMemcpy2DAsync(dst, (size_t)R * 4, src, (size_t)src_row * 4, (size_t)p->npi * 4, (size_t)nrows, hipMemcpyDefault, st));
    return CDFT_OK;
}

// Lay the pool out for npt records (contents lost when the layout changes).
static int pool_layout(cdft_moc_plan *p, int npt, hipStream_t st) {
    const int64_t P = ((int64_t)p->npi + 3) & ~(int64_t)3, rows = (int64_t)p->npj * p->npk;
    const size_t bytes = (size_t)(rows * (npt + 1) * P) * 4 + 64;
    if (p->pool_npt != npt || p->pool_P != P || p->pool.bytes < bytes) {
        p->pool_npt = 0; p->pool_w = false;
        p->staged = nullptr; p->staged_npt = 0;
        CDFT_TRY(p->pool.reserve(bytes));
        // the cells between npi and P and the 64 bytes behind the last row are read (and masked): never uninitialised memory
        CDFT_HIP(hipMemsetAsync(p->pool.ptr, 0, bytes, st));
        p->pool_npt = npt; p->pool_P = P;
    }
    return CDFT_OK;
}
// ... and bring the current weights in
static int pool_prepare(cdft_moc_plan *p, int npt, hipStream_t st) {
    CDFT_TRY(pool_layout(p, npt, st));
    if (!p->pool_w) {
        CDFT_TRY(pool_put(p, 0, 0, p->w, p->npi, (int64_t)p->npj * p->npk, st));
        p->pool_w = true;
    }
    return CDFT_OK;
}

int cdft_moc_reserve(cdft_moc_plan *p, int npt, void *stream) {
    if (!p || npt < 1) return fail(CDFT_ERR_ARG, "cdft_moc_reserve: bad argument");
    if (p->sh) {
        int cur = 0, rc = CDFT_OK;
        (void)hipGetDevice(&cur);
        for (size_t b = 0; b < p->sh->part.size() && rc == CDFT_OK; ++b) {
            if (hipSetDevice(p->sh->dev[b]) != hipSuccess) { rc = fail(CDFT_ERR_HIP, "cdft_moc_reserve: hipSetDevice(%d) failed", p->sh->dev[b]); break; }
            rc = cdft_moc_reserve((cdft_moc_plan *)p->sh->part[b], npt, p->sh->st[b]);
            if (rc == CDFT_OK && hipStreamSynchronize(p->sh->st[b]) != hipSuccess) rc = fail(CDFT_ERR_HIP, "cdft_moc_reserve: stream of block %zu", b);
        }
        (void)hipSetDevice(cur);
        return rc;
    }
    if (!pool_wanted()) return CDFT_OK;
    return pool_layout(p, npt, as_stream(stream));
}

int cdft_moc_stage(cdft_moc_plan *p, const float *zv, int64_t rec_stride, int npt, int zv_space, void *stream) {
    TraceRange trace_("cdft_moc_stage");
    if (!p || !zv) return fail(CDFT_ERR_ARG, "cdft_moc_stage: NULL argument");
    const int64_t n3 = (int64_t)p->npi * p->npj * p->npk;
    if (npt < 1 || rec_stride < n3)
        return fail(CDFT_ERR_ARG, "cdft_moc_stage: npt=%d rec_stride=%lld (need npt>=1, stride>=%lld)", npt, (long long)rec_stride, (long long)n3);
    if (p->sh) {
        if (zv_space == CDFT_MEM_DEVICE) CDFT_HIP(hipStreamSynchronize(as_stream(stream)));
        CDFT_TRY(shard::moc_stage(*p->sh, zv, rec_stride, npt));
        p->staged_npt = npt;
        return CDFT_OK;
    }
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_moc_stage: cdft_moc_set_mesh has not succeeded (the weights are staged with the records)");
    hipStream_t st = as_stream(stream);
    p->staged = nullptr; p->staged_npt = 0; p->open_npt = 0;
    if (!pool_wanted()) {
        const void *d_zv;
        CDFT_TRY(to_device(zv, ((size_t)(npt - 1) * rec_stride + n3) * sizeof(float), zv_space, p->st_zv, st, &d_zv));
        p->staged = (const float *)d_zv; p->staged_stride = rec_stride; p->staged_npt = npt;
        return CDFT_OK;
    }
    if (zv_space != CDFT_MEM_DEVICE && zv_space != CDFT_MEM_HOST && zv_space != CDFT_MEM_PINNED)
        return fail(CDFT_ERR_ARG, "cdft_moc_stage: bad memory-space tag %d", zv_space);
    CDFT_TRY(pool_prepare(p, npt, st));
    for (int t = 0; t < npt; ++t)
        CDFT_TRY(pool_put(p, 1 + t, 0, zv + (int64_t)t * rec_stride, p->npi, (int64_t)p->npj * p->npk, st));
    if (zv_space == CDFT_MEM_HOST) CDFT_HIP(hipStreamSynchronize(st));            // (a pageable array may change once the call is back)
    p->staged_npt = npt;
    return CDFT_OK;
}

int cdft_moc_stage_rows(cdft_moc_plan *p, const float *zv, int64_t rec_stride, int64_t lev_stride, int npt, int zv_space, void *stream) {
    TraceRange trace_("cdft_moc_stage_rows");
    if (!p || !zv) return fail(CDFT_ERR_ARG, "cdft_moc_stage_rows: NULL argument");
    if (p->sh) return fail(CDFT_ERR_STATE, "cdft_moc_stage_rows: single-device plans only (a sharded plan cuts the domain itself: cdft_moc_stage)");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_moc_stage_rows: cdft_moc_set_mesh has not succeeded");
    const int64_t nij = (int64_t)p->npi * p->npj;
    if (npt < 1 || lev_stride < nij || rec_stride < (int64_t)(p->npk - 1) * lev_stride + nij)
        return fail(CDFT_ERR_ARG, "cdft_moc_stage_rows: npt=%d rec_stride=%lld lev_stride=%lld for levels of %lld cells", npt,
                    (long long)rec_stride, (long long)lev_stride, (long long)nij);
    hipStream_t st = as_stream(stream);
    p->staged = nullptr; p->staged_npt = 0;
    CDFT_TRY(pool_prepare(p, npt, st));
    for (int t = 0; t < npt; ++t)
        for (int k = 0; k < p->npk; ++k)
            CDFT_TRY(pool_put(p, 1 + t, (int64_t)k * p->npj, zv + (int64_t)t * rec_stride + (int64_t)k * lev_stride, p->npi, p->npj, st));
    if (zv_space == CDFT_MEM_HOST) CDFT_HIP(hipStreamSynchronize(st));
    p->staged_npt = npt;
    return CDFT_OK;
}

// The caller writes the records into the pool itself (a host feed whose H2D copies ARE the staging): _open lays the pool out
// and brings the weights in, _dest says where record t's rows start and how far apart they lie, _close declares them staged.
int cdft_moc_stage_open(cdft_moc_plan *p, int npt, void *stream) {
    TraceRange trace_("cdft_moc_stage_open");
    if (!p || npt < 1) return fail(CDFT_ERR_ARG, "cdft_moc_stage_open: bad argument");
    if (p->sh) return fail(CDFT_ERR_STATE, "cdft_moc_stage_open: single-device plans only (a sharded plan cuts the domain itself: cdft_moc_stage)");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_moc_stage_open: cdft_moc_set_mesh has not succeeded (the weights are staged with the records)");
    p->staged = nullptr; p->staged_npt = 0; p->open_npt = 0;
    CDFT_TRY(pool_prepare(p, npt, as_stream(stream)));
    p->open_npt = npt;
    return CDFT_OK;
}

int cdft_moc_stage_dest(cdft_moc_plan *p, int record, int level, void **dst, size_t *row_pitch_bytes) {
    if (!p || !dst || !row_pitch_bytes) return fail(CDFT_ERR_ARG, "cdft_moc_stage_dest: NULL argument");
    if (p->sh || p->open_npt < 1) return fail(CDFT_ERR_STATE, "cdft_moc_stage_dest: cdft_moc_stage_open has not succeeded");
    if (record < 0 || record >= p->open_npt || level < 0 || level >= p->npk)
        return fail(CDFT_ERR_ARG, "cdft_moc_stage_dest: record %d of %d, level %d of %d", record, p->open_npt, level, p->npk);
    const int64_t R = (int64_t)(p->pool_npt + 1) * p->pool_P;
    *dst = (float *)p->pool.ptr + (int64_t)level * p->npj * R + (int64_t)(1 + record) * p->pool_P;
    *row_pitch_bytes = (size_t)R * 4;
    return CDFT_OK;
}

int cdft_moc_stage_close(cdft_moc_plan *p) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_moc_stage_close: NULL plan");
    if (p->sh || p->open_npt < 1) return fail(CDFT_ERR_STATE, "cdft_moc_stage_close: cdft_moc_stage_open has not succeeded");
    p->staged_npt = p->open_npt;
    p->open_npt = 0;
    return CDFT_OK;
}

// ------------------------------------------------------------------------------------------------ the stream route
// The window's slot 0 holds the raw metric e3v of its rows (as read from the file); in place it becomes the weight
// e1v * (e3v * vmask), the reference's products in the reference's order (src/cdfmoc.f90:581-593, :374).
__global__ void __launch_bounds__(256) moc_window_weight_kernel(float *__restrict__ win, int64_t R, const float *__restrict__ e1v,
                                                                const int8_t *__restrict__ vm, int npi, int npj, int64_t nrows) {
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int64_t lev = r / npj, jj = r - lev * npj;
        float *w = win + r * R;
        const float *e1 = e1v + jj * npi;
        const int8_t *m = vm ? vm + r * (int64_t)npi : nullptr;      // (the window's vmask is dense: rows npi bytes apart)
        for (int i = threadIdx.x; i < npi; i += 256) {
            const float e3 = m ? __fmul_rn(w[i], (float)m[i]) : w[i];
            w[i] = __fmul_rn(e1[i], e3);
        }
    }
}

int cdft_moc_stream_begin(cdft_moc_plan *p, int npt, int nlev, void *stream) {
    TraceRange trace_("cdft_moc_stream_begin");
    if (!p || npt < 1 || nlev < 1) return fail(CDFT_ERR_ARG, "cdft_moc_stream_begin: bad argument");
    if (p->sh) return fail(CDFT_ERR_STATE, "cdft_moc_stream_begin: single-device plans only");
    if (!p->basins_set) return fail(CDFT_ERR_STATE, "cdft_moc_stream_begin: cdft_moc_set_basins (or cdft_moc_set_mesh) has not succeeded");
    hipStream_t st = as_stream(stream);
    if (nlev > p->npk - 1) nlev = p->npk - 1;
    const int64_t P = ((int64_t)p->npi + 3) & ~(int64_t)3, rows = (int64_t)p->npj * nlev;
    const size_t bytes = (size_t)(rows * (npt + 1) * P) * 4 + 64;
    p->win_npt = 0;
    CDFT_TRY(p->win.reserve(bytes));
    CDFT_TRY(p->win_vm.reserve((size_t)rows * p->npi + 64));
    const size_t out_bytes = (size_t)p->nbasins * p->npj * p->npk * npt * sizeof(double);
    CDFT_TRY(p->st_out.reserve(out_bytes));
    // the cells between npi and P and the bytes behind the last row are read (and masked): never uninitialised memory
    CDFT_HIP(hipMemsetAsync(p->win.ptr, 0, bytes, st));
    CDFT_HIP(hipMemsetAsync(p->st_out.ptr, 0, out_bytes, st));
    p->win_npt = npt; p->win_nlev = nlev; p->win_P = P;
    return CDFT_OK;
}

int cdft_moc_stream_dest(cdft_moc_plan *p, int slot, int level, void **dst, size_t *row_pitch_bytes) {
    if (!p || !dst || !row_pitch_bytes) return fail(CDFT_ERR_ARG, "cdft_moc_stream_dest: NULL argument");
    if (p->win_npt < 1) return fail(CDFT_ERR_STATE, "cdft_moc_stream_dest: cdft_moc_stream_begin has not succeeded");
    if (slot < 0 || slot > p->win_npt || level < 0 || level >= p->win_nlev)
        return fail(CDFT_ERR_ARG, "cdft_moc_stream_dest: slot %d of 0..%d, level %d of a window of %d", slot, p->win_npt, level, p->win_nlev);
    const int64_t R = (int64_t)(p->win_npt + 1) * p->win_P;
    *dst = (float *)p->win.ptr + (int64_t)level * p->npj * R + (int64_t)slot * p->win_P;
    *row_pitch_bytes = (size_t)R * 4;
    return CDFT_OK;
}

int cdft_moc_stream_vmask_dest(cdft_moc_plan *p, int level, void **dst) {
    if (!p || !dst) return fail(CDFT_ERR_ARG, "cdft_moc_stream_vmask_dest: NULL argument");
    if (p->win_npt < 1) return fail(CDFT_ERR_STATE, "cdft_moc_stream_vmask_dest: cdft_moc_stream_begin has not succeeded");
    if (level < 0 || level >= p->win_nlev) return fail(CDFT_ERR_ARG, "cdft_moc_stream_vmask_dest: level %d of a window of %d", level, p->win_nlev);
    *dst = (int8_t *)p->win_vm.ptr + (int64_t)level * p->npj * p->npi;
    return CDFT_OK;
}

int cdft_moc_stream_compute(cdft_moc_plan *p, int k0, int nlev, int with_vmask, void *stream) {
    TraceRange trace_("cdft_moc_stream_compute");
    if (!p) return fail(CDFT_ERR_ARG, "cdft_moc_stream_compute: NULL plan");
    if (p->win_npt < 1) return fail(CDFT_ERR_STATE, "cdft_moc_stream_compute: cdft_moc_stream_begin has not succeeded");
    if (k0 < 0 || nlev < 1 || nlev > p->win_nlev || k0 + nlev > p->npk - 1)
        return fail(CDFT_ERR_ARG, "cdft_moc_stream_compute: levels %d .. %d of 0 .. %d, a window of %d", k0, k0 + nlev - 1, p->npk - 2, p->win_nlev);
    hipStream_t st = as_stream(stream);
    float *win = (float *)p->win.ptr;
    const int64_t R = (int64_t)(p->win_npt + 1) * p->win_P, rows = (int64_t)p->npj * nlev;
    const unsigned grid = (unsigned)(rows < 65536 ? rows : 65536);
    hipLaunchKernelGGL(moc_window_weight_kernel, dim3(grid), dim3(256), 0, st, win, R, (const float *)p->e1v2d.ptr,
                       with_vmask ? (const int8_t *)p->win_vm.ptr : nullptr, p->npi, p->npj, rows);
    CDFT_LAUNCH_CHECK();
    double *out = (double *)p->st_out.ptr + (int64_t)k0 * p->npj * p->nbasins;
    return zonal_launch(p, win, R, win + p->win_P, p->win_P, R, p->win_npt, out, st, true, nlev);
}

int cdft_moc_stream_end(cdft_moc_plan *p, double *dmoc, int dmoc_space, void *stream) {
    TraceRange trace_("cdft_moc_stream_end");
    if (!p || !dmoc) return fail(CDFT_ERR_ARG, "cdft_moc_stream_end: NULL argument");
    if (p->win_npt < 1) return fail(CDFT_ERR_STATE, "cdft_moc_stream_end: cdft_moc_stream_begin has not succeeded");
    hipStream_t st = as_stream(stream);
    const int npt = p->win_npt;
    const size_t out_bytes = (size_t)p->nbasins * p->npj * p->npk * npt * sizeof(double);
    double *d_out = (double *)p->st_out.ptr;
    const int nbj = p->nbasins * p->npj;
    const int64_t nthr = (int64_t)nbj * npt;
    hipLaunchKernelGGL(moc_cumsum_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, d_out, nbj, p->npk, npt);
    CDFT_LAUNCH_CHECK();
    if (dmoc_space == CDFT_MEM_DEVICE) {
        CDFT_HIP(hipMemcpyAsync(dmoc, d_out, out_bytes, hipMemcpyDeviceToDevice, st));
    } else if (dmoc_space == CDFT_MEM_HOST || dmoc_space == CDFT_MEM_PINNED) {
        CDFT_HIP(hipMemcpyAsync(dmoc, d_out, out_bytes, hipMemcpyDeviceToHost, st));
        if (dmoc_space == CDFT_MEM_HOST) CDFT_HIP(hipStreamSynchronize(st));
    } else {
        return fail(CDFT_ERR_ARG, "cdft_moc_stream_end: bad memory-space tag %d", dmoc_space);
    }
    p->win_npt = 0;                                           // (the window and the sums stay allocated for the next batch)
    return CDFT_OK;
}

int cdft_moc_release_staged(cdft_moc_plan *p) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_moc_release_staged: NULL plan");
    p->staged = nullptr; p->staged_npt = 0; p->open_npt = 0;
    if (p->sh) {
        int cur = 0, rc = CDFT_OK;
        (void)hipGetDevice(&cur);
        for (size_t b = 0; b < p->sh->part.size() && rc == CDFT_OK; ++b) {
            if (hipSetDevice(p->sh->dev[b]) != hipSuccess) rc = fail(CDFT_ERR_HIP, "cdft_moc_release_staged: hipSetDevice(%d) failed", p->sh->dev[b]);
            else rc = cdft_moc_release_staged((cdft_moc_plan *)p->sh->part[b]);
        }
        (void)hipSetDevice(cur);
        return rc;
    }
    p->pool.release(); p->win.release(); p->win_vm.release();
    p->pool_npt = 0; p->pool_P = 0; p->pool_w = false; p->win_npt = 0;
    return CDFT_OK;
}

int cdft_moc_stats(cdft_moc_plan *p, int64_t *zonal_launches, int64_t *records) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_moc_stats: NULL plan");
    int64_t nl = p->n_launch, nr = p->n_launch_rec;
    if (p->sh) {                       // a call on a sharded plan is one launch per block: block 0 speaks for the call
        cdft_moc_plan *q = (cdft_moc_plan *)p->sh->part[0];
        nl = q->n_launch; nr = q->n_launch_rec;
    }
    if (zonal_launches) *zonal_launches = nl;
    if (records) *records = nr;
    return CDFT_OK;
}

int cdft_moc_compute_staged(cdft_moc_plan *p, double *dmoc, int dmoc_space, void *stream) {
    TraceRange trace_("cdft_moc_compute_staged");
    if (!p || !dmoc) return fail(CDFT_ERR_ARG, "cdft_moc_compute_staged: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_moc_compute_staged: cdft_moc_set_mesh has not succeeded");
    if (p->staged_npt < 1) return fail(CDFT_ERR_STATE, "cdft_moc_compute_staged: cdft_moc_stage has not run (or a cdft_moc_compute since)");
    if (p->sh) return shard::moc_run(*p->sh, p->staged_npt, dmoc);
    if (p->staged) return moc_compute_on(p, p->staged, p->staged_stride, p->staged_npt, dmoc, dmoc_space, as_stream(stream));
    if (!p->pool_w) CDFT_TRY(pool_prepare(p, p->pool_npt, as_stream(stream)));   // a new mesh since the records were staged: its weights
    const float *pool = (const float *)p->pool.ptr;
    const int64_t R = (int64_t)(p->pool_npt + 1) * p->pool_P;
    return moc_compute_on(p, pool + p->pool_P, p->pool_P, p->staged_npt, dmoc, dmoc_space, as_stream(stream), R, pool, R);
}

// the staged pool computed a window of nlev levels per launch: the launches the stream route makes, on resident data
int cdft_moc_compute_staged_windows(cdft_moc_plan *p, int nlev, double *dmoc, int dmoc_space, void *stream) {
    TraceRange trace_("cdft_moc_compute_staged_windows");
    if (!p || !dmoc || nlev < 1) return fail(CDFT_ERR_ARG, "cdft_moc_compute_staged_windows: bad argument");
    if (p->sh) return fail(CDFT_ERR_STATE, "cdft_moc_compute_staged_windows: single-device plans only");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_moc_compute_staged_windows: cdft_moc_set_mesh has not succeeded");
    if (p->staged_npt < 1 || p->staged) return fail(CDFT_ERR_STATE, "cdft_moc_compute_staged_windows: no records staged in the pool");
    hipStream_t st = as_stream(stream);
    if (!p->pool_w) CDFT_TRY(pool_prepare(p, p->pool_npt, st));
    const int npt = p->staged_npt;
    const size_t out_bytes = (size_t)p->nbasins * p->npj * p->npk * npt * sizeof(double);
    double *d_out = dmoc;
    if (dmoc_space == CDFT_MEM_HOST || dmoc_space == CDFT_MEM_PINNED) {
        CDFT_TRY(p->st_out.reserve(out_bytes));
        d_out = (double *)p->st_out.ptr;
    } else if (dmoc_space != CDFT_MEM_DEVICE) {
        return fail(CDFT_ERR_ARG, "cdft_moc_compute_staged_windows: bad memory-space tag %d", dmoc_space);
    }
    const float *pool = (const float *)p->pool.ptr;
    const int64_t R = (int64_t)(p->pool_npt + 1) * p->pool_P;
    for (int k0 = 0; k0 < p->npk - 1; k0 += nlev) {
        const int n = std::min(nlev, p->npk - 1 - k0);
        const float *w = pool + (int64_t)k0 * p->npj * R;
        CDFT_TRY(zonal_launch(p, w, R, w + p->pool_P, p->pool_P, R, npt, d_out + (int64_t)k0 * p->npj * p->nbasins, st, true, n));
    }
    const int nbj = p->nbasins * p->npj;
    const int64_t nthr = (int64_t)nbj * npt;
    hipLaunchKernelGGL(moc_cumsum_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, d_out, nbj, p->npk, npt);
    CDFT_LAUNCH_CHECK();
    if (dmoc_space != CDFT_MEM_DEVICE) {
        CDFT_HIP(hipMemcpyAsync(dmoc, d_out, out_bytes, hipMemcpyDeviceToHost, st));
        if (dmoc_space == CDFT_MEM_HOST) CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_moc_compute_pitched(cdft_moc_plan *p, const float *zv, int64_t rec_stride, int64_t row_stride, int npt, double *dmoc,
                             int dmoc_space, void *stream) {
    TraceRange trace_("cdft_moc_compute_pitched");
    if (!p || !zv || !dmoc) return fail(CDFT_ERR_ARG, "cdft_moc_compute_pitched: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_moc_compute_pitched: cdft_moc_set_mesh has not succeeded");
    if (p->sh) return fail(CDFT_ERR_STATE, "cdft_moc_compute_pitched: single-device plans only (a sharded plan lays its blocks out itself: cdft_moc_stage)");
    if (npt < 1 || row_stride < p->npi || rec_stride < 1)
        return fail(CDFT_ERR_ARG, "cdft_moc_compute_pitched: npt=%d rec_stride=%lld row_stride=%lld (rows of %d cells)", npt,
                    (long long)rec_stride, (long long)row_stride, p->npi);
    // records and rows must not overlap: either the records are the outer dimension, or the rows are
    const int64_t rows = (int64_t)p->npj * p->npk;
    const bool rec_outer = rec_stride >= (rows - 1) * row_stride + p->npi;
    const bool row_outer = row_stride >= (int64_t)(npt - 1) * rec_stride + p->npi && rec_stride >= p->npi;
    if (!rec_outer && !row_outer)
        return fail(CDFT_ERR_ARG, "cdft_moc_compute_pitched: rec_stride=%lld and row_stride=%lld make rows of %d cells overlap",
                    (long long)rec_stride, (long long)row_stride, p->npi);
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, zv) != hipSuccess || at.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(CDFT_ERR_ARG, "cdft_moc_compute_pitched: zv must be device memory (host records go through cdft_moc_stage)");
    }
    return moc_compute_on(p, zv, rec_stride, npt, dmoc, dmoc_space, as_stream(stream), row_stride);
}

}  // extern "C"

// get_e3v (src/cdfmoc.f90:576-596): e3v * vmask, the mask converted as the reference's INTEGER(2) is
__global__ void __launch_bounds__(256) moc_mask_e3v_kernel(float *__restrict__ e3v, const int8_t *__restrict__ vmask, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        e3v[i] = __fmul_rn(e3v[i], (float)(int)vmask[i]);
}

int cdft_moc_mask_e3v(float *e3v, const int8_t *vmask, int64_t n, void *stream) {
    if (!e3v || !vmask || n < 0) return fail(CDFT_ERR_ARG, "cdft_moc_mask_e3v: NULL argument or negative size");
    CDFT_TRY(require_device());
    if (n == 0) return CDFT_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(moc_mask_e3v_kernel, dim3(grid), dim3(256), 0, as_stream(stream), e3v, vmask, n);
    CDFT_LAUNCH_CHECK();
    return CDFT_OK;
}

int cdft_moc_decomp_set_mesh(cdft_moc_plan *p, const float *e3v, const float *e1u, const float *zcoef,
                             const float *gdept, const uint8_t *tmask, const uint8_t *umask, int memspace,
                             void *stream) {
    if (!p || !e3v || !e1u || !zcoef || !gdept || !tmask || !umask)
        return fail(CDFT_ERR_ARG, "cdft_moc_decomp_set_mesh: NULL argument");
    if (p->sh) return fail(CDFT_ERR_STATE, "cdft_moc_decomp_set_mesh: -decomp runs on one device (its U-point stencil and depth "
                                           "means cross rows); create the plan with cdft_moc_create");
    if (p->npi < 3 || p->npj < 3)
        return fail(CDFT_ERR_ARG, "cdft_moc_decomp_set_mesh: the 4-point stencil needs npiglo, npjglo >= 3");
    hipStream_t st = as_stream(stream);
    const size_t nij = (size_t)p->npi * p->npj, n3 = nij * p->npk;
    auto keep = [&](DevBuf &b, const void *src, size_t bytes) -> int {   // the plan owns a copy
        CDFT_TRY(b.reserve(bytes));
        CDFT_HIP(hipMemcpyAsync(b.ptr, src, bytes, memspace == CDFT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        return CDFT_OK;
    };
    CDFT_TRY(keep(p->d_e3v, e3v, n3 * 4));
    CDFT_TRY(keep(p->d_e1u, e1u, nij * 4));
    CDFT_TRY(keep(p->d_zcoef, zcoef, nij * 4));
    CDFT_TRY(keep(p->d_tmask, tmask, n3));
    CDFT_TRY(keep(p->d_umask, umask, n3));
    p->gdept.assign(gdept, gdept + p->npk);                   // gdept is always a HOST array (npk values)
    p->pref3d_n = 0;                                            // (the reference depths on the device are formed again)
    CDFT_HIP(hipStreamSynchronize(st));
    p->dec_set = true;
    return CDFT_OK;
}

int cdft_moc_decomp(cdft_moc_plan *p, const float *zv, const float *ztemp, const float *zsal, int eos, int in_space,
                    double *dmoc, double *dmoc_sh, double *dmoc_bt, double *dmoc_ag, int out_space, void *stream) {
    TraceRange trace_("cdft_moc_decomp");
    TraceMute mute_;                                             // (the sigmai calls below are part of this one)
    if (!p || !zv || !ztemp || !zsal || !dmoc || !dmoc_sh || !dmoc_bt || !dmoc_ag)
        return fail(CDFT_ERR_ARG, "cdft_moc_decomp: NULL argument");
    if (!p->mesh_set || !p->dec_set)
        return fail(CDFT_ERR_STATE, "cdft_moc_decomp: cdft_moc_set_mesh and cdft_moc_decomp_set_mesh must have succeeded");
    if (eos != CDFT_EOS_80 && eos != CDFT_EOS_TEOS10) return fail(CDFT_ERR_ARG, "cdft_moc_decomp: eos must be CDFT_EOS_80 or CDFT_EOS_TEOS10");
    hipStream_t st = as_stream(stream);
    const int npi = p->npi, npj = p->npj, npk = p->npk, nb = p->nbasins;
    const int64_t nij = (int64_t)npi * npj, n3 = nij * npk;
    const size_t ob = (size_t)nb * npj * npk * 8;
    const void *dzv, *dt, *ds;
    if (p->staged) { p->staged = nullptr; p->staged_npt = 0; }   // (records kept as they came may live in st_zv, reused here)
    CDFT_TRY(to_device(zv, n3 * 4, in_space, p->st_zv, st, &dzv));
    CDFT_TRY(to_device(ztemp, n3 * 4, in_space, p->d_t, st, &dt));
    CDFT_TRY(to_device(zsal, n3 * 4, in_space, p->d_s, st, &ds));
    CDFT_TRY(p->d_zvgeo.reserve(n3 * 4));
    CDFT_TRY(p->d_sig8.reserve(nij * 8));
    CDFT_TRY(p->d_dvbt.reserve(nij * 8));
    CDFT_TRY(p->d_hdep.reserve(nij * 4));
    CDFT_TRY(p->d_dvgeo.reserve(nij * 8));
    CDFT_TRY(p->d_raw.reserve(3 * ob));                       // raw zonal sums: bt, sh, btw
    double *raw_bt = (double *)p->d_raw.ptr, *raw_sh = raw_bt + ob / 8, *raw_btw = raw_sh + ob / 8;
    double *o_tot = dmoc, *o_sh = dmoc_sh, *o_bt = dmoc_bt, *o_ag = dmoc_ag;
    if (out_space == CDFT_MEM_HOST) {
        CDFT_TRY(p->d_o4.reserve(4 * ob));
        o_tot = (double *)p->d_o4.ptr; o_sh = o_tot + ob / 8; o_bt = o_sh + ob / 8; o_ag = o_bt + ob / 8;
        CDFT_HIP(hipMemcpyAsync(o_sh, dmoc_sh, ob, hipMemcpyHostToDevice, st));   // IN/OUT
    } else if (out_space != CDFT_MEM_DEVICE) {
        return fail(CDFT_ERR_ARG, "cdft_moc_decomp: bad memory-space tag %d", out_space);
    }
    const float *e3v = (const float *)p->d_e3v.ptr, *fzv = (const float *)dzv;
    const uint8_t *tm = (const uint8_t *)p->d_tmask.ptr, *um = (const uint8_t *)p->d_umask.ptr;
    double *dvbt = (double *)p->d_dvbt.ptr, *dvgeo = (double *)p->d_dvgeo.ptr, *sig8 = (double *)p->d_sig8.ptr;   // (sig8: one level, or all of them below)
    float *hdep = (float *)p->d_hdep.ptr, *zvgeo = (float *)p->d_zvgeo.ptr;
    const dim3 g2((unsigned)((nij + 255) / 256)), b2(256), grow((unsigned)npj * (unsigned)npk);
    auto zonal_r8 = [&](const double *field, double *raw) -> int {
        // the class kernel of cdfzonalsum (a term added to ONE slot instead of nb select-adds and a five-sum wave reduction per row);
        // CDFT_DEC_ZONAL=select: the round-2 kernel (A/B and test hook)
        {
            const char *ez = getenv("CDFT_DEC_ZONAL");
            if (!(ez && ez[0] == 's') && npk >= 2 && zonal_class_fits_moc(npi, p->nclass))
                return zonal_class_moc_r8(p->w, field, p->cls, p->pat, p->nclass, npi, npj, npk - 1, nb, raw, st);
        }
        if (nb == 1)      hipLaunchKernelGGL((dec_zonal_r8_kernel<1>), grow, b2, 0, st, p->w, p->bits, field, npi, npj, npk, nb, raw, (int64_t)npi);
        else if (nb <= 5) hipLaunchKernelGGL((dec_zonal_r8_kernel<5>), grow, b2, 0, st, p->w, p->bits, field, npi, npj, npk, nb, raw, (int64_t)npi);
        else              hipLaunchKernelGGL((dec_zonal_r8_kernel<8>), grow, b2, 0, st, p->w, p->bits, field, npi, npj, npk, nb, raw, (int64_t)npi);
        CDFT_LAUNCH_CHECK();
        return CDFT_OK;
    };
    // 1) total MOC of the record (src/cdfmoc.f90:352-388)
    CDFT_TRY(cdft_moc_compute(p, fzv, n3, 1, CDFT_MEM_DEVICE, o_tot, CDFT_MEM_DEVICE, stream));
    // 2.1) barotropic part (:360-365, :397-419)
    hipLaunchKernelGGL(dec_vmean_kernel, g2, b2, 0, st, e3v, fzv, nij, npk, dvbt, hdep);
    CDFT_LAUNCH_CHECK();
    CDFT_TRY(zonal_r8(dvbt, raw_bt));
    // 2.2) geostrophic shear part (:435-478): levels from the bottom up
    CDFT_HIP(hipMemsetAsync(dvgeo, 0, nij * 8, st));
    CDFT_HIP(hipMemsetAsync(dvbt, 0, nij * 8, st));
    CDFT_HIP(hipMemsetAsync(zvgeo + (int64_t)(npk - 1) * nij, 0, nij * 4, st));
    if (npk >= 2 && !getenv("CDFT_MOC_DECOMP_LEVELS")) {       // (the hook keeps the level-by-level launches: tests compare the two)
        // sigmai of the levels 0 .. npk-2 in ONE launch (reference depth gdept(jk) at every point of level jk), then one kernel
        // that walks every column from the bottom up
        const int64_t nlow = (int64_t)(npk - 1) * nij;
        CDFT_TRY(p->d_sig8.reserve((size_t)nlow * 8));
        sig8 = (double *)p->d_sig8.ptr;
        if (!p->d_pref3d.ptr || p->pref3d_n != nlow) {
            CDFT_TRY(p->d_pref3d.reserve((size_t)nlow * 4 + (size_t)npk * 4));
            float *dg = (float *)p->d_pref3d.ptr + nlow;           // (gdept on the device, behind the field)
            CDFT_HIP(hipMemcpyAsync(dg, p->gdept.data(), (size_t)npk * 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(dec_pref_kernel, dim3((unsigned)((nij + 255) / 256), (unsigned)(npk - 1)), dim3(256), 0, st, dg, nij, (float *)p->d_pref3d.ptr);
            CDFT_LAUNCH_CHECK();
            CDFT_HIP(hipStreamSynchronize(st));                   // (gdept.data() is the plan's, but keep the copy simple)
            p->pref3d_n = nlow;
        }
        CDFT_TRY(cdft_eos_sigmai2d((const float *)dt, (const float *)ds, (const float *)p->d_pref3d.ptr, nlow, eos, 0, sig8,
                                   CDFT_MEM_DEVICE, CDFT_MEM_DEVICE, stream));
        const char *esh = getenv("CDFT_DEC_SHEAR");                  // A/B hook: "load" = every neighbour loaded (the round-4 kernel)
        if (esh && esh[0] == 'l')
            hipLaunchKernelGGL(dec_shear_all_kernel, dim3((unsigned)((npi + 255) / 256), (unsigned)npj), dim3(256), 0, st, sig8, tm, um,
                               (const float *)p->d_e1u.ptr, (const float *)p->d_zcoef.ptr, p->bits, e3v, fzv + (int64_t)(npk - 2) * nij, npi, npj,
                               npk, dvgeo, zvgeo, dvbt);
        else
            hipLaunchKernelGGL(dec_shear_share_kernel, dim3((unsigned)((npi + 4 * kShearCols - 1) / (4 * kShearCols)), (unsigned)npj), dim3(256), 0, st,
                               sig8, tm, um, (const float *)p->d_e1u.ptr, (const float *)p->d_zcoef.ptr, p->bits, e3v,
                               fzv + (int64_t)(npk - 2) * nij, npi, npj, npk, dvgeo, zvgeo, dvbt);
        CDFT_LAUNCH_CHECK();
    } else
    for (int jk = npk - 2; jk >= 0; --jk) {
        const int64_t o = (int64_t)jk * nij;
        CDFT_TRY(cdft_eos_sigmai((const float *)dt + o, (const float *)ds + o, nij, p->gdept[jk], eos, 0, sig8,
                                 CDFT_MEM_DEVICE, CDFT_MEM_DEVICE, stream));
        hipLaunchKernelGGL(dec_shear_kernel, g2, b2, 0, st, sig8, tm + o, um + o, (const float *)p->d_e1u.ptr,
                           (const float *)p->d_zcoef.ptr, p->bits, e3v + o, fzv + (int64_t)(npk - 2) * nij, npi, npj,
                           dvgeo, zvgeo + o, dvbt);
        CDFT_LAUNCH_CHECK();
    }
    CDFT_HIP(hipMemsetAsync(raw_sh, 0, ob, st));              // level npk of the raw sums stays 0
    CDFT_TRY(zonal_launch(p, p->w, (int64_t)npi, zvgeo, n3, (int64_t)npi, 1, raw_sh, st, false));
    // 2.2.1) its pseudo-barotropic part (:480-502)
    hipLaunchKernelGGL(dec_divide_kernel, g2, b2, 0, st, dvbt, hdep, nij);
    CDFT_LAUNCH_CHECK();
    CDFT_TRY(zonal_r8(dvbt, raw_btw));
    // :416-419, :505-516
    const int nbj = nb * npj;
    hipLaunchKernelGGL(dec_finish_kernel, dim3((unsigned)((nbj + 255) / 256)), b2, 0, st, o_tot, raw_bt, raw_sh, raw_btw,
                       nbj, npk, o_sh, o_bt, o_ag);
    CDFT_LAUNCH_CHECK();
    if (out_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(dmoc, o_tot, ob, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipMemcpyAsync(dmoc_sh, o_sh, ob, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipMemcpyAsync(dmoc_bt, o_bt, ob, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipMemcpyAsync(dmoc_ag, o_ag, ob, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_moc_profile_enable(cdft_moc_plan *p, int on) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_moc_profile_enable: NULL plan");
    if (p->sh) {
        for (void *q : p->sh->part) CDFT_TRY(cdft_moc_profile_enable((cdft_moc_plan *)q, on));
        return CDFT_OK;
    }
    p->profiling = on != 0;
    return CDFT_OK;
}

int cdft_moc_profile_read(cdft_moc_plan *p, double *zonal_ms_total, int *launches) {
    if (!p || !zonal_ms_total || !launches) return fail(CDFT_ERR_ARG, "cdft_moc_profile_read: NULL argument");
    if (p->sh) {                                         // per launch the slowest block: that is what the call waited for
        std::vector<float> each(65536);
        int n = 0;
        CDFT_TRY(cdft_moc_profile_read_each(p, each.data(), (int)each.size(), &n));
        double tot = 0.0;
        for (int i = 0; i < n && i < (int)each.size(); ++i) tot += each[i];
        *zonal_ms_total = tot;
        *launches = n;
        return CDFT_OK;
    }
    double tot = 0.0;
    for (size_t i = 0; i + 1 < p->ev_used; i += 2) {
        float ms = 0.f;
        CDFT_HIP(hipEventSynchronize(p->ev[i + 1]));
        CDFT_HIP(hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]));
        tot += ms;
    }
    *zonal_ms_total = tot;
    *launches = (int)(p->ev_used / 2);
    p->ev_used = 0;
    return CDFT_OK;
}

int cdft_moc_profile_read_each(cdft_moc_plan *p, float *ms_each, int capacity, int *launches) {
    if (!p || !ms_each || !launches || capacity < 0)
        return fail(CDFT_ERR_ARG, "cdft_moc_profile_read_each: bad argument");
    if (p->sh) {                                         // per launch the slowest block (cdft_moc_profile_read_block: one block)
        const int nblk = p->sh->nblk();
        std::vector<float> one((size_t)(capacity > 0 ? capacity : 1));
        int cur = 0;
        (void)hipGetDevice(&cur);
        int rc = CDFT_OK, nmin = -1;
        for (int b = 0; b < nblk && rc == CDFT_OK; ++b) {
            int n = 0;
            if (hipSetDevice(p->sh->dev[b]) != hipSuccess) { rc = fail(CDFT_ERR_HIP, "cdft_moc_profile_read_each: hipSetDevice(%d) failed", p->sh->dev[b]); break; }
            rc = cdft_moc_profile_read_each((cdft_moc_plan *)p->sh->part[b], one.data(), capacity, &n);
            for (int i = 0; rc == CDFT_OK && i < n && i < capacity; ++i) ms_each[i] = (b == 0 || one[i] > ms_each[i]) ? one[i] : ms_each[i];
            if (nmin < 0 || n < nmin) nmin = n;
        }
        (void)hipSetDevice(cur);
        CDFT_TRY(rc);
        *launches = nmin < 0 ? 0 : nmin;
        return CDFT_OK;
    }
    const int n = (int)(p->ev_used / 2);
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        CDFT_HIP(hipEventSynchronize(p->ev[2 * i + 1]));
        CDFT_HIP(hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]));
        if (i < capacity) ms_each[i] = ms;
    }
    *launches = n;
    p->ev_used = 0;
    return CDFT_OK;
}

int cdft_moc_profile_read_block(cdft_moc_plan *p, int block, float *ms_each, int capacity, int *launches) {
    if (!p || !ms_each || !launches || capacity < 0) return fail(CDFT_ERR_ARG, "cdft_moc_profile_read_block: bad argument");
    if (!p->sh) {
        if (block != 0) return fail(CDFT_ERR_ARG, "cdft_moc_profile_read_block: block %d of a plan with one block", block);
        return cdft_moc_profile_read_each(p, ms_each, capacity, launches);
    }
    if (block < 0 || block >= p->sh->nblk()) return fail(CDFT_ERR_ARG, "cdft_moc_profile_read_block: block %d of %d", block, p->sh->nblk());
    int cur = 0;
    (void)hipGetDevice(&cur);
    CDFT_HIP(hipSetDevice(p->sh->dev[block]));
    int rc = cdft_moc_profile_read_each((cdft_moc_plan *)p->sh->part[block], ms_each, capacity, launches);
    (void)hipSetDevice(cur);
    return rc;
}

int cdft_moc_block_info(cdft_moc_plan *p, int block, int *device, int *j0, int *j1) {
    if (!p || !device || !j0 || !j1) return fail(CDFT_ERR_ARG, "cdft_moc_block_info: NULL argument");
    if (!p->sh) {
        if (block != 0) return fail(CDFT_ERR_ARG, "cdft_moc_block_info: block %d of a plan with one block", block);
        *device = -1; *j0 = 0; *j1 = p->npj;
        (void)hipGetDevice(device);
        return CDFT_OK;
    }
    if (block < 0 || block >= p->sh->nblk()) return fail(CDFT_ERR_ARG, "cdft_moc_block_info: block %d of %d", block, p->sh->nblk());
    *device = p->sh->dev[block]; *j0 = p->sh->j0[block]; *j1 = p->sh->j1[block];
    return CDFT_OK;
}

int cdft_moc_destroy(cdft_moc_plan *p) {
    if (!p) return CDFT_OK;
    if (p->sh) {
        (void)shard::for_blocks(*p->sh, [&](int b) { return cdft_moc_destroy((cdft_moc_plan *)p->sh->part[b]); });
        shard::destroy(p->sh);
        delete p;
        return CDFT_OK;
    }
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    if (p->w) (void)hipFree(p->w);
    if (p->bits) (void)hipFree(p->bits);
    if (p->cls) (void)hipFree(p->cls);
    if (p->flag) (void)hipFree(p->flag);
    p->st_e1v.release(); p->st_e3v.release(); p->st_mask.release(); p->st_vm.release(); p->st_zv.release(); p->st_out.release(); p->seg_out.release();
    p->pool.release(); p->win.release(); p->win_vm.release(); p->e1v2d.release();
    DevBuf *dec[] = {&p->d_e3v, &p->d_e1u, &p->d_zcoef, &p->d_tmask, &p->d_umask, &p->d_t, &p->d_s, &p->d_zvgeo, &p->d_sig8,
                     &p->d_dvbt, &p->d_hdep, &p->d_dvgeo, &p->d_raw, &p->d_o4, &p->d_pref3d};
    for (DevBuf *b : dec) b->release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"
