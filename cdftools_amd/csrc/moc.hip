// moc.hip -- cdfmoc plan: mesh preparation, zonal-kernel dispatch, bottom-to-surface cumsum.
// C ABI: include/cdftools_hip.h ("cdfmoc").  Reference loops: src/cdfmoc.f90:325-336 (masks),
// :343-348,:576-596 (e3v), :355-379 (zonal sum), :382-388 (vertical cumulated sum).
#include <cstdlib>
#include <vector>
#include "moc_zonal.hpp"

using namespace cdft;
using namespace cdft::moc;

struct cdft_moc_plan {
    int npi = 0, npj = 0, npk = 0, nbasins = 0;
    float   *w    = nullptr;   // (npi,npj,npk) r4 weight e1v*e3v
    uint8_t *bits = nullptr;   // (npi,npj) basin bits
    uint8_t *cls  = nullptr;   // (npi,npj) class of the cell's bit pattern (class mode), 0xFF = none
    int     *flag = nullptr;   // device: [0] error flag (mask value check), [1..256] pattern present
    uint8_t pat[kMaxClass] = {0};
    int     nclass = 0;        // distinct non-empty patterns, 0 when there are more than kMaxClass
    bool mesh_set = false;
    int  tune_waves = 0, tune_nv = 0, tune_mode = 0;
    DevBuf st_e1v, st_e3v, st_mask, st_zv, st_out;   // staging for host arrays
    bool profiling = false;
    std::vector<hipEvent_t> ev;                      // (start, stop) pairs, one per profiled launch
    size_t ev_used = 0;
};

namespace {

// w(i,j,k) = e1v(i,j) * e3v(i,j,k), rounded to r4 like the reference's first product
// (src/cdfmoc.f90:374: e1v(ji,jj)*e3v(ji,jj,jk) is evaluated first, in r4).
__global__ void moc_weight_kernel(const float *__restrict__ e1v, const float *__restrict__ e3v,
                                  float *__restrict__ w, int64_t nij, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride)
        w[e] = __fmul_rn(e1v[e % nij], e3v[e]);
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
        if (v && !flag[1 + v]) flag[1 + v] = 1;          // benign race: every writer stores 1
    }
}

// cls(i,j) = class index of the cell's basin-bit pattern (lut), 0xFF for a cell in no basin
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
    for (int jk = npk - 2; jk >= 0; --jk) {
        const double v = base[(int64_t)jk * nbj];
        below = below + v / 1.0e6;
        base[(int64_t)jk * nbj] = below;
    }
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// pick waves per row and float4 vectors per lane for a row of npi cells
void choose_shape(const cdft_moc_plan *p, bool aligned, int *nwaves, int *nv, int *mode) {
    const int nvec = (p->npi + 6) / 4;             // worst-case aligned vectors covering a row
    int md = p->tune_mode ? p->tune_mode : ((p->nbasins > 1 && p->nclass > 0) ? kModeClass : kModeSelect);
    if (md == kModeClass && (!aligned || p->nbasins == 1 || p->nclass == 0 || p->nclass > kMaxClass))
        md = kModeSelect;
    int k, v;
    if (!aligned) {
        md = kModeSelect;
        v  = 4;
        k  = ceil_div(nvec, kWave * v);
    } else if (p->tune_waves > 0 || p->tune_nv > 0) {
        if (p->tune_waves > 0 && p->tune_nv > 0) { k = p->tune_waves; v = p->tune_nv; }
        else if (p->tune_waves > 0) { k = p->tune_waves; v = ceil_div(nvec, kWave * k); }
        else { v = p->tune_nv; k = ceil_div(nvec, kWave * v); }
    } else {
        int vmax = (md == kModeFma) ? 4 : (md == kModeClass ? 9 : 8);   // measured best at npiglo = 4322
        k = ceil_div(nvec, kWave * vmax);
        if (k > kSmallWaves) {          // long rows: 1024-thread workgroups, fewer registers
            vmax = (md == kModeFma) ? 2 : kBigNV;
            k = ceil_div(nvec, kWave * vmax);
        }
        v = ceil_div(nvec, kWave * k);
    }
    *nwaves = k; *nv = v; *mode = md;
}

}  // namespace

extern "C" {

int cdft_moc_create(cdft_moc_plan **plan, int npiglo, int npjglo, int npk, int nbasins) {
    if (!plan) return fail(CDFT_ERR_ARG, "cdft_moc_create: NULL plan pointer");
    *plan = nullptr;
    if (npiglo < 1 || npjglo < 1 || npk < 2)
        return fail(CDFT_ERR_ARG, "cdft_moc_create: bad sizes npiglo=%d npjglo=%d npk=%d (npk >= 2)",
                    npiglo, npjglo, npk);
    if (nbasins < 1 || nbasins > 8)
        return fail(CDFT_ERR_ARG, "cdft_moc_create: nbasins=%d outside 1..8", nbasins);
    if ((npiglo + 6) / 4 > kWave * 4 * kMaxWaves)   // the unaligned path holds 4 vectors per lane
        return fail(CDFT_ERR_ARG, "cdft_moc_create: npiglo=%d exceeds the supported row length %d",
                    npiglo, 4 * kWave * 4 * kMaxWaves - 6);
    if ((int64_t)npjglo * (npk - 1) > 0x7fffffffLL)
        return fail(CDFT_ERR_ARG, "cdft_moc_create: npjglo*(npk-1) exceeds the grid limit");
    CDFT_TRY(require_device());
    cdft_moc_plan *p = new cdft_moc_plan;
    p->npi = npiglo; p->npj = npjglo; p->npk = npk; p->nbasins = nbasins;
    const size_t n3 = (size_t)npiglo * npjglo * npk;
    if (hipMalloc((void **)&p->w, n3 * sizeof(float) + 64) != hipSuccess ||
        hipMalloc((void **)&p->bits, (size_t)npiglo * npjglo + 64) != hipSuccess ||
        hipMalloc((void **)&p->cls, (size_t)npiglo * npjglo + 64 + 256) != hipSuccess ||
        hipMalloc((void **)&p->flag, 257 * sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        cdft_moc_destroy(p);
        return fail(CDFT_ERR_ALLOC, "cdft_moc_create: device allocation of %zu bytes failed",
                    n3 * sizeof(float));
    }
    *plan = p;
    return CDFT_OK;
}

int cdft_moc_set_mesh(cdft_moc_plan *p, const float *e1v, const float *e3v, const int16_t *ibmask,
                      int memspace, void *stream) {
    if (!p || !e1v || !e3v || !ibmask) return fail(CDFT_ERR_ARG, "cdft_moc_set_mesh: NULL argument");
    hipStream_t st = as_stream(stream);
    const int64_t nij = (int64_t)p->npi * p->npj, n3 = nij * p->npk;
    const void *d_e1v, *d_e3v, *d_msk;
    CDFT_TRY(to_device(e1v, nij * sizeof(float), memspace, p->st_e1v, st, &d_e1v));
    CDFT_TRY(to_device(e3v, n3 * sizeof(float), memspace, p->st_e3v, st, &d_e3v));
    CDFT_TRY(to_device(ibmask, nij * p->nbasins * sizeof(int16_t), memspace, p->st_mask, st, &d_msk));
    CDFT_HIP(hipMemsetAsync(p->flag, 0, 257 * sizeof(int), st));
    CDFT_HIP(hipMemsetAsync(p->w + n3, 0, 64, st));
    const int threads = 256;
    int blocks = (int)((n3 + threads - 1) / threads);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(moc_weight_kernel, dim3(blocks), dim3(threads), 0, st, (const float *)d_e1v,
                       (const float *)d_e3v, p->w, nij, n3);
    CDFT_LAUNCH_CHECK();
    blocks = (int)((nij + threads - 1) / threads);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(moc_bits_kernel, dim3(blocks), dim3(threads), 0, st, (const int16_t *)d_msk,
                       p->bits, p->nbasins, nij, p->flag);
    CDFT_LAUNCH_CHECK();
    int flags[257];
    CDFT_HIP(hipMemcpyAsync(flags, p->flag, sizeof(flags), hipMemcpyDeviceToHost, st));
    CDFT_HIP(hipStreamSynchronize(st));
    const int flag = flags[0];
    // class mode: the distinct non-empty bit patterns, in increasing order
    uint8_t lut[256];
    p->nclass = 0;
    for (int v = 0; v < 256; ++v) lut[v] = 0xFF;
    for (int v = 1; v < 256; ++v)
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
    // the staged copies of host arrays are no longer needed (e3v is as large as the weight)
    p->st_e1v.release(); p->st_e3v.release(); p->st_mask.release();
    if (flag) {
        p->mesh_set = false;
        return fail(CDFT_ERR_ARG, "cdft_moc_set_mesh: ibmask holds values other than 0/1");
    }
    p->mesh_set = true;
    return CDFT_OK;
}

int cdft_moc_set_tuning(cdft_moc_plan *p, int waves_per_row, int vec_per_lane, int mode) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_moc_set_tuning: NULL plan");
    if (waves_per_row < 0 || waves_per_row > kMaxWaves || vec_per_lane < 0 || vec_per_lane > kMaxNV ||
        mode < 0 || mode > 3)
        return fail(CDFT_ERR_ARG, "cdft_moc_set_tuning: out of range (%d,%d,%d)", waves_per_row,
                    vec_per_lane, mode);
    if (waves_per_row > 0 && vec_per_lane > 0 &&
        waves_per_row * vec_per_lane * kWave < (p->npi + 6) / 4)
        return fail(CDFT_ERR_ARG, "cdft_moc_set_tuning: %d waves x %d vectors do not cover a row of %d",
                    waves_per_row, vec_per_lane, p->npi);
    p->tune_waves = waves_per_row; p->tune_nv = vec_per_lane; p->tune_mode = mode;
    return CDFT_OK;
}

int cdft_moc_describe(cdft_moc_plan *p, char *buf, size_t buflen) {
    if (!p || !buf) return fail(CDFT_ERR_ARG, "cdft_moc_describe: NULL argument");
    int k, v, md;
    choose_shape(p, true, &k, &v, &md);
    const int nb = p->nbasins == 1 ? 1 : (p->nbasins <= 5 ? 5 : 8);
    if (md == kModeClass)
        snprintf(buf, buflen, "moc_zonal_class_kernel<NV=%d,KC=%d> waves_per_row=%d basin_patterns=%d", v, kMaxClass, k, p->nclass);
    else
        snprintf(buf, buflen, "moc_zonal_kernel<NV=%d,NB=%d,ALIGNED=1,MODE=%s> waves_per_row=%d", v, nb,
                 md == kModeFma ? "fma" : "select", k);
    return CDFT_OK;
}

int cdft_moc_compute(cdft_moc_plan *p, const float *zv, int64_t rec_stride, int npt, int zv_space,
                     double *dmoc, int dmoc_space, void *stream) {
    if (!p || !zv || !dmoc) return fail(CDFT_ERR_ARG, "cdft_moc_compute: NULL argument");
    if (!p->mesh_set) return fail(CDFT_ERR_STATE, "cdft_moc_compute: cdft_moc_set_mesh has not succeeded");
    const int64_t n3 = (int64_t)p->npi * p->npj * p->npk;
    if (npt < 1 || rec_stride < n3)
        return fail(CDFT_ERR_ARG, "cdft_moc_compute: npt=%d rec_stride=%lld (need npt>=1, stride>=%lld)",
                    npt, (long long)rec_stride, (long long)n3);
    hipStream_t st = as_stream(stream);
    const void *d_zv;
    CDFT_TRY(to_device(zv, ((size_t)(npt - 1) * rec_stride + n3) * sizeof(float), zv_space, p->st_zv,
                       st, &d_zv));
    const size_t out_bytes = (size_t)p->nbasins * p->npj * p->npk * npt * sizeof(double);
    double *d_out = dmoc;
    if (dmoc_space == CDFT_MEM_HOST) {
        CDFT_TRY(p->st_out.reserve(out_bytes));
        d_out = (double *)p->st_out.ptr;
    } else if (dmoc_space != CDFT_MEM_DEVICE) {
        return fail(CDFT_ERR_ARG, "cdft_moc_compute: bad memory-space tag %d", dmoc_space);
    }

    ZonalArgs a;
    a.w = p->w; a.bits = p->bits; a.cls = p->cls; a.nclass = p->nclass;
    for (int c = 0; c < kMaxClass; ++c) a.pat[c] = p->pat[c];
    a.zv = (const float *)d_zv; a.rec_stride = rec_stride; a.npt = npt;
    a.out = d_out; a.npi = p->npi; a.npj = p->npj; a.npk = p->npk; a.nbasins = p->nbasins;
    const bool aligned = ((uintptr_t)d_zv % 16 == 0) && (rec_stride % 4 == 0);
    int k, v, md;
    choose_shape(p, aligned, &k, &v, &md);
    if (k > kMaxWaves || v > kMaxNV)
        return fail(CDFT_ERR_ARG, "cdft_moc_compute: row of %d cells needs %d waves x %d vectors", p->npi, k, v);
    hipEvent_t ev1 = nullptr;
    if (p->profiling) {
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

    const int nbj = p->nbasins * p->npj;
    const int64_t nthr = (int64_t)nbj * npt;
    hipLaunchKernelGGL(moc_cumsum_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, d_out,
                       nbj, p->npk, npt);
    CDFT_LAUNCH_CHECK();
    if (dmoc_space == CDFT_MEM_HOST) {
        CDFT_HIP(hipMemcpyAsync(dmoc, d_out, out_bytes, hipMemcpyDeviceToHost, st));
        CDFT_HIP(hipStreamSynchronize(st));
    }
    return CDFT_OK;
}

int cdft_moc_profile_enable(cdft_moc_plan *p, int on) {
    if (!p) return fail(CDFT_ERR_ARG, "cdft_moc_profile_enable: NULL plan");
    p->profiling = on != 0;
    return CDFT_OK;
}

int cdft_moc_profile_read(cdft_moc_plan *p, double *zonal_ms_total, int *launches) {
    if (!p || !zonal_ms_total || !launches) return fail(CDFT_ERR_ARG, "cdft_moc_profile_read: NULL argument");
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

int cdft_moc_destroy(cdft_moc_plan *p) {
    if (!p) return CDFT_OK;
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    if (p->w) (void)hipFree(p->w);
    if (p->bits) (void)hipFree(p->bits);
    if (p->cls) (void)hipFree(p->cls);
    if (p->flag) (void)hipFree(p->flag);
    p->st_e1v.release(); p->st_e3v.release(); p->st_mask.release(); p->st_zv.release(); p->st_out.release();
    delete p;
    return CDFT_OK;
}

}  // extern "C"
