// What the memory system gives the shape of the level-walking kernels (vint_kernel, mean_level_kernel, heatc_level_kernel):
// one workgroup owns 256 * VPT float4 vectors of the horizontal plane and walks npk levels of NF r4 fields (plus, with
// MASK, one byte field), D levels requested ahead of the one being consumed; no arithmetic beyond keeping the loads alive.
// Build: hipcc -O3 --offload-arch=gfx950 tools/level_walk_probe.hip -o /tmp/level_walk_probe ; run: /tmp/level_walk_probe [npi npj npk]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));

struct Args { const float *f[3]; const uint8_t *m; int64_t nij; int npk; float *out; };

template <int NF, int VPT, bool MASK> struct Lev { f4 x[NF][VPT]; uint32_t m[VPT]; };

template <int NF, int VPT, int D, bool MASK, bool NT, bool ALWAYS = false>
__global__ void __launch_bounds__(256) walk_kernel(Args a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    int64_t p[VPT];
    const int64_t nv = a.nij / 4;
#pragma unroll
    for (int v = 0; v < VPT; ++v) { const int64_t q = (wave * VPT + v) * 64 + lane; p[v] = q < nv ? q * 4 : 0; }
    Lev<NF, VPT, MASK> L[D + 1];
    auto fetch = [&](Lev<NF, VPT, MASK> &l, int k) {
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int64_t e = (int64_t)k * a.nij + p[v];
#pragma unroll
            for (int f = 0; f < NF; ++f)
                l.x[f][v] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(a.f[f] + e)) : *reinterpret_cast<const f4 *>(a.f[f] + e);
            if (MASK) {
                typedef uint32_t u32u __attribute__((aligned(1)));
                l.m[v] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32u *>(a.m + e)) : *reinterpret_cast<const u32u *>(a.m + e);
            }
        }
    };
    float acc = 0.f;
    auto use = [&](const Lev<NF, VPT, MASK> &l) {
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
#pragma unroll
            for (int f = 0; f < NF; ++f) acc += l.x[f][v].x + l.x[f][v].y + l.x[f][v].z + l.x[f][v].w;
            if (MASK) acc += (float)l.m[v];
        }
    };
    // the loop runs over groups of D+1 levels so that every buffer has a fixed name (registers, not scratch)
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(L[d], d < a.npk ? d : a.npk - 1);
    for (int k0 = 0; k0 < a.npk; k0 += D + 1) {
#pragma unroll
        for (int d = 0; d <= D; ++d) {
            const int k = k0 + d;
            if (ALWAYS || k < a.npk) {
                // ALWAYS: the request is made on every way through the loop (past the last level: that level again), so that the
                // compiler's waits can count D levels of younger loads; otherwise they count none beyond the level itself
                if (ALWAYS) fetch(L[(d + D) % (D + 1)], k + D < a.npk ? k + D : a.npk - 1);
                else if (k + D < a.npk) fetch(L[(d + D) % (D + 1)], k + D);
                use(L[d]);
            }
        }
    }
    if (acc == 12345.678f) a.out[0] = acc;
}

template <int NF, int VPT, int D, bool MASK, bool NT, bool ALWAYS = false>
void run(const Args &a, int reps) {
    const int64_t nv = a.nij / 4, nthr = (nv + VPT - 1) / VPT;
    const dim3 grid((unsigned)((nthr + 255) / 256)), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walk_kernel<NF, VPT, D, MASK, NT, ALWAYS>), grid, block, 0, 0, a);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0.f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((walk_kernel<NF, VPT, D, MASK, NT, ALWAYS>), grid, block, 0, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double bytes = (double)a.nij * a.npk * (4.0 * NF + (MASK ? 1.0 : 0.0));
    printf("{\"fields\": %d, \"byte_mask\": %d, \"vectors_per_lane\": %d, \"levels_ahead\": %d, \"nontemporal\": %d, \"unconditional_requests\": %d, \"workgroups\": %u, "
           "\"ms_avg\": %.4f, \"ms_min\": %.4f, \"TBps_avg\": %.3f, \"TBps_best\": %.3f}\n",
           NF, (int)MASK, VPT, D, (int)NT, (int)ALWAYS, grid.x, sum / reps, best, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int npi = argc > 3 ? atoi(argv[1]) : 1442, npj = argc > 3 ? atoi(argv[2]) : 1207, npk = argc > 3 ? atoi(argv[3]) : 75;
    Args a;
    a.nij = (int64_t)npi * npj; a.npk = npk;
    const size_t n3 = (size_t)a.nij * npk;
    for (int f = 0; f < 3; ++f) { CK(hipMalloc((void **)&a.f[f], n3 * 4 + 64)); CK(hipMemset((void *)a.f[f], 0, n3 * 4 + 64)); }
    CK(hipMalloc((void **)&a.m, n3 + 64)); CK(hipMemset((void *)a.m, 1, n3 + 64));
    CK(hipMalloc((void **)&a.out, 64));
    printf("{\"grid\": [%d, %d, %d]}\n", npi, npj, npk);
    const int R = 20;
    // cdfpsi: 2 fields; cdfmhst -vt: 3; cdfheatc / cdfmean: 2 + byte mask
    run<2, 4, 0, false, true>(a, R); run<2, 4, 1, false, true>(a, R); run<2, 4, 2, false, true>(a, R); run<2, 4, 3, false, true>(a, R);
    run<2, 8, 0, false, true>(a, R); run<2, 8, 1, false, true>(a, R); run<2, 2, 1, false, true>(a, R); run<2, 2, 3, false, true>(a, R);
    run<3, 4, 0, false, true>(a, R); run<3, 4, 1, false, true>(a, R); run<3, 4, 2, false, true>(a, R);
    run<3, 2, 1, false, true>(a, R); run<3, 2, 2, false, true>(a, R); run<3, 2, 3, false, true>(a, R); run<3, 6, 1, false, true>(a, R);
    run<2, 4, 0, true, true>(a, R); run<2, 4, 1, true, true>(a, R); run<2, 4, 2, true, true>(a, R);
    run<2, 2, 0, true, true>(a, R); run<2, 2, 1, true, true>(a, R); run<2, 2, 3, true, true>(a, R); run<2, 8, 0, true, true>(a, R);
    run<3, 4, 1, false, false>(a, R); run<2, 4, 1, true, false>(a, R);
    // the same with requests the compiler can count (npk must be a multiple of D + 1: 75 = 3 * 25 = 5 * 15)
    run<2, 4, 2, false, true, true>(a, R); run<2, 4, 4, false, true, true>(a, R); run<3, 4, 2, false, true, true>(a, R); run<3, 2, 4, false, true, true>(a, R);
    run<2, 4, 2, true, true, true>(a, R); run<2, 2, 4, true, true, true>(a, R); run<2, 2, 2, true, true, true>(a, R); run<3, 2, 2, false, true, true>(a, R);
    return 0;
}
