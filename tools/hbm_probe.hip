// HBM ceiling probe for one MI355X: what a pure streaming read (the shape of every kernel of this
// path), a copy and a 2-read/1-write stream reach, for several vectors-per-lane and grid sizes.
// The fractions in DESIGN.md are quoted against the 8 TB/s spec figure; this gives the practical
// denominator next to it.   Build: hipcc -O3 --offload-arch=gfx950 tools/hbm_probe.hip -o /tmp/hbm_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// each workgroup streams a contiguous chunk of VPT*blockDim vectors per iteration, grid-stride
template <int VPT>
__global__ void __launch_bounds__(256) read_kernel(const f4 *__restrict__ a, size_t nvec, float *out) {
    float acc = 0.f;
    const size_t chunk = (size_t)VPT * blockDim.x;
    for (size_t base = (size_t)blockIdx.x * chunk; base < nvec; base += (size_t)gridDim.x * chunk) {
        f4 v[VPT];
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const size_t p = base + (size_t)r * blockDim.x + threadIdx.x;
            v[r] = p < nvec ? __builtin_nontemporal_load(a + p) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int r = 0; r < VPT; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
    }
    if (acc == 12345.678f) out[0] = acc;   // never true: keeps the loads alive
}

template <int VPT>
__global__ void __launch_bounds__(256) copy_kernel(const f4 *__restrict__ a, f4 *__restrict__ b, size_t nvec) {
    const size_t chunk = (size_t)VPT * blockDim.x;
    for (size_t base = (size_t)blockIdx.x * chunk; base < nvec; base += (size_t)gridDim.x * chunk) {
        f4 v[VPT];
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const size_t p = base + (size_t)r * blockDim.x + threadIdx.x;
            if (p < nvec) v[r] = __builtin_nontemporal_load(a + p);
        }
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const size_t p = base + (size_t)r * blockDim.x + threadIdx.x;
            if (p < nvec) __builtin_nontemporal_store(v[r], b + p);
        }
    }
}

template <int VPT>
__global__ void __launch_bounds__(256) read2_kernel(const f4 *__restrict__ a, const f4 *__restrict__ b, size_t nvec, float *out) {
    float acc = 0.f;
    const size_t chunk = (size_t)VPT * blockDim.x;
    for (size_t base = (size_t)blockIdx.x * chunk; base < nvec; base += (size_t)gridDim.x * chunk) {
        f4 v[VPT], w[VPT];
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const size_t p = base + (size_t)r * blockDim.x + threadIdx.x;
            v[r] = p < nvec ? a[p] : f4{0, 0, 0, 0};
            w[r] = p < nvec ? b[p] : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int r = 0; r < VPT; ++r) acc += v[r].x * w[r].x + v[r].y * w[r].y + v[r].z * w[r].z + v[r].w * w[r].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// The cdfmoc access pattern: every workgroup reads its chunk (one grid row) from each of K streams
// (records) that lie `stride` vectors apart, one stream after the other with the next one
// prefetched.  grid == nrows: one workgroup per row, dispatched as slots free up (the workgroups
// drift apart, so all K regions are read at once at scattered rows).  grid < nrows: persistent
// workgroups that walk the rows with a grid stride (they start each row together).
template <int VPT, bool NT = false>
__global__ void __launch_bounds__(256) multi_kernel(const f4 *__restrict__ a, size_t stride, int K, size_t nrows, float *out) {
    float acc = 0.f;
    const size_t chunk = (size_t)VPT * blockDim.x;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const f4 *p = a + row * chunk + threadIdx.x;
        f4 v[VPT], n[VPT];
#pragma unroll
        for (int r = 0; r < VPT; ++r) v[r] = NT ? __builtin_nontemporal_load(p + (size_t)r * blockDim.x) : p[(size_t)r * blockDim.x];
        for (int t = 0; t < K; ++t) {
            if (t + 1 < K) {
#pragma unroll
                for (int r = 0; r < VPT; ++r) n[r] = NT ? __builtin_nontemporal_load(p + (size_t)(t + 1) * stride + (size_t)r * blockDim.x) : p[(size_t)(t + 1) * stride + (size_t)r * blockDim.x];
            }
#pragma unroll
            for (int r = 0; r < VPT; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
#pragma unroll
            for (int r = 0; r < VPT; ++r) v[r] = n[r];
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <class F>
static float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? (size_t)atol(argv[1]) : 8;
    const size_t bytes = gib << 30, nvec = bytes / 16;
    f4 *a, *b;
    float *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    printf("{\"probe\": \"hbm\", \"buffer_GiB\": %zu}\n", gib);
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32, 0};
    for (int gi = 0; gi < 5; ++gi) {
        auto run = [&](auto vpt_tag) {
            constexpr int VPT = decltype(vpt_tag)::value;
            const size_t chunk = (size_t)VPT * 256;
            const unsigned grid = grids[gi] ? grids[gi] : (unsigned)((nvec + chunk - 1) / chunk);
            float r = time_ms([&] { hipLaunchKernelGGL(read_kernel<VPT>, dim3(grid), dim3(256), 0, 0, a, nvec, out); }, 10);
            float c = time_ms([&] { hipLaunchKernelGGL(copy_kernel<VPT>, dim3(grid), dim3(256), 0, 0, a, b, nvec); }, 10);
            float d = time_ms([&] { hipLaunchKernelGGL(read2_kernel<VPT>, dim3(grid), dim3(256), 0, 0, a, b, nvec, out); }, 10);
            printf("{\"grid\": %u, \"vpt\": %d, \"read_GBps\": %.0f, \"copy_GBps\": %.0f, \"read2_GBps\": %.0f}\n", grid, VPT,
                   bytes / r * 1e-6, 2.0 * bytes / c * 1e-6, 2.0 * bytes / d * 1e-6);
            fflush(stdout);
        };
        run(std::integral_constant<int, 1>{});
        run(std::integral_constant<int, 2>{});
        run(std::integral_constant<int, 4>{});
        run(std::integral_constant<int, 8>{});
    }
    // multi-stream pattern: K streams of `per` bytes inside buffer a (needs K*per <= bytes)
    for (int K : {1, 2, 4, 13}) {
        const size_t per = (bytes / 13) & ~(size_t)0xFFFFF, stride = per / 16;
        auto run = [&](auto vpt_tag, int threads) {
            constexpr int VPT = decltype(vpt_tag)::value;
            const size_t chunk = (size_t)VPT * threads, nrows = stride / chunk;
            for (unsigned grid : {0u, 256u * 2, 256u * 4, 256u * 8}) {
                const unsigned g = grid ? grid : (unsigned)nrows;
                float m = time_ms([&] { hipLaunchKernelGGL(multi_kernel<VPT>, dim3(g), dim3(threads), 0, 0, a, stride, K, nrows, out); }, 5);
                printf("{\"streams\": %d, \"chunk_KiB\": %zu, \"threads\": %d, \"vpt\": %d, \"grid\": \"%s\", \"nwg\": %u, \"GBps\": %.0f}\n", K,
                       chunk * 16 / 1024, threads, VPT, grid ? "persistent" : "per-row", g, (double)K * nrows * chunk * 16 / m * 1e-6);
                fflush(stdout);
            }
        };
        run(std::integral_constant<int, 6>{}, 192);     // the cdfmoc shape: 3 waves x 6 vectors = 18 KiB
        run(std::integral_constant<int, 8>{}, 256);     // 32 KiB
        run(std::integral_constant<int, 16>{}, 256);    // 64 KiB
    }
    // the level walk of cdfheatc / cdfmhst / cdfpsi / cdfcurl: 75 (x fields) slabs of one eORCA12 level
    // (62 MB apart), every workgroup reads its chunk of each slab in turn
    if (bytes >= (size_t)225 * 15585132 * 4 + (64 << 20)) {
        const size_t stride = 15585132 / 4;                 // one level, in vectors
        for (int K : {75, 150, 225}) {
            auto run = [&](auto vpt_tag, int threads) {
                constexpr int VPT = decltype(vpt_tag)::value;
                const size_t chunk = (size_t)VPT * threads, nrows = stride / chunk;
                float m0 = time_ms([&] { hipLaunchKernelGGL((multi_kernel<VPT, false>), dim3((unsigned)nrows), dim3(threads), 0, 0, a, stride, K, nrows, out); }, 5);
                float m1 = time_ms([&] { hipLaunchKernelGGL((multi_kernel<VPT, true>), dim3((unsigned)nrows), dim3(threads), 0, 0, a, stride, K, nrows, out); }, 5);
                printf("{\"pattern\": \"level walk\", \"slabs\": %d, \"chunk_KiB\": %zu, \"nwg\": %zu, \"plain_GBps\": %.0f, \"nt_GBps\": %.0f}\n", K,
                       chunk * 16 / 1024, nrows, (double)K * nrows * chunk * 16 / m0 * 1e-6, (double)K * nrows * chunk * 16 / m1 * 1e-6);
                fflush(stdout);
            };
            run(std::integral_constant<int, 1>{}, 256);      // 4 KiB  (cdfcurl / cdfvT tile)
            run(std::integral_constant<int, 4>{}, 256);      // 16 KiB (vertical integrals, cdfheatc)
            run(std::integral_constant<int, 8>{}, 256);      // 32 KiB
            run(std::integral_constant<int, 16>{}, 256);     // 64 KiB
        }
    }
    CK(hipGetLastError());
    return 0;
}
