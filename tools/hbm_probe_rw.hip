// What a 2-read / 1-write level walk reaches on one MI355X: the traffic pattern of cdfcurl (u, v in, curl out, every
// workgroup owning a chunk of the horizontal plane and walking the levels) without its stencil -- the practical
// ceiling that kernel is measured against.  Build: hipcc -O3 --offload-arch=gfx950 tools/hbm_probe_rw.hip -o /tmp/hbm_probe_rw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: plain loads, nontemporal stores; 1: nontemporal loads and stores; 2: all plain
template <int VPT, int MODE, bool SYNC>
__global__ void __launch_bounds__(256) walk_kernel(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c, size_t stride,
                                                   int K, size_t nrows) {
    const size_t chunk = (size_t)VPT * blockDim.x;
    const size_t row = blockIdx.x;
    if (row >= nrows) return;
    const size_t p0 = row * chunk + threadIdx.x;
    f4 va[VPT], vb[VPT], na[VPT], nb[VPT];
    auto ld = [&](const f4 *q) { return MODE == 1 ? __builtin_nontemporal_load(q) : *q; };
#pragma unroll
    for (int r = 0; r < VPT; ++r) { va[r] = ld(a + p0 + (size_t)r * blockDim.x); vb[r] = ld(b + p0 + (size_t)r * blockDim.x); }
    for (int t = 0; t < K; ++t) {
        if (t + 1 < K) {
#pragma unroll
            for (int r = 0; r < VPT; ++r) {
                na[r] = ld(a + (size_t)(t + 1) * stride + p0 + (size_t)r * blockDim.x);
                nb[r] = ld(b + (size_t)(t + 1) * stride + p0 + (size_t)r * blockDim.x);
            }
        }
        if (SYNC) __syncthreads();
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const f4 o = va[r] * vb[r] + va[r];
            f4 *q = c + (size_t)t * stride + p0 + (size_t)r * blockDim.x;
            if (MODE == 2) *q = o; else __builtin_nontemporal_store(o, q);
        }
#pragma unroll
        for (int r = 0; r < VPT; ++r) { va[r] = na[r]; vb[r] = nb[r]; }
    }
}

// flat streaming triad: grid-stride over contiguous chunks (no level structure)
template <int VPT>
__global__ void __launch_bounds__(256) triad_kernel(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c, size_t nvec) {
    const size_t chunk = (size_t)VPT * blockDim.x;
    for (size_t base = (size_t)blockIdx.x * chunk; base < nvec; base += (size_t)gridDim.x * chunk) {
        f4 va[VPT], vb[VPT];
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const size_t p = base + (size_t)r * blockDim.x + threadIdx.x;
            if (p < nvec) { va[r] = __builtin_nontemporal_load(a + p); vb[r] = __builtin_nontemporal_load(b + p); }
        }
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const size_t p = base + (size_t)r * blockDim.x + threadIdx.x;
            if (p < nvec) __builtin_nontemporal_store(va[r] * vb[r] + va[r], c + p);
        }
    }
}

// flat sweep in address order, one 4 KB chunk per workgroup, where every level ALSO reads NM 2-D planes (cdfcurl's
// metrics) at the same horizontal position: plain loads, so they can stay in L2 / Infinity Cache from one level to the next
// (NM planes of 62 MB against 256 MB of Infinity Cache); HALO: one more read of `a` a row (1081 vectors) ahead
template <int NM, int HALO>
__global__ void __launch_bounds__(256) flat_metric_kernel(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c,
                                                          const f4 *__restrict__ m, size_t stride, size_t nvec) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= nvec) return;
    const size_t pp = p % stride;
    f4 va = __builtin_nontemporal_load(a + p), vb = __builtin_nontemporal_load(b + p);
    f4 acc = va * vb;
    if (HALO) acc += a[min(p + 1081, nvec - 1)];
#pragma unroll
    for (int q = 0; q < NM; ++q) acc += m[(size_t)q * stride + pp] * va;
    __builtin_nontemporal_store(acc, c + p);
}

// short-lived workgroups, one (tile, level) item each, handed out TILE-major: the K levels of a tile are consecutive
// blocks (XCD = 0: blocks b..b+K-1 on whatever XCD they land; XCD = 1: all K levels of a tile on one XCD, tiles dealt
// round-robin to the 8 XCDs) -- what a cdfcurl with per-tile metrics in L2 would look like
template <int VPT, int XCD>
__global__ void __launch_bounds__(256) item_kernel(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c, size_t stride,
                                                   int K, size_t ntiles) {
    size_t tile, lev;
    if (XCD) {
        const size_t x = blockIdx.x & 7, s = blockIdx.x >> 3;
        tile = (s / K) * 8 + x; lev = s % K;
    } else {
        tile = blockIdx.x / K; lev = blockIdx.x % K;
    }
    if (tile >= ntiles) return;
    const size_t p0 = lev * stride + tile * VPT * blockDim.x + threadIdx.x;
    f4 va[VPT], vb[VPT];
#pragma unroll
    for (int r = 0; r < VPT; ++r) { va[r] = __builtin_nontemporal_load(a + p0 + (size_t)r * blockDim.x); vb[r] = __builtin_nontemporal_load(b + p0 + (size_t)r * blockDim.x); }
#pragma unroll
    for (int r = 0; r < VPT; ++r) __builtin_nontemporal_store(va[r] * vb[r] + va[r], c + p0 + (size_t)r * blockDim.x);
}

// level walk with WG threads: one workgroup = `threads` lanes, barrier per level
template <int VPT>
__global__ void __launch_bounds__(1024) walk_big_kernel(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c, size_t stride,
                                                        int K, size_t nrows) {
    const size_t chunk = (size_t)VPT * blockDim.x;
    const size_t row = blockIdx.x;
    if (row >= nrows) return;
    const size_t p0 = row * chunk + threadIdx.x;
    f4 va[VPT], vb[VPT], na[VPT], nb[VPT];
#pragma unroll
    for (int r = 0; r < VPT; ++r) { va[r] = __builtin_nontemporal_load(a + p0 + (size_t)r * blockDim.x); vb[r] = __builtin_nontemporal_load(b + p0 + (size_t)r * blockDim.x); }
    for (int t = 0; t < K; ++t) {
        if (t + 1 < K) {
#pragma unroll
            for (int r = 0; r < VPT; ++r) {
                na[r] = __builtin_nontemporal_load(a + (size_t)(t + 1) * stride + p0 + (size_t)r * blockDim.x);
                nb[r] = __builtin_nontemporal_load(b + (size_t)(t + 1) * stride + p0 + (size_t)r * blockDim.x);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < VPT; ++r) __builtin_nontemporal_store(va[r] * vb[r] + va[r], c + (size_t)t * stride + p0 + (size_t)r * blockDim.x);
#pragma unroll
        for (int r = 0; r < VPT; ++r) { va[r] = na[r]; vb[r] = nb[r]; }
    }
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

int main() {
    const size_t level = 15585132, K = 75;                 // eORCA12: cells per level, levels
    const size_t stride = (level + 3) / 4;                 // vectors per level
    const size_t bytes = stride * 16 * K;
    f4 *a, *b, *c;
    CK(hipMalloc(&a, bytes + 65536)); CK(hipMalloc(&b, bytes + 65536)); CK(hipMalloc(&c, bytes + 65536));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    auto run = [&](auto vpt_tag, auto mode_tag, auto sync_tag) {
        constexpr int VPT = decltype(vpt_tag)::value, MODE = decltype(mode_tag)::value;
        constexpr bool SYNC = decltype(sync_tag)::value;
        const size_t chunk = (size_t)VPT * 256, nrows = stride / chunk;
        float m = time_ms([&] { hipLaunchKernelGGL((walk_kernel<VPT, MODE, SYNC>), dim3((unsigned)nrows), dim3(256), 0, 0, a, b, c, stride, (int)K, nrows); }, 5);
        printf("{\"pattern\": \"level walk 2 reads + 1 write\", \"chunk_KiB\": %zu, \"mode\": \"%s\", \"barrier_per_level\": %d, \"ms\": %.3f, \"GBps\": %.0f}\n",
               chunk * 16 / 1024, MODE == 0 ? "plain loads, nt stores" : MODE == 1 ? "nt loads, nt stores" : "plain", SYNC ? 1 : 0, m,
               3.0 * K * nrows * chunk * 16 / m * 1e-6);
        fflush(stdout);
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
    using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>; using M2 = std::integral_constant<int, 2>;
    using T = std::true_type; using F = std::false_type;
    run(I1{}, M0{}, F{}); run(I1{}, M1{}, F{}); run(I1{}, M2{}, F{}); run(I1{}, M1{}, T{});
    run(I2{}, M0{}, F{}); run(I2{}, M1{}, F{}); run(I2{}, M1{}, T{});
    run(I4{}, M0{}, F{}); run(I4{}, M1{}, F{}); run(I4{}, M1{}, T{});
    const size_t nvec = stride * K;
    for (int g : {2048, 8192, 0}) {
        auto tri = [&](auto vpt_tag) {
            constexpr int VPT = decltype(vpt_tag)::value;
            const size_t chunk = (size_t)VPT * 256;
            const unsigned grid = g ? g : (unsigned)((nvec + chunk - 1) / chunk);
            float m = time_ms([&] { hipLaunchKernelGGL(triad_kernel<VPT>, dim3(grid), dim3(256), 0, 0, a, b, c, nvec); }, 5);
            printf("{\"pattern\": \"flat triad 2 reads + 1 write\", \"vpt\": %d, \"grid\": %u, \"ms\": %.3f, \"GBps\": %.0f}\n", VPT, grid, m, 3.0 * nvec * 16 / m * 1e-6);
            fflush(stdout);
        };
        tri(I1{}); tri(I2{}); tri(I4{});
    }
    {
        auto item = [&](auto vpt_tag, auto x_tag) {
            constexpr int VPT = decltype(vpt_tag)::value, X = decltype(x_tag)::value;
            const size_t chunk = (size_t)VPT * 256, ntiles = stride / chunk;
            const size_t nblk = ((ntiles + 7) / 8) * 8 * K;
            float m = time_ms([&] { hipLaunchKernelGGL((item_kernel<VPT, X>), dim3((unsigned)nblk), dim3(256), 0, 0, a, b, c, stride, (int)K, ntiles); }, 5);
            printf("{\"pattern\": \"one (tile, level) item per workgroup, tile-major\", \"xcd_aware\": %d, \"chunk_KiB\": %zu, \"ms\": %.3f, \"GBps\": %.0f}\n", X,
                   chunk * 16 / 1024, m, 3.0 * K * ntiles * chunk * 16 / m * 1e-6);
            fflush(stdout);
        };
        item(I1{}, M0{}); item(I1{}, M1{}); item(I2{}, M0{}); item(I2{}, M1{}); item(I4{}, M0{}); item(I4{}, M1{});
    }
    {
        f4 *m;
        CK(hipMalloc(&m, stride * 16 * 4 + 65536)); CK(hipMemset(m, 3, stride * 16 * 4));
        auto fm = [&](auto nm_tag, auto halo_tag) {
            constexpr int NM = decltype(nm_tag)::value, HALO = decltype(halo_tag)::value;
            float t = time_ms([&] { hipLaunchKernelGGL((flat_metric_kernel<NM, HALO>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, a, b, c, m, stride, nvec); }, 5);
            printf("{\"pattern\": \"flat sweep, level-major, 2 reads + 1 write + metric planes per level\", \"metric_planes\": %d, \"halo_row\": %d, \"ms\": %.3f, \"GBps_of_the_3_streams\": %.0f}\n",
                   NM, HALO, t, 3.0 * nvec * 16 / t * 1e-6);
            fflush(stdout);
        };
        using I0 = std::integral_constant<int, 0>; using I3 = std::integral_constant<int, 3>;
        fm(I0{}, I0{}); fm(I1{}, I0{}); fm(I2{}, I0{}); fm(I3{}, I0{}); fm(I4{}, I0{}); fm(I3{}, I1{}); fm(I4{}, I1{});
        CK(hipFree(m));
    }
    for (int threads : {512, 1024}) {
        auto big = [&](auto vpt_tag) {
            constexpr int VPT = decltype(vpt_tag)::value;
            const size_t chunk = (size_t)VPT * threads, nrows = stride / chunk;
            float m = time_ms([&] { hipLaunchKernelGGL(walk_big_kernel<VPT>, dim3((unsigned)nrows), dim3(threads), 0, 0, a, b, c, stride, (int)K, nrows); }, 5);
            printf("{\"pattern\": \"level walk 2 reads + 1 write, barrier\", \"threads\": %d, \"chunk_KiB\": %zu, \"ms\": %.3f, \"GBps\": %.0f}\n", threads,
                   chunk * 16 / 1024, m, 3.0 * K * nrows * chunk * 16 / m * 1e-6);
            fflush(stdout);
        };
        big(I1{}); big(I2{});
    }
    CK(hipGetLastError());
    return 0;
}
