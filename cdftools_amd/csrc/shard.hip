// shard.hip -- row-block sharded plans over several GPUs of one node, driven by one process (see shard.hpp).
// Host code only: every kernel launch goes through the ordinary single-device entry points of the C ABI.
#include "shard.hpp"
#include "rccl_dl.hpp"
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <string>
#include <cstdlib>
#include <mutex>
#include <thread>

using namespace cdft;

// ------------------------------------------------------------------------------------------ RCCL, looked up at run time
namespace cdft {
namespace rccl {

const Api *api() {
    static Api a;
    static int state = 0;                                // 0 not tried, 1 ready, -1 failed
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (state == 1) return &a;
    if (state == -1) { set_error("RCCL is not available on this machine (librccl.so.1 or one of its entry points was not found)"); return nullptr; }
    void *h = nullptr;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    bool ok = h != nullptr;
    auto take = [&](const char *name) -> void * {
        void *f = ok ? dlsym(h, name) : nullptr;
        if (!f) ok = false;
        return f;
    };
    a.CommInitAll = (int (*)(comm_t *, int, const int *))take("ncclCommInitAll");
    a.CommDestroy = (int (*)(comm_t))take("ncclCommDestroy");
    a.CommAbort = (int (*)(comm_t))take("ncclCommAbort");
    a.AllGather = (int (*)(const void *, void *, size_t, int, comm_t, hipStream_t))take("ncclAllGather");
    a.Broadcast = (int (*)(const void *, void *, size_t, int, int, comm_t, hipStream_t))take("ncclBroadcast");
    a.Send = (int (*)(const void *, size_t, int, int, comm_t, hipStream_t))take("ncclSend");
    a.Recv = (int (*)(void *, size_t, int, int, comm_t, hipStream_t))take("ncclRecv");
    a.GroupStart = (int (*)())take("ncclGroupStart");
    a.GroupEnd = (int (*)())take("ncclGroupEnd");
    a.GetErrorString = (const char *(*)(int))take("ncclGetErrorString");
    state = ok ? 1 : -1;
    if (!ok) { set_error("RCCL is not available on this machine (%s)", h ? "an entry point is missing" : dlerror()); return nullptr; }
    return &a;
}

}  // namespace rccl
}  // namespace cdft

namespace {
std::atomic<int> g_collective{-1};                       // -1: not asked yet (CDFT_SHARD_COLLECTIVE decides)
int collective_mode() {
    int m = g_collective.load();
    if (m >= 0) return m;
    const char *e = getenv("CDFT_SHARD_COLLECTIVE");
    m = (e && (!strcmp(e, "rccl") || !strcmp(e, "RCCL") || !strcmp(e, "nccl"))) ? CDFT_COLLECTIVE_RCCL : CDFT_COLLECTIVE_PEER;
    g_collective.store(m);
    return m;
}
}  // namespace

extern "C" int cdft_shard_set_collective(int mode) {
    if (mode != CDFT_COLLECTIVE_PEER && mode != CDFT_COLLECTIVE_RCCL) return fail(CDFT_ERR_ARG, "cdft_shard_set_collective: mode %d", mode);
    g_collective.store(mode);
    return CDFT_OK;
}
extern "C" int cdft_shard_get_collective(int *mode) {
    if (!mode) return fail(CDFT_ERR_ARG, "cdft_shard_get_collective: NULL argument");
    *mode = collective_mode();
    return CDFT_OK;
}

namespace cdft {
namespace shard {

int from_env(int *ndev, int *devices, int capacity) {
    if (!ndev || !devices || capacity < 1) return fail(CDFT_ERR_ARG, "cdft_shard_from_env: bad argument");
    int n = 0;
    const char *map = getenv("CDFT_GPU_MAP");
    const char *cnt = getenv("CDFT_GPUS");
    if (map && *map) {
        const char *p = map;
        while (*p) {
            char *end = nullptr;
            long v = strtol(p, &end, 10);
            if (end == p || v < 0) return fail(CDFT_ERR_ARG, "CDFT_GPU_MAP=\"%s\": expected a comma-separated list of device numbers", map);
            if (n >= capacity) return fail(CDFT_ERR_ARG, "CDFT_GPU_MAP lists more than %d blocks", capacity);
            devices[n++] = (int)v;
            p = end;
            while (*p == ',' || *p == ' ') ++p;
        }
    } else if (cnt && *cnt) {
        char *end = nullptr;
        long v = strtol(cnt, &end, 10);
        if (end == cnt || v < 1 || v > capacity) return fail(CDFT_ERR_ARG, "CDFT_GPUS=\"%s\": expected 1..%d", cnt, capacity);
        for (n = 0; n < (int)v; ++n) devices[n] = n;
    }
    if (n == 0) { n = 1; devices[0] = -1; }          // -1: whatever device is current (the unsharded default)
    if (n > 1 || devices[0] >= 0) {
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess) { (void)hipGetLastError(); have = 0; }
        for (int b = 0; b < n; ++b)
            if (devices[b] >= have)
                return fail(have ? CDFT_ERR_ARG : CDFT_ERR_NODEVICE, "block %d asks for HIP device %d, the node has %d", b, devices[b], have);
    }
    *ndev = n;
    return CDFT_OK;
}

void row_block(int npj, int n, int b, int *j0, int *j1) {
    const int base = npj / n, extra = npj % n;
    *j0 = b * base + (b < extra ? b : extra);
    *j1 = *j0 + base + (b < extra ? 1 : 0);
}

int make(ShardSet **out, int npi, int npj, int npk, int nb, int ndev, const int *devices, int halo_north, int nbuf) {
    *out = nullptr;
    if (ndev < 1 || !devices) return fail(CDFT_ERR_ARG, "sharded plan: no devices");
    if (ndev > npj) return fail(CDFT_ERR_ARG, "sharded plan: %d blocks for %d rows", ndev, npj);
    CDFT_TRY(require_device());
    int have = 0;
    CDFT_HIP(hipGetDeviceCount(&have));
    for (int b = 0; b < ndev; ++b)
        if (devices[b] >= have) return fail(CDFT_ERR_ARG, "sharded plan: block %d asks for HIP device %d, the node has %d", b, devices[b], have);
    ShardSet *sh = new ShardSet;
    sh->npi = npi; sh->npj = npj; sh->npk = npk; sh->nb = nb;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int b = 0; b < ndev; ++b) {
        int j0, j1;
        row_block(npj, ndev, b, &j0, &j1);
        sh->dev.push_back(devices[b] < 0 ? cur : devices[b]);
        sh->j0.push_back(j0);
        sh->j1.push_back(j1);
        sh->r1.push_back((halo_north && j1 < npj) ? j1 + 1 : j1);
        sh->part.push_back(nullptr);
        sh->st.push_back(nullptr);
        sh->buf.emplace_back((size_t)nbuf);
    }
    for (int b = 0; b < ndev; ++b) {
        hipError_t e = hipSetDevice(sh->dev[b]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&sh->st[b], hipStreamNonBlocking);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipSetDevice(cur);
            destroy(sh);
            return fail(CDFT_ERR_HIP, "sharded plan: stream on device %d -> %s", devices[b], hipGetErrorString(e));
        }
    }
    // Blocks on distinct devices exchange rows of resident arrays and cdfpsi's carry directly: let every device of the set,
    // and the caller's current device (where a resident full-domain array lives), map the others' memory where the node
    // allows it (xGMI / PCIe peer-to-peer).  Where it does not, the runtime stages the copies; "already enabled" is fine.
    {
        std::vector<int> ds(sh->dev);
        ds.push_back(cur);
        for (int a : ds)
            for (int b : ds) {
                int can = 0;
                if (a == b || hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
                if (hipSetDevice(a) == hipSuccess) (void)hipDeviceEnablePeerAccess(b, 0);
                (void)hipGetLastError();
            }
    }
    (void)hipSetDevice(cur);
    if (collective_mode() == CDFT_COLLECTIVE_RCCL) {
        for (int a = 0; a < ndev; ++a)
            for (int b = a + 1; b < ndev; ++b)
                if (sh->dev[a] == sh->dev[b]) {
                    destroy(sh);
                    return fail(CDFT_ERR_ARG, "sharded plan: blocks %d and %d are both on device %d -- RCCL takes one rank per device "
                                "(CDFT_SHARD_COLLECTIVE=peer, the default, does not mind)", a, b, devices[a]);
                }
        const rccl::Api *api = rccl::api();
        if (!api) { destroy(sh); return CDFT_ERR_HIP; }
        std::vector<rccl::comm_t> comms((size_t)ndev, nullptr);
        int r = api->CommInitAll(comms.data(), ndev, sh->dev.data());
        (void)hipSetDevice(cur);
        if (r != 0) { destroy(sh); return fail(CDFT_ERR_HIP, "sharded plan: ncclCommInitAll over %d devices -> %s", ndev, api->GetErrorString(r)); }
        sh->comm.assign(comms.begin(), comms.end());
    }
    *out = sh;
    return CDFT_OK;
}

void destroy(ShardSet *sh) {
    if (!sh) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (sh->rccl() && rccl::api())
        for (void *c : sh->comm) if (c) (void)rccl::api()->CommDestroy(c);
    sh->comm.clear();
    for (int b = 0; b < sh->nblk(); ++b) {
        (void)hipSetDevice(sh->dev[b]);
        for (DevBuf &d : sh->buf[b]) d.release();
        if (sh->st[b]) (void)hipStreamDestroy(sh->st[b]);
    }
    (void)hipSetDevice(cur);
    (void)hipGetLastError();            // nothing sticky may outlive a plan (CDFT_LAUNCH_CHECK reads the last error)
    delete sh;
}

// early_fail(b): called (on block b's thread) when the block fails before fn(b) could run -- for_blocks_collective's hook
static int for_blocks_impl(ShardSet &sh, const std::function<int(int)> &fn, const std::function<void(int)> *early_fail) {
    const int n = sh.nblk();
    std::vector<int> rc(n, CDFT_OK);
    std::vector<std::string> msg(n);
    auto body = [&](int b) {
        TraceMute mute_;                                         // (the caller's range times the whole call: common.hpp)
        if (hipSetDevice(sh.dev[b]) != hipSuccess) {
            (void)hipGetLastError();
            rc[b] = CDFT_ERR_HIP;
            msg[b] = "hipSetDevice failed";
            if (early_fail) (*early_fail)(b);
            return;
        }
        rc[b] = fn(b);
        if (rc[b] != CDFT_OK) msg[b] = cdft_last_error();       // the message is thread-local: carry it over
    };
    if (n == 1) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        body(0);
        (void)hipSetDevice(cur);
    } else {
        std::vector<std::thread> th;
        for (int b = 0; b < n; ++b) th.emplace_back(body, b);
        for (auto &t : th) t.join();
    }
    for (int b = 0; b < n; ++b)
        if (rc[b] != CDFT_OK) return fail(rc[b], "row block %d (device %d): %s", b, sh.dev[b], msg[b].c_str());
    return CDFT_OK;
}

int for_blocks(ShardSet &sh, const std::function<int(int)> &fn) { return for_blocks_impl(sh, fn, nullptr); }

// The collectives of an operation, issued from the blocks' threads: every block must reach its call, or the others wait for
// the missing rank for ever.  So (1) the fallible per-block work -- reserves, copies, kernels -- runs in a for_blocks pass of
// its own whose status is checked BEFORE this one starts, and (2) if a block fails here all the same -- its hipSetDevice, the
// enqueue itself (`enqueue`: the block's ONE RCCL call), or what follows it on the stream (`finish`) -- every communicator
// is aborted: the blocks already waiting on their streams come back with an error instead of hanging, the set is left
// without communicators and the next sharded RCCL call reports that.
// The abort is made by the first failing block, and only once EVERY block is past its enqueue point (has returned from its
// RCCL call, or has seen the `failed` flag and skipped it): no thread can then hand an aborted -- freed -- communicator to
// RCCL.  A block that fails early raises the flag first, so the blocks not yet at their call skip it; a block already inside
// its call returns from it (an enqueue does not wait for the other ranks' kernels).  Should a call not return all the same,
// the abort goes ahead after kAbortWaitMs: ncclCommAbort is what RCCL offers against a blocked call.
static int for_blocks_collective(ShardSet &sh, const std::function<int(int)> &enqueue, const std::function<int(int)> &finish) {
    constexpr int kAbortWaitMs = 10000;
    const int n = sh.nblk();
    std::atomic<int> failed{0}, passed{0}, aborted{0};
    auto abort_all = [&]() {
        if (failed.exchange(1)) return;                          // (another block is already doing it)
        for (int waited = 0; passed.load() < n && waited < kAbortWaitMs; ++waited) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (rccl::api())
            for (void *c : sh.comm) if (c) (void)rccl::api()->CommAbort(c);
        aborted.store(1);
    };
    const std::function<void(int)> early = [&](int) { passed.fetch_add(1); abort_all(); };
    const int rc = for_blocks_impl(sh, [&](int b) {
        if (failed.load()) {                                     // another block has failed: this one never issues its call
            passed.fetch_add(1);
            return fail(CDFT_ERR_HIP, "collective not issued: another row block failed first");
        }
        // CDFT_SHARD_TEST_FAIL=enqueue|finish (test hook, tests/test_sharded_plans.py): block 0 fails there
        const char *hook = getenv("CDFT_SHARD_TEST_FAIL");
        int r = (b == 0 && hook && hook[0] == 'e') ? fail(CDFT_ERR_HIP, "CDFT_SHARD_TEST_FAIL: enqueue of block 0") : enqueue(b);
        passed.fetch_add(1);
        if (r == CDFT_OK) r = finish(b);
        if (r == CDFT_OK && b == 0 && hook && hook[0] == 'f') r = fail(CDFT_ERR_HIP, "CDFT_SHARD_TEST_FAIL: after the collective of block 0");
        if (r != CDFT_OK) {
            const std::string keep = cdft_last_error();          // (abort_all may wait; keep this block's own message)
            abort_all();
            set_error("%s", keep.c_str());
        }
        return r;
    }, &early);
    if (failed.load()) {
        while (!aborted.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));   // (the aborting thread has been joined: set already)
        sh.comm.assign(sh.comm.size(), nullptr);
    }
    return rc;
}
static int comms_alive(const ShardSet &sh, const char *who) {
    for (void *c : sh.comm)
        if (!c) return fail(CDFT_ERR_STATE, "%s: the plan's RCCL communicators were aborted by an earlier failure; make a new plan", who);
    return CDFT_OK;
}

static int rows_copy(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipStream_t st) {
    if (width == 0 || height == 0) return CDFT_OK;
    if (dpitch == width && spitch == width) {
        CDFT_HIP(hipMemcpyAsync(dst, src, width * height, hipMemcpyDefault, st));
    } else {
        CDFT_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDefault, st));
    }
    return CDFT_OK;
}

int rows_in(void *blk, const void *full, size_t rowbytes, int npj, int j0, int j1, size_t nslab, size_t slab_stride,
            hipStream_t st) {
    const size_t width = rowbytes * (size_t)(j1 - j0);
    const size_t spitch = slab_stride ? slab_stride : rowbytes * (size_t)npj;
    return rows_copy(blk, width, (const char *)full + rowbytes * (size_t)j0, spitch, width, nslab, st);
}

int rows_out(void *full, const void *blk, size_t rowbytes, int npj, int j0, int j1, size_t nslab, size_t slab_stride,
             hipStream_t st) {
    const size_t width = rowbytes * (size_t)(j1 - j0);
    const size_t dpitch = slab_stride ? slab_stride : rowbytes * (size_t)npj;
    return rows_copy((char *)full + rowbytes * (size_t)j0, dpitch, blk, width, width, nslab, st);
}

}  // namespace shard
}  // namespace cdft

extern "C" int cdft_shard_from_env(int *ndev, int *devices, int capacity) { return shard::from_env(ndev, devices, capacity); }

// =================================================================================================== cdfmoc
namespace cdft {
namespace shard {

int moc_parts(ShardSet &sh) {
    return for_blocks(sh, [&](int b) {
        cdft_moc_plan *p = nullptr;
        CDFT_TRY(cdft_moc_create(&p, sh.npi, sh.j1[b] - sh.j0[b], sh.npk, sh.nb));
        sh.part[b] = p;
        return CDFT_OK;
    });
}

int moc_set_mesh(ShardSet &sh, const float *e1v, const float *e3v, const int8_t *vmask, const int16_t *ibmask) {
    return for_blocks(sh, [&](int b) {
        const int j0 = sh.j0[b], j1 = sh.j1[b], nloc = j1 - j0;
        const size_t nij = (size_t)sh.npi * nloc, n3 = nij * sh.npk;
        hipStream_t st = sh.st[b];
        DevBuf *B = sh.buf[b].data();            // 0: e1v, 1: e3v, 2: vmask, 3: ibmask, (4: zv, 5: out)
        CDFT_TRY(B[0].reserve(nij * 4));
        CDFT_TRY(B[1].reserve(n3 * 4));
        CDFT_TRY(B[3].reserve(nij * sh.nb * 2));
        CDFT_TRY(rows_in(B[0].ptr, e1v, (size_t)sh.npi * 4, sh.npj, j0, j1, 1, 0, st));
        CDFT_TRY(rows_in(B[1].ptr, e3v, (size_t)sh.npi * 4, sh.npj, j0, j1, sh.npk, 0, st));
        CDFT_TRY(rows_in(B[3].ptr, ibmask, (size_t)sh.npi * sh.nb * 2, sh.npj, j0, j1, 1, 0, st));
        if (vmask) {
            CDFT_TRY(B[2].reserve(n3));
            CDFT_TRY(rows_in(B[2].ptr, vmask, (size_t)sh.npi, sh.npj, j0, j1, sh.npk, 0, st));
        }
        int rc = cdft_moc_set_mesh_vmask((cdft_moc_plan *)sh.part[b], (const float *)B[0].ptr, (const float *)B[1].ptr,
                                         vmask ? (const int8_t *)B[2].ptr : nullptr, (const int16_t *)B[3].ptr, CDFT_MEM_DEVICE, st);
        (void)hipStreamSynchronize(st);
        for (int i = 0; i < 4; ++i) B[i].release();
        return rc;
    });
}

// the records of a call in each block's HBM (B[4], dense, kept until the next call)
static int moc_rows_to_block(ShardSet &sh, int b, const float *zv, int64_t rec_stride, int npt) {
    const int j0 = sh.j0[b], j1 = sh.j1[b], nloc = j1 - j0;
    const size_t n3 = (size_t)sh.npi * nloc * sh.npk;
    DevBuf *B = sh.buf[b].data();
    CDFT_TRY(B[4].reserve(n3 * npt * 4));
    for (int t = 0; t < npt; ++t)
        CDFT_TRY(rows_in((float *)B[4].ptr + (size_t)t * n3, zv + (size_t)t * rec_stride, (size_t)sh.npi * 4, sh.npj, j0, j1,
                         sh.npk, 0, sh.st[b]));
    return CDFT_OK;
}

// dmoc(nbasins,npjglo,npk,npt): the rows of this block, every level and record, in place
static int moc_block_out(ShardSet &sh, int b, int npt, double *dmoc) {
    hipStream_t st = sh.st[b];
    CDFT_TRY(rows_out(dmoc, sh.buf[b][5].ptr, (size_t)sh.nb * 8, sh.npj, sh.j0[b], sh.j1[b], (size_t)sh.npk * npt, 0, st));
    CDFT_HIP(hipStreamSynchronize(st));
    return CDFT_OK;
}

int moc_compute(ShardSet &sh, const float *zv, int64_t rec_stride, int npt, double *dmoc) {
    return for_blocks(sh, [&](int b) {
        const int nloc = sh.j1[b] - sh.j0[b];
        DevBuf *B = sh.buf[b].data();
        CDFT_TRY(moc_rows_to_block(sh, b, zv, rec_stride, npt));
        CDFT_TRY(B[5].reserve((size_t)sh.nb * nloc * sh.npk * npt * 8));
        CDFT_TRY(cdft_moc_compute((cdft_moc_plan *)sh.part[b], (const float *)B[4].ptr, (int64_t)sh.npi * nloc * sh.npk, npt, CDFT_MEM_DEVICE,
                                  (double *)B[5].ptr, CDFT_MEM_DEVICE, sh.st[b]));
        return moc_block_out(sh, b, npt, dmoc);
    });
}

// cdft_moc_stage: every block takes its rows [j0,j1) of every level of every record where they lie, straight into the pool
// of its plan (moc.hip: a row's weights and records in one run)
int moc_stage(ShardSet &sh, const float *zv, int64_t rec_stride, int npt) {
    hipPointerAttribute_t at;
    const bool dev = hipPointerGetAttributes(&at, zv) == hipSuccess && at.type == hipMemoryTypeDevice;
    if (!dev) (void)hipGetLastError();
    return for_blocks(sh, [&](int b) {
        sh.buf[b][4].release();                                  // (the dense records of an earlier cdft_moc_compute)
        CDFT_TRY(cdft_moc_stage_rows((cdft_moc_plan *)sh.part[b], zv + (size_t)sh.npi * sh.j0[b], rec_stride, (int64_t)sh.npi * sh.npj, npt,
                                     dev ? CDFT_MEM_DEVICE : CDFT_MEM_HOST, sh.st[b]));
        CDFT_HIP(hipStreamSynchronize(sh.st[b]));
        return CDFT_OK;
    });
}

int moc_run(ShardSet &sh, int npt, double *dmoc) {
    return for_blocks(sh, [&](int b) {
        const int nloc = sh.j1[b] - sh.j0[b];
        DevBuf *B = sh.buf[b].data();
        CDFT_TRY(B[5].reserve((size_t)sh.nb * nloc * sh.npk * npt * 8));
        CDFT_TRY(cdft_moc_compute_staged((cdft_moc_plan *)sh.part[b], (double *)B[5].ptr, CDFT_MEM_DEVICE, sh.st[b]));
        return moc_block_out(sh, b, npt, dmoc);
    });
}

// =================================================================================================== cdfheatc
int heatc_parts(ShardSet &sh) {
    return for_blocks(sh, [&](int b) {
        cdft_heatc_plan *p = nullptr;
        CDFT_TRY(cdft_heatc_create(&p, sh.npi, sh.j1[b] - sh.j0[b], sh.npk));
        sh.part[b] = p;
        return CDFT_OK;
    });
}

int heatc_set_mesh(ShardSet &sh, const float *e1t, const float *e2t, const float *e3t, const float *e31d, const void *tmask,
                   int mask_is_r4) {
    return for_blocks(sh, [&](int b) {
        const int j0 = sh.j0[b], j1 = sh.j1[b], nloc = j1 - j0;
        const size_t nij = (size_t)sh.npi * nloc, n3 = nij * sh.npk, ms = mask_is_r4 ? 4 : 1;
        hipStream_t st = sh.st[b];
        DevBuf *B = sh.buf[b].data();            // 0: e1t, 1: e2t, 2: e3t, 3: tmask, 4: e31d (5: temp, 6: mxl)
        CDFT_TRY(B[0].reserve(nij * 4));
        CDFT_TRY(B[1].reserve(nij * 4));
        CDFT_TRY(B[3].reserve(n3 * ms));
        CDFT_TRY(rows_in(B[0].ptr, e1t, (size_t)sh.npi * 4, sh.npj, j0, j1, 1, 0, st));
        CDFT_TRY(rows_in(B[1].ptr, e2t, (size_t)sh.npi * 4, sh.npj, j0, j1, 1, 0, st));
        CDFT_TRY(rows_in(B[3].ptr, tmask, (size_t)sh.npi * ms, sh.npj, j0, j1, sh.npk, 0, st));
        if (e3t) {
            CDFT_TRY(B[2].reserve(n3 * 4));
            CDFT_TRY(rows_in(B[2].ptr, e3t, (size_t)sh.npi * 4, sh.npj, j0, j1, sh.npk, 0, st));
        } else {
            CDFT_TRY(B[4].reserve((size_t)sh.npk * 4));
            CDFT_HIP(hipMemcpyAsync(B[4].ptr, e31d, (size_t)sh.npk * 4, hipMemcpyDefault, st));
        }
        int rc = cdft_heatc_set_mesh((cdft_heatc_plan *)sh.part[b], (const float *)B[0].ptr, (const float *)B[1].ptr,
                                     e3t ? (const float *)B[2].ptr : nullptr, e3t ? nullptr : (const float *)B[4].ptr, B[3].ptr,
                                     mask_is_r4, CDFT_MEM_DEVICE, st);
        (void)hipStreamSynchronize(st);
        for (int i = 0; i < 5; ++i) B[i].release();
        return rc;
    });
}

int heatc_compute(ShardSet &sh, const float *temp, const float *rmxldep, const float *gdepw, int mxloption, double *sums,
                  int out_space, hipStream_t out_stream, int sum_mode) {
    const int n = sh.nblk();
    std::vector<std::vector<double>> part(n, std::vector<double>((size_t)3 * sh.npk));
    if (sum_mode == CDFT_HEATC_SUM_R4SEQ) {
        // The reference's sequential r4 sums run through the rows in memory order, i.e. through the blocks south to
        // north: stage every block's rows (in parallel), then run the chains block after block, each starting from the
        // r4 values the previous one ended with -- the same additions in the same order as on one device.
        CDFT_TRY(for_blocks(sh, [&](int b) {
            const int j0 = sh.j0[b], j1 = sh.j1[b];
            const size_t nij = (size_t)sh.npi * (j1 - j0);
            DevBuf *B = sh.buf[b].data();
            CDFT_TRY(B[5].reserve(nij * sh.npk * 4));
            CDFT_TRY(rows_in(B[5].ptr, temp, (size_t)sh.npi * 4, sh.npj, j0, j1, sh.npk, 0, sh.st[b]));
            if (mxloption != 0) {
                CDFT_TRY(B[6].reserve(nij * 4));
                CDFT_TRY(rows_in(B[6].ptr, rmxldep, (size_t)sh.npi * 4, sh.npj, j0, j1, 1, 0, sh.st[b]));
            }
            CDFT_HIP(hipStreamSynchronize(sh.st[b]));
            return CDFT_OK;
        }));
        std::vector<float> carry((size_t)3 * sh.npk);
        int cur = 0;
        (void)hipGetDevice(&cur);
        int rc = CDFT_OK;
        TraceMute mute_;                                         // (the blocks' calls below are inside the caller's: common.hpp)
        for (int b = 0; b < n && rc == CDFT_OK; ++b) {
            if (hipSetDevice(sh.dev[b]) != hipSuccess) { rc = fail(CDFT_ERR_HIP, "cdft_heatc_compute: hipSetDevice(%d) failed", sh.dev[b]); break; }
            cdft_heatc_plan *p = (cdft_heatc_plan *)sh.part[b];
            if (b > 0) rc = cdft_heatc_chain(p, carry.data(), nullptr);
            if (rc == CDFT_OK)
                rc = cdft_heatc_compute(p, (const float *)sh.buf[b][5].ptr, CDFT_MEM_DEVICE, mxloption ? (const float *)sh.buf[b][6].ptr : nullptr,
                                        gdepw, mxloption, part[b].data(), CDFT_MEM_HOST, sh.st[b]);
            if (rc == CDFT_OK) rc = cdft_heatc_chain(p, nullptr, carry.data());
        }
        (void)hipSetDevice(cur);
        CDFT_TRY(rc);
        const std::vector<double> &tot = part[n - 1];
        if (out_space == CDFT_MEM_HOST) {
            memcpy(sums, tot.data(), tot.size() * 8);
        } else {
            CDFT_HIP(hipMemcpyAsync(sums, tot.data(), tot.size() * 8, hipMemcpyHostToDevice, out_stream));
            CDFT_HIP(hipStreamSynchronize(out_stream));
        }
        return CDFT_OK;
    }
    const size_t nsum = (size_t)3 * sh.npk;
    std::vector<double> gathered(sh.rccl() ? nsum * n : 0);
    if (sh.rccl()) CDFT_TRY(comms_alive(sh, "cdft_heatc_compute"));
    // pass 1: everything that can fail on one block alone (status checked before any collective is issued)
    CDFT_TRY(for_blocks(sh, [&](int b) {
        const int j0 = sh.j0[b], j1 = sh.j1[b], nloc = j1 - j0;
        const size_t nij = (size_t)sh.npi * nloc;
        hipStream_t st = sh.st[b];
        DevBuf *B = sh.buf[b].data();
        CDFT_TRY(B[5].reserve(nij * sh.npk * 4));
        CDFT_TRY(rows_in(B[5].ptr, temp, (size_t)sh.npi * 4, sh.npj, j0, j1, sh.npk, 0, st));
        if (mxloption != 0) {
            CDFT_TRY(B[6].reserve(nij * 4));
            CDFT_TRY(rows_in(B[6].ptr, rmxldep, (size_t)sh.npi * 4, sh.npj, j0, j1, 1, 0, st));
        }
        if (!sh.rccl())
            return cdft_heatc_compute((cdft_heatc_plan *)sh.part[b], (const float *)B[5].ptr, CDFT_MEM_DEVICE,
                                      mxloption ? (const float *)B[6].ptr : nullptr, gdepw, mxloption, part[b].data(), CDFT_MEM_HOST, st);
        // RCCL: the block's sums stay in its HBM for the all-gather of pass 2
        CDFT_TRY(B[7].reserve(nsum * 8));
        CDFT_TRY(B[8].reserve(nsum * 8 * n));
        CDFT_TRY(cdft_heatc_compute((cdft_heatc_plan *)sh.part[b], (const float *)B[5].ptr, CDFT_MEM_DEVICE,
                                    mxloption ? (const float *)B[6].ptr : nullptr, gdepw, mxloption, (double *)B[7].ptr, CDFT_MEM_DEVICE, st));
        CDFT_HIP(hipStreamSynchronize(st));                       // (a failed kernel shows here, not behind the collective)
        return CDFT_OK;
    }));
    // pass 2 (RCCL): ONE all-gather hands every block the (npk,3) sums of all of them in block order (1.8 KB per block at 75
    // levels); block 0's copy comes to the host, where they are added as below
    if (sh.rccl())
        CDFT_TRY(for_blocks_collective(sh, [&](int b) {
            DevBuf *B = sh.buf[b].data();
            CDFT_RCCL(rccl::api()->AllGather(B[7].ptr, B[8].ptr, nsum, rccl::kDouble, sh.comm[b], sh.st[b]));
            return CDFT_OK;
        }, [&](int b) {
            hipStream_t st = sh.st[b];
            if (b == 0) CDFT_HIP(hipMemcpyAsync(gathered.data(), sh.buf[b][8].ptr, nsum * 8 * n, hipMemcpyDeviceToHost, st));
            CDFT_HIP(hipStreamSynchronize(st));
            return CDFT_OK;
        }));
    if (sh.rccl())
        for (int b = 0; b < n; ++b) memcpy(part[b].data(), gathered.data() + (size_t)b * nsum, nsum * 8);
    // src/cdfheatc.f90:225-229 sums the whole level; here the blocks' r8 sums are added in block order
    std::vector<double> tot(part[0]);
    for (int b = 1; b < n; ++b)
        for (size_t i = 0; i < tot.size(); ++i) tot[i] += part[b][i];
    if (out_space == CDFT_MEM_HOST) {
        memcpy(sums, tot.data(), tot.size() * 8);
    } else {
        CDFT_HIP(hipMemcpyAsync(sums, tot.data(), tot.size() * 8, hipMemcpyHostToDevice, out_stream));
        CDFT_HIP(hipStreamSynchronize(out_stream));
    }
    return CDFT_OK;
}

// =================================================================================================== cdfmhst
int mhst_parts(ShardSet &sh) {
    return for_blocks(sh, [&](int b) {
        cdft_mhst_plan *p = nullptr;
        CDFT_TRY(cdft_mhst_create(&p, sh.npi, sh.r1[b] - sh.j0[b], sh.npk, sh.nb));     // with the halo row to the north
        sh.part[b] = p;
        return CDFT_OK;
    });
}

int mhst_set_mesh(ShardSet &sh, const float *e1v, const float *e3v, const float *e31d, const float *masks) {
    return for_blocks(sh, [&](int b) {
        const int j0 = sh.j0[b], r1 = sh.r1[b], nrd = r1 - j0;
        const size_t nij = (size_t)sh.npi * nrd;
        hipStream_t st = sh.st[b];
        DevBuf *B = sh.buf[b].data();            // 0: e1v, 1: e3v, 2: masks, 3: e31d (4..6: fields, 7..8: out)
        CDFT_TRY(B[0].reserve(nij * 4));
        CDFT_TRY(B[2].reserve(nij * sh.nb * 4));
        CDFT_TRY(rows_in(B[0].ptr, e1v, (size_t)sh.npi * 4, sh.npj, j0, r1, 1, 0, st));
        CDFT_TRY(rows_in(B[2].ptr, masks, (size_t)sh.npi * 4, sh.npj, j0, r1, sh.nb, 0, st));
        if (e3v) {
            CDFT_TRY(B[1].reserve(nij * sh.npk * 4));
            CDFT_TRY(rows_in(B[1].ptr, e3v, (size_t)sh.npi * 4, sh.npj, j0, r1, sh.npk, 0, st));
        } else {
            CDFT_TRY(B[3].reserve((size_t)sh.npk * 4));
            CDFT_HIP(hipMemcpyAsync(B[3].ptr, e31d, (size_t)sh.npk * 4, hipMemcpyDefault, st));
        }
        int rc = cdft_mhst_set_mesh((cdft_mhst_plan *)sh.part[b], (const float *)B[0].ptr, e3v ? (const float *)B[1].ptr : nullptr,
                                    e3v ? nullptr : (const float *)B[3].ptr, (const float *)B[2].ptr, CDFT_MEM_DEVICE, st);
        (void)hipStreamSynchronize(st);
        for (int i = 0; i < 4; ++i) B[i].release();
        return rc;
    });
}

int mhst_compute(ShardSet &sh, const float *vt, const float *vs, const float *zv, const float *zt, const float *zs, int zdim,
                 double *heat, double *salt) {
    return for_blocks(sh, [&](int b) {
        const int j0 = sh.j0[b], j1 = sh.j1[b], r1 = sh.r1[b], nrd = r1 - j0;
        const size_t n3 = (size_t)sh.npi * nrd * sh.npk;
        const int nlo = zdim ? sh.npk : 1;
        hipStream_t st = sh.st[b];
        DevBuf *B = sh.buf[b].data();
        const float *in[3] = {vt ? vt : zv, vt ? vs : zt, vt ? nullptr : zs};
        for (int i = 0; i < 3; ++i)
            if (in[i]) {
                CDFT_TRY(B[4 + i].reserve(n3 * 4));
                CDFT_TRY(rows_in(B[4 + i].ptr, in[i], (size_t)sh.npi * 4, sh.npj, j0, r1, sh.npk, 0, st));
            }
        const size_t ob = (size_t)sh.nb * nrd * nlo * 8;
        CDFT_TRY(B[7].reserve(ob));
        CDFT_TRY(B[8].reserve(ob));
        const float *a = (const float *)B[4].ptr, *c = (const float *)B[5].ptr, *d = (const float *)B[6].ptr;
        CDFT_TRY(cdft_mhst_compute((cdft_mhst_plan *)sh.part[b], vt ? a : nullptr, vt ? c : nullptr, vt ? nullptr : a,
                                   vt ? nullptr : c, vt ? nullptr : d, CDFT_MEM_DEVICE, zdim, (double *)B[7].ptr, (double *)B[8].ptr,
                                   CDFT_MEM_DEVICE, st));
        // heat, salt (nbasins,npjglo[,npk]): keep the rows this block owns (its halo row belongs to the next block)
        for (int k = 0; k < nlo; ++k) {
            const size_t so = (size_t)k * sh.nb * nrd * 8, fo = (size_t)k * sh.nb * sh.npj * 8;
            CDFT_TRY(rows_out((char *)heat + fo, (char *)B[7].ptr + so, (size_t)sh.nb * 8, sh.npj, j0, j1, 1, 0, st));
            CDFT_TRY(rows_out((char *)salt + fo, (char *)B[8].ptr + so, (size_t)sh.nb * 8, sh.npj, j0, j1, 1, 0, st));
        }
        CDFT_HIP(hipStreamSynchronize(st));
        return CDFT_OK;
    });
}

// =================================================================================================== cdfpsi
int psi_parts(ShardSet &sh) {
    return for_blocks(sh, [&](int b) {
        cdft_psi_plan *p = nullptr;
        CDFT_TRY(cdft_psi_create(&p, sh.npi, sh.j1[b] - sh.j0[b], sh.npk));
        sh.part[b] = p;
        return CDFT_OK;
    });
}

// the three stages of cdft_psi_* on row blocks: the staged entry points of a sharded plan land here
int psi_integrate(ShardSet &sh, const float *zu, const float *e2u, const float *e3u, const float *zv, const float *e1v,
                  const float *e3v, const float *e31d) {
    // the vertical integrals (and the i-scan of the V path, local to a row): all blocks at once
    return for_blocks(sh, [&](int b) {
        const int j0 = sh.j0[b], j1 = sh.j1[b], nloc = j1 - j0;
        const size_t nij = (size_t)sh.npi * nloc, n3 = nij * sh.npk;
        hipStream_t st = sh.st[b];
        DevBuf *B = sh.buf[b].data();            // 0: zu 1: e2u 2: e3u 3: zv 4: e1v 5: e3v 6: e31d 7: fmask 8: carry 9: last 10: psi4 11: dpsi8
        const float *src3[2][2] = {{zu, e3u}, {zv, e3v}};
        const float *src2[2] = {e2u, e1v};
        for (int c = 0; c < 2; ++c) {
            if (!src3[c][0]) continue;
            CDFT_TRY(B[3 * c].reserve(n3 * 4));
            CDFT_TRY(rows_in(B[3 * c].ptr, src3[c][0], (size_t)sh.npi * 4, sh.npj, j0, j1, sh.npk, 0, st));
            CDFT_TRY(B[3 * c + 1].reserve(nij * 4));
            CDFT_TRY(rows_in(B[3 * c + 1].ptr, src2[c], (size_t)sh.npi * 4, sh.npj, j0, j1, 1, 0, st));
            if (src3[c][1]) {
                CDFT_TRY(B[3 * c + 2].reserve(n3 * 4));
                CDFT_TRY(rows_in(B[3 * c + 2].ptr, src3[c][1], (size_t)sh.npi * 4, sh.npj, j0, j1, sh.npk, 0, st));
            }
        }
        if (e31d) {
            CDFT_TRY(B[6].reserve((size_t)sh.npk * 4));
            CDFT_HIP(hipMemcpyAsync(B[6].ptr, e31d, (size_t)sh.npk * 4, hipMemcpyDefault, st));
        }
        CDFT_TRY(B[8].reserve((size_t)sh.npi * 8));
        CDFT_TRY(B[9].reserve((size_t)sh.npi * 8));
        auto f = [&](int i, const float *given) { return given ? (const float *)B[i].ptr : nullptr; };
        CDFT_TRY(cdft_psi_integrate((cdft_psi_plan *)sh.part[b], f(0, zu), f(1, zu ? e2u : nullptr), f(2, zu ? e3u : nullptr), f(3, zv),
                                    f(4, zv ? e1v : nullptr), f(5, zv ? e3v : nullptr), f(6, e31d), CDFT_MEM_DEVICE, st));
        CDFT_HIP(hipStreamSynchronize(st));
        return CDFT_OK;
    });
}

namespace {
// the caller's current device, put back however the function is left (the loops below switch devices and may return early)
struct DeviceGuard {
    int cur = 0;
    DeviceGuard() { (void)hipGetDevice(&cur); }
    ~DeviceGuard() { (void)hipSetDevice(cur); }
};
}  // namespace

int psi_scan_u(ShardSet &sh) {
    // the running difference along j crosses the blocks (src/cdfpsi.f90:395-398): visit them south to north, the
    // last row of one is the carry of the next -- 8*npiglo bytes from HBM to HBM, by a peer copy or, in RCCL mode, by an
    // ncclSend / ncclRecv pair (one host thread, two devices: inside a group)
    const int n = sh.nblk();
    DeviceGuard guard;
    if (sh.rccl()) CDFT_TRY(comms_alive(sh, "cdft_psi_scan_u"));
    for (int b = 0; b < n; ++b) {
        CDFT_HIP(hipSetDevice(sh.dev[b]));
        DevBuf *B = sh.buf[b].data();
        if (!B[8].ptr || !B[9].ptr) return fail(CDFT_ERR_STATE, "cdft_psi_scan_u: cdft_psi_integrate has not run");
        if (b > 0 && sh.rccl()) {
            const rccl::Api *api = rccl::api();
            CDFT_RCCL(api->GroupStart());
            CDFT_RCCL(api->Send(sh.buf[b - 1][9].ptr, (size_t)sh.npi, rccl::kDouble, b, sh.comm[b - 1], sh.st[b - 1]));
            CDFT_RCCL(api->Recv(B[8].ptr, (size_t)sh.npi, rccl::kDouble, b - 1, sh.comm[b], sh.st[b]));
            CDFT_RCCL(api->GroupEnd());
        } else if (b > 0) {
            CDFT_HIP(hipMemcpyPeerAsync(B[8].ptr, sh.dev[b], sh.buf[b - 1][9].ptr, sh.dev[b - 1], (size_t)sh.npi * 8, sh.st[b]));
        }
        CDFT_TRY(cdft_psi_scan_u((cdft_psi_plan *)sh.part[b], b > 0 ? (const double *)B[8].ptr : nullptr, CDFT_MEM_DEVICE, b == 0 ? 1 : 0,
                                 (double *)B[9].ptr, CDFT_MEM_DEVICE, sh.st[b]));
        if (hipStreamSynchronize(sh.st[b]) != hipSuccess) return fail(CDFT_ERR_HIP, "cdft_psi_scan_u: scan of block %d failed", b);
    }
    return CDFT_OK;
}

static int owner_of_row(const ShardSet &sh, int ij) {
    for (int b = 0; b < sh.nblk(); ++b)
        if (ij > sh.j0[b] && ij <= sh.j1[b]) return b;
    return -1;
}

int psi_value(ShardSet &sh, int ii, int ij, double *value) {
    if (ii < 1 || ii > sh.npi || ij < 1 || ij > sh.npj)
        return fail(CDFT_ERR_ARG, "cdft_psi_value: point (%d,%d) outside the domain", ii, ij);
    DeviceGuard guard;
    const int b = owner_of_row(sh, ij);                     // the block that owns the row
    CDFT_HIP(hipSetDevice(sh.dev[b]));
    return cdft_psi_value((cdft_psi_plan *)sh.part[b], ii, ij - sh.j0[b], value, sh.st[b]);
}

int psi_finish(ShardSet &sh, int use_ref, double ref_value, int iiref, int ijref, const float *fmask, float *psi4, double *dpsi8) {
    double ref0 = ref_value;
    const bool bcast = !use_ref && sh.rccl();
    if (!use_ref && !bcast) CDFT_TRY(psi_value(sh, iiref, ijref, &ref0));     // :423, from the block that owns the point
    if (bcast && (iiref < 1 || iiref > sh.npi || ijref < 1 || ijref > sh.npj))
        return fail(CDFT_ERR_ARG, "cdft_psi_value: point (%d,%d) outside the domain", iiref, ijref);
    const int owner = bcast ? owner_of_row(sh, ijref) : -1;
    if (bcast) {
        // RCCL mode: the owner's value reaches every block by ncclBroadcast.  First what can fail on one block alone (the
        // one-word buffers, the owner reading its value), checked; then the collective, every block reaching its call.
        CDFT_TRY(comms_alive(sh, "cdft_psi_finish"));
        CDFT_TRY(for_blocks(sh, [&](int b) {
            DevBuf *B = sh.buf[b].data();
            hipStream_t st = sh.st[b];
            CDFT_TRY(B[12].reserve(8));
            if (b == owner) {
                double v = 0.0;
                CDFT_TRY(cdft_psi_value((cdft_psi_plan *)sh.part[b], iiref, ijref - sh.j0[b], &v, st));
                CDFT_HIP(hipMemcpyAsync(B[12].ptr, &v, 8, hipMemcpyHostToDevice, st));
                CDFT_HIP(hipStreamSynchronize(st));
            }
            return CDFT_OK;
        }));
        std::vector<double> refs((size_t)sh.nblk(), 0.0);
        CDFT_TRY(for_blocks_collective(sh, [&](int b) {
            DevBuf *B = sh.buf[b].data();
            CDFT_RCCL(rccl::api()->Broadcast(B[12].ptr, B[12].ptr, 1, rccl::kDouble, owner, sh.comm[b], sh.st[b]));
            return CDFT_OK;
        }, [&](int b) {
            hipStream_t st = sh.st[b];
            CDFT_HIP(hipMemcpyAsync(&refs[b], sh.buf[b][12].ptr, 8, hipMemcpyDeviceToHost, st));
            CDFT_HIP(hipStreamSynchronize(st));
            return CDFT_OK;
        }));
        ref0 = refs[0];                                           // (every block received the owner's word: the same bits)
        for (int b = 1; b < sh.nblk(); ++b)
            if (memcmp(&refs[b], &ref0, 8) != 0) return fail(CDFT_ERR_HIP, "cdft_psi_finish: block %d received another reference value than block 0", b);
    }
    // normalisation, mask, SNGL: every block writes its rows of the caller's arrays
    return for_blocks(sh, [&](int b) {
        const int j0 = sh.j0[b], j1 = sh.j1[b];
        const size_t off = (size_t)sh.npi * j0, nij = (size_t)sh.npi * (j1 - j0);
        DevBuf *B = sh.buf[b].data();
        hipStream_t st = sh.st[b];
        const double ref = ref0;
        if (fmask) {
            CDFT_TRY(B[7].reserve(nij * 4));
            CDFT_TRY(rows_in(B[7].ptr, fmask, (size_t)sh.npi * 4, sh.npj, j0, j1, 1, 0, st));
        }
        CDFT_TRY(B[10].reserve(nij * 4));
        if (dpsi8) CDFT_TRY(B[11].reserve(nij * 8));
        CDFT_TRY(cdft_psi_finish((cdft_psi_plan *)sh.part[b], 1, ref, 1, 1, fmask ? (const float *)B[7].ptr : nullptr, CDFT_MEM_DEVICE,
                                 (float *)B[10].ptr, dpsi8 ? (double *)B[11].ptr : nullptr, CDFT_MEM_DEVICE, st));
        CDFT_HIP(hipMemcpyAsync(psi4 + off, B[10].ptr, nij * 4, hipMemcpyDefault, st));        // rows of a 2-D array: contiguous
        if (dpsi8) CDFT_HIP(hipMemcpyAsync(dpsi8 + off, B[11].ptr, nij * 8, hipMemcpyDefault, st));
        CDFT_HIP(hipStreamSynchronize(st));
        return CDFT_OK;
    });
}

// =================================================================================================== cdfvT
int vt_compute(int ndev, const int *devices, const float *u, const float *v, const float *t, const float *s, const float *e3u,
               const float *e3v, int64_t fstride, int nfr, int npi, int npj, int npk, void *vt, void *vs, void *ut, void *us,
               void *e3um, void *e3vm, int out_r4) {
    ShardSet *shp = nullptr;
    CDFT_TRY(make(&shp, npi, npj, npk, 0, ndev, devices, 1, 12));
    ShardSet &sh = *shp;
    const size_t os = out_r4 ? 4 : 8;
    int rc = for_blocks(sh, [&](int b) {
        const int j0 = sh.j0[b], j1 = sh.j1[b], r1 = sh.r1[b], nrd = r1 - j0;
        const size_t n3 = (size_t)npi * nrd * npk;
        hipStream_t st = sh.st[b];
        DevBuf *B = sh.buf[b].data();            // 0..5: u v t s e3u e3v (all frames), 6..11: results
        const float *in[6] = {u, v, t, s, e3u, e3v};
        void *out[6] = {vt, vs, ut, us, e3um, e3vm};
        for (int i = 0; i < 6; ++i) {
            if (!in[i]) continue;
            CDFT_TRY(B[i].reserve(n3 * nfr * 4));
            for (int f = 0; f < nfr; ++f)
                CDFT_TRY(rows_in((float *)B[i].ptr + (size_t)f * n3, in[i] + (size_t)f * fstride, (size_t)npi * 4, npj, j0, r1, npk, 0, st));
        }
        for (int i = 0; i < 6; ++i)
            if (out[i] && (i < 4 || e3u)) CDFT_TRY(B[6 + i].reserve(n3 * os));
        auto o = [&](int i) { return (out[i] && (i < 4 || e3u)) ? B[6 + i].ptr : nullptr; };
        CDFT_TRY(cdft_vt_compute((const float *)B[0].ptr, (const float *)B[1].ptr, (const float *)B[2].ptr, (const float *)B[3].ptr,
                                 e3u ? (const float *)B[4].ptr : nullptr, e3u ? (const float *)B[5].ptr : nullptr, (int64_t)n3, nfr, npi, nrd,
                                 npk, CDFT_MEM_DEVICE, o(0), o(1), o(2), o(3), o(4), o(5), out_r4, CDFT_MEM_DEVICE, st));
        // the rows this block owns: its north halo row was only read (src/cdfvT.f90:200-206 look at jj+1); the global
        // last row stays 0 as in the reference, and the kernel leaves the last row of whatever block it is given at 0,
        // which for an interior block is the halo row that is dropped here
        for (int i = 0; i < 6; ++i)
            if (o(i)) {
                for (int k = 0; k < npk; ++k)
                    CDFT_TRY(rows_out((char *)out[i] + (size_t)k * npi * npj * os, (char *)B[6 + i].ptr + (size_t)k * npi * nrd * os,
                                      (size_t)npi * os, npj, j0, j1, 1, 0, st));
            }
        CDFT_HIP(hipStreamSynchronize(st));
        return CDFT_OK;
    });
    destroy(shp);
    return rc;
}

}  // namespace shard
}  // namespace cdft
