// shard.hpp -- row-block sharded plans: ONE process drives several GPUs (include/cdftools_hip.h, "sharded plans").
//
// The reference's only parallel split of these loops is OpenMP over jj (src/cdfmoc.f90:369-377, :382-388); its direct
// analogue on a multi-GPU node is the row-block (j) partition of SURVEY.md 8e, placed here BEHIND the drop-in
// boundary: a sharded plan is a shell that owns one ordinary single-device plan per row block (each created and used
// under hipSetDevice(its device) on its own stream, from its own host thread so that the PCIe transfers of pageable
// host arrays run side by side), scatters the caller's full-domain arrays to them with strided copies and lands the
// results in place in the caller's arrays.  Every public entry point of a shardable plan starts with
// `if (p->sh) return shard::...`, so the Fortran programs use the same calls for 1 and N devices.
//
// Exchanges between blocks (none for cdfmoc / cdfmhst / cdfvT, whose rows are independent):
//   cdfheatc  (npk,3) r8 partial sums of each block, added in block order on the host;
//   cdfpsi    the last row of block b (8*npiglo bytes) is the carry of block b+1: device-to-device copy
//             (hipMemcpyPeerAsync, xGMI between distinct GPUs), blocks visited in order -- the additions and their
//             order are those of one device, so the result is bit-identical for any number of blocks;
//             the owner of (iiref,ijref) hands its value to the others through the host (one r8).
// Two transports for these exchanges (cdft_shard_set_collective / CDFT_SHARD_COLLECTIVE):
//   peer (default)  what is described above: within one process the devices address each other directly, the payloads
//                   are KBs, and it also runs with two blocks on one device -- which is how the path is tested on a 1-GPU
//                   box (CDFT_GPU_MAP=0,0);
//   rccl            the same exchanges as RCCL collectives over xGMI (rccl_dl.hpp): cdfheatc's block sums by ONE
//                   ncclAllGather of (npk,3) r8 per block, added in block order by every block -- the additions of the
//                   default, bit for bit; cdfpsi's carry hop as an ncclSend / ncclRecv pair and its reference value by
//                   ncclBroadcast from the block that owns the point.  RCCL refuses two ranks on one device, so this mode
//                   needs distinct devices.
// RCCL is also the transport where the ranks are separate processes (cdftools_amd/shard.py, bench.py).
#pragma once
#include <functional>
#include <string>
#include <vector>
#include "common.hpp"

namespace cdft {

struct ShardSet {
    int npi = 0, npj = 0, npk = 0, nb = 0;
    std::vector<int> dev, j0, j1;          // block b: device, rows [j0,j1) of the domain it OWNS
    std::vector<int> r1;                   // rows [j0,r1) it READS (r1 = j1 + 1 with a north halo row, else j1)
    std::vector<hipStream_t> st;           // one stream per block, created on its device
    std::vector<void *> part;              // the single-device plans (opaque: cdft_moc_plan*, ...)
    std::vector<std::vector<DevBuf>> buf;  // per block: staging buffers in that device's HBM
    std::vector<void *> comm;              // CDFT_COLLECTIVE_RCCL: one RCCL communicator per block (ncclCommInitAll), else empty
    int nblk() const { return (int)dev.size(); }
    bool rccl() const { return !comm.empty(); }
};

namespace shard {

// CDFT_GPU_MAP="a,b,..." (block b on device map[b]; duplicates allowed) or CDFT_GPUS=n (devices 0..n-1).
int from_env(int *ndev, int *devices, int capacity);
// rows [j0,j1) of block b out of n over npj rows: sizes differ by at most one, the first blocks get the extras
void row_block(int npj, int n, int b, int *j0, int *j1);
int make(ShardSet **out, int npi, int npj, int npk, int nb, int ndev, const int *devices, int halo_north, int nbuf);
void destroy(ShardSet *sh);
// run fn(b) for every block, each on its own host thread with its device current; the first failure wins
int for_blocks(ShardSet &sh, const std::function<int(int)> &fn);
// Copy rows [j0,j1) of `nslab` slabs between a full-domain array (slab = rowbytes*npj bytes, slabs `slab_stride`
// bytes apart; 0 = contiguous) and a packed block (slab = rowbytes*(j1-j0) bytes, contiguous).  Any memory spaces.
int rows_in(void *blk, const void *full, size_t rowbytes, int npj, int j0, int j1, size_t nslab, size_t slab_stride,
            hipStream_t st);
int rows_out(void *full, const void *blk, size_t rowbytes, int npj, int j0, int j1, size_t nslab, size_t slab_stride,
             hipStream_t st);

}  // namespace shard
}  // namespace cdft
