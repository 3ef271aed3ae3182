"""Row-block (j) partition of the horizontal domain over the GPUs of one node.

cdfmoc's zonal sum runs along i inside a row and its cumulated sum along k inside a row, so
contiguous blocks of j rows are independent units: no data-path collective.  The only exchange is
collecting each rank's (npt,npk,npj_loc,nbasins) r8 result on rank 0, which writes moc.nc -- one
fused gather per run (RCCL on GPU tensors, gloo in the CPU tests).

The other tools of the path shard over the same row blocks; what each one exchanges:

  cdfmhst   rows independent (from a vt file);    gather_block_rows of (npj_loc, nb); halo_window when
            from V,T,S it needs T/S row j+1       the products are formed on the fly (cdfmhst.f90:312-322)
  cdfvT     needs T/S row j+1                     halo_window: every rank READS one more row, no exchange
  cdfcurl   needs u / e1u row j+1 (-T: and j-1)    halo_window, same
  cdfheatc  per-level domain sums                 heatc_sums: all_gather of (npk,3) r8, summed in rank order
  cdfpsi    running difference along j (U path)   psi_chain: the last row of a block is the carry of the
                                                  next (point-to-point, rank order), the reference value
                                                  is broadcast by the rank that owns (iiref, ijref)

All of them are deterministic: a fixed rank order replaces the order of a single process, no atomics
and no reduction tree whose shape depends on timing.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def row_block(npjglo: int, world: int, rank: int):
    """Rows [j0, j1) owned by `rank`: blocks differ by at most one row, lower ranks get the extras."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    base, extra = divmod(npjglo, world)
    j0 = rank * base + min(rank, extra)
    j1 = j0 + base + (1 if rank < extra else 0)
    return j0, j1


def max_rows(npjglo: int, world: int) -> int:
    return -(-npjglo // world)


def gather_rows_start(local: torch.Tensor, npjglo: int, group=None, dst: int = 0, recv_bufs=None):
    """Start collecting per-rank results (npt,npk,npj_loc,nb) on rank `dst` without blocking: point-to-point
    transfers of each block's exact size straight into per-rank receive buffers (no padding copy, nothing sent by
    `dst` to itself), batched so that RCCL runs them side by side -- they arrive over different xGMI links -- on the
    communication stream while the caller computes the next step.  -> handle for gather_rows_finish.
    `local` and `recv_bufs` must not be touched until gather_rows_finish returned.
    recv_bufs: on `dst`, a list of world tensors (npt,npk,rows_of_rank_r,nb) to reuse (entry `dst` is ignored)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return (None, local, None, npjglo, world, dst)
    rank = dist.get_rank(group)
    npt, npk, nloc, nb = local.shape
    j0, j1 = row_block(npjglo, world, rank)
    if nloc != j1 - j0:
        raise ValueError(f"rank {rank}: local block has {nloc} rows, partition says {j1 - j0}")
    local = local.contiguous()
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    if rank != dst:
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, local, peer(dst), group)])
        return (works, local, None, npjglo, world, dst)
    bufs = list(recv_bufs) if recv_bufs is not None else [None] * world
    ops = []
    for r in range(world):
        if r == dst:
            bufs[r] = local
            continue
        a, b = row_block(npjglo, world, r)
        if bufs[r] is None:
            bufs[r] = local.new_empty((npt, npk, b - a, nb))
        elif tuple(bufs[r].shape) != (npt, npk, b - a, nb):
            raise ValueError(f"recv_bufs[{r}] has shape {tuple(bufs[r].shape)}, expected {(npt, npk, b - a, nb)}")
        ops.append(dist.P2POp(dist.irecv, bufs[r], peer(r), group))
    works = dist.batch_isend_irecv(ops)
    return (works, local, bufs, npjglo, world, dst)


def gather_rows_finish(handle, out=None):
    """Wait for a gather started by gather_rows_start; -> (npt,npk,npjglo,nb) on rank dst (written into `out` when
    given), None elsewhere."""
    works, local, bufs, npjglo, world, dst = handle[:6]
    if world == 1:
        return local
    for w in works:
        w.wait()                                  # the current stream now waits for the transfers
    if bufs is None:
        return None
    npt, npk, _, nb = local.shape
    if out is None:
        out = local.new_empty((npt, npk, npjglo, nb))
    torch.cat(bufs, dim=2, out=out)              # the blocks in rank order along j: one kernel, not one strided copy per rank
    return out


def gather_rows(local: torch.Tensor, npjglo: int, group=None, dst: int = 0, recv_bufs=None):
    """Collect per-rank results (npt,npk,npj_loc,nb) on rank `dst` -> (npt,npk,npjglo,nb) there,
    None elsewhere (gather_rows_start + gather_rows_finish)."""
    return gather_rows_finish(gather_rows_start(local, npjglo, group=group, dst=dst, recv_bufs=recv_bufs))


class StepPipeline:
    """The step loop of bench.py: `compute(out)` fills this rank's (npt,npk,npj_loc,nb) block of one step; with N > 1
    rank 0 collects every step's blocks.  Two result buffers: the gather of step i (communication stream) runs under
    the kernels of step i+1, and a buffer is handed to compute() again only after its gather has been waited for and
    assembled.  `result` is the last assembled (npt,npk,npjglo,nb) array on rank 0 (the local block when N = 1).
    The same class is driven on 2 gloo ranks with the oracle as `compute` by tests/test_shard.py."""

    def __init__(self, compute, npjglo, local_shape, dtype=torch.float64, device=None, group=None, nbuf=2):
        self.compute, self.npj, self.group = compute, npjglo, group
        self.world, self.rank = _world_rank(group)
        self.nbuf = nbuf if self.world > 1 else 1
        npt, npk, nloc, nb = local_shape
        self.local = [torch.empty(local_shape, dtype=dtype, device=device) for _ in range(self.nbuf)]
        self.recv = [None] * self.nbuf
        self.out = None
        if self.world > 1 and self.rank == 0:
            for b in range(self.nbuf):
                self.recv[b] = [None if r == 0 else torch.empty((npt, npk, row_block(npjglo, self.world, r)[1] -
                                                                  row_block(npjglo, self.world, r)[0], nb), dtype=dtype, device=device)
                                for r in range(self.world)]
            self.out = torch.empty((npt, npk, npjglo, nb), dtype=dtype, device=device)
        self.pending = [None] * self.nbuf
        self.i = 0
        self.gathers = 0
        self.result = None

    def _finish(self, b):
        if self.pending[b] is not None:
            res = gather_rows_finish(self.pending[b], out=self.out)
            self.pending[b] = None
            self.gathers += 1
            if res is not None:
                self.result = res

    def step(self):
        b = self.i % self.nbuf
        self.i += 1
        self._finish(b)                                   # step i - nbuf: gathered -> assembled on rank 0
        self.compute(self.local[b])
        if self.world > 1:
            self.pending[b] = gather_rows_start(self.local[b], self.npj, group=self.group, recv_bufs=self.recv[b])
        else:
            self.result = self.local[b]

    def drain(self):
        """Every started gather is waited for and assembled, oldest first."""
        for k in range(self.nbuf):
            self._finish((self.i + k) % self.nbuf)


def halo_window(npjglo: int, world: int, rank: int, north: int = 1, south: int = 0):
    """Rows to READ for a stencil that looks `north` rows ahead and `south` rows back (cdfvT,
    src/cdfvT.f90:259-262 and cdfcurl, src/cdfcurl.f90:327-334: north=1; cdfcurl -T averages the four
    surrounding F points, src/cdfcurl.f90:347-356: south=1 as well) -> (r0, r1, lo, hi): compute on
    rows [r0, r1) of the files and keep rows [lo, hi) of that result.  The reference leaves the
    rows its loops do not reach (global row npjglo; row 1 with -T) at their initial zero, and the
    kernels do the same for the edge rows of whatever block they are given: the edge blocks need no
    halo there, and every other block drops the rows it only read for a neighbour's sake."""
    j0, j1 = row_block(npjglo, world, rank)
    r0, r1 = max(j0 - south, 0), min(j1 + north, npjglo)
    return r0, r1, j0 - r0, j1 - r0


def _world_rank(group):
    if not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def gather_block_rows(local: torch.Tensor, npjglo: int, axis: int = 0, group=None, dst: int = 0):
    """gather_rows for any layout: `axis` is the j axis of `local`."""
    if local.dim() == 4 and axis == 2:
        return gather_rows(local, npjglo, group=group, dst=dst)
    # bring to (1, 1, nloc, rest) and back
    x = local.movedim(axis, 0)
    lead = x.shape[0]
    rest = x.shape[1:]
    flat = x.reshape(1, 1, lead, -1)
    out = gather_rows(flat, npjglo, group=group, dst=dst)
    if out is None:
        return None
    return out.reshape((npjglo,) + tuple(rest)).movedim(0, axis)


def heatc_sums(local: torch.Tensor, group=None) -> torch.Tensor:
    """cdfheatc, src/cdfheatc.f90:262-273: per-level (surface, volume, heat) sums of this rank's rows
    (npk,3) r8 -> the domain sums, identical on every rank: blocks are added in rank order."""
    world, _ = _world_rank(group)
    if world == 1:
        return local
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous(), group=group)
    total = parts[0].clone()
    for r in range(1, world):
        total += parts[r]
    return total


def psi_chain(stages, npjglo: int, iiref: int, ijref: int, fmask=None, group=None, device=None):
    """cdfpsi over row blocks, src/cdfpsi.f90:380-438.  `stages` is a cdftools_amd.hip.PsiPlan built
    for this rank's rows after its integrate() call (or anything with the same scan_u / value /
    finish methods): the U path's running difference along j is chained through the ranks in order
    (each block's last row is the next block's carry: same additions in the same order as one
    process, hence the same bits), the owner of the reference point broadcasts its value, and each
    rank finishes its own rows.  Returns finish()'s result for the local rows.

    The chain serialises only the 2D scan (8 bytes per column per rank hop); the vertical integral,
    which reads the 3D fields, runs concurrently on all ranks before it."""
    world, rank = _world_rank(group)
    if world == 1:
        stages.scan_u(None, True)
        return stages.finish(None, iiref, ijref, fmask)
    if not (1 <= ijref <= npjglo):
        raise ValueError(f"reference row {ijref} outside 1..{npjglo}")
    on_gpu = device is not None and torch.device(device).type == "cuda"
    carry = None
    if rank > 0:
        carry = torch.empty(stages.npi, dtype=torch.float64, device=device)
        dist.recv(carry, src=rank - 1, group=group)
        if not on_gpu:
            carry = carry.numpy()
    # on a GPU the carry and the last row stay in HBM: scan_u takes and returns device arrays through the C ABI
    # (memory-space tags of cdft_psi_scan_u), so a hop is one xGMI transfer of 8*npiglo bytes, no PCIe round trip
    last = stages.scan_u(carry, rank == 0, device_out=True) if on_gpu else stages.scan_u(carry, rank == 0)
    if rank < world - 1:
        dist.send(last if on_gpu else torch.as_tensor(last, dtype=torch.float64), dst=rank + 1, group=group)
    owner = next(r for r in range(world) if row_block(npjglo, world, r)[0] < ijref <= row_block(npjglo, world, r)[1])
    ref = torch.zeros(1, dtype=torch.float64, device=device)
    if rank == owner:
        ref[0] = stages.value(iiref, ijref - row_block(npjglo, world, rank)[0])
    dist.broadcast(ref, src=owner, group=group)
    return stages.finish(float(ref.item()), fmask=fmask)
