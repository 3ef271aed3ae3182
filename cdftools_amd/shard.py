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


def gather_rows_start(local: torch.Tensor, npjglo: int, group=None, dst: int = 0, pad_buf=None, recv_bufs=None):
    """Start collecting per-rank results (npt,npk,npj_loc,nb) on rank `dst` without blocking: the
    gather runs on the communication stream while the caller computes the next step.  Blocks are
    padded to the largest block so that one gather moves them all.  -> handle for gather_rows_finish.
    The buffers (local, pad_buf, recv_bufs) must not be touched until gather_rows_finish returned."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return (None, local, None, npjglo, world, dst)
    rank = dist.get_rank(group)
    npt, npk, nloc, nb = local.shape
    j0, j1 = row_block(npjglo, world, rank)
    if nloc != j1 - j0:
        raise ValueError(f"rank {rank}: local block has {nloc} rows, partition says {j1 - j0}")
    nmax = max_rows(npjglo, world)
    if nloc == nmax:
        send = local.contiguous()
    else:
        send = pad_buf if pad_buf is not None else local.new_zeros((npt, npk, nmax, nb))
        send[:, :, :nloc] = local
    bufs = None
    if rank == dst:
        bufs = recv_bufs if recv_bufs is not None else [torch.empty_like(send) for _ in range(world)]
    work = dist.gather(send, bufs, dst=dst, group=group, async_op=True)
    return (work, local, bufs, npjglo, world, dst, send)


def gather_rows_finish(handle):
    """Wait for a gather started by gather_rows_start; -> (npt,npk,npjglo,nb) on rank dst, None elsewhere."""
    work, local, bufs, npjglo, world, dst = handle[:6]
    if world == 1:
        return local
    work.wait()                                   # the current stream now waits for the gather
    if bufs is None:
        return None
    npt, npk, _, nb = local.shape
    out = local.new_empty((npt, npk, npjglo, nb))
    for r in range(world):
        a, b = row_block(npjglo, world, r)
        out[:, :, a:b] = bufs[r][:, :, : b - a]
    return out


def gather_rows(local: torch.Tensor, npjglo: int, group=None, dst: int = 0, pad_buf=None, recv_bufs=None):
    """Collect per-rank results (npt,npk,npj_loc,nb) on rank `dst` -> (npt,npk,npjglo,nb) there,
    None elsewhere (gather_rows_start + gather_rows_finish)."""
    return gather_rows_finish(gather_rows_start(local, npjglo, group=group, dst=dst, pad_buf=pad_buf, recv_bufs=recv_bufs))


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
    carry = None
    if rank > 0:
        buf = torch.empty(stages.npi, dtype=torch.float64, device=device)
        dist.recv(buf, src=rank - 1, group=group)
        carry = buf.cpu().numpy()
    last = stages.scan_u(carry, rank == 0)
    if rank < world - 1:
        dist.send(torch.as_tensor(last, dtype=torch.float64, device=device), dst=rank + 1, group=group)
    owner = next(r for r in range(world) if row_block(npjglo, world, r)[0] < ijref <= row_block(npjglo, world, r)[1])
    ref = torch.zeros(1, dtype=torch.float64, device=device)
    if rank == owner:
        ref[0] = stages.value(iiref, ijref - row_block(npjglo, world, rank)[0])
    dist.broadcast(ref, src=owner, group=group)
    return stages.finish(float(ref.item()), fmask=fmask)
