"""Row-block (j) partition of the horizontal domain over the GPUs of one node.

cdfmoc's zonal sum runs along i inside a row and its cumulated sum along k inside a row, so
contiguous blocks of j rows are independent units: no data-path collective.  The only exchange is
collecting each rank's (npt,npk,npj_loc,nbasins) r8 result on rank 0, which writes moc.nc -- one
fused gather per run (RCCL on GPU tensors, gloo in the CPU tests).
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


def gather_rows(local: torch.Tensor, npjglo: int, group=None, dst: int = 0, pad_buf=None, recv_bufs=None):
    """Collect per-rank results (npt,npk,npj_loc,nb) on rank `dst` -> (npt,npk,npjglo,nb) there,
    None elsewhere.  Blocks are padded to the largest block so that one gather moves them all."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
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
    if rank == dst:
        bufs = recv_bufs if recv_bufs is not None else [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        out = local.new_empty((npt, npk, npjglo, nb))
        for r in range(world):
            a, b = row_block(npjglo, world, r)
            out[:, :, a:b] = bufs[r][:, :, : b - a]
        return out
    dist.gather(send, None, dst=dst, group=group)
    return None
