"""Row-block partition + result gather (cdftools_amd/shard.py), the N>1 path of bench.py.
World-size-2 gloo processes on CPU: each rank runs the cdfmoc oracle on its rows, rank 0 gathers
and must reproduce the single-process result bit for bit (rows are independent units)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
import synth
from cdftools_amd import shard


def test_row_blocks_tile_the_domain():
    for npj in (1, 2, 7, 149, 3606):
        for world in (1, 2, 3, 4, 8):
            if world > npj:
                continue
            edges = [shard.row_block(npj, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == npj
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == shard.max_rows(npj, world)
    with pytest.raises(ValueError):
        shard.row_block(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, npi, npj, npk, nb, npt, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.make_grid(npi, npj, npk, seed=3, nbasins=nb)
        e1v, e3v, ib = synth.moc_inputs(g)
        zv = synth.velocity(g, npt, seed=9)
        j0, j1 = shard.row_block(npj, world, rank)
        loc = oracle_lib.moc(e1v[j0:j1], e3v[:, j0:j1], ib[j0:j1], zv[:, :, j0:j1])
        full = shard.gather_rows(torch.from_numpy(loc), npj)
        if rank == 0:
            ref = oracle_lib.moc(e1v, e3v, ib, zv)
            q.put(bool(np.array_equal(full.numpy(), ref)))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("npj", [18, 19])            # even and ragged split
def test_two_rank_gloo_gather_matches_single_process(npj):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 24, npj, 7, 5, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True
