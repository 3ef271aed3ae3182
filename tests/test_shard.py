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


def _run_ranks(target, world, args, results=1, timeout=420):
    """Start `world` processes of target(rank, world, port, *args, q) and return the `results` objects they put on q.  The
    port is found by binding and closing, so another process can take it in between, and on a busy machine the first
    `import torch` of a spawned process alone can take a minute: one more attempt, on another port, before failing."""
    ctx = mp.get_context("spawn")
    last = None
    for attempt in range(2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            out = [q.get(timeout=timeout) for _ in range(results)]
            for p in procs:
                p.join(timeout)
            if all(p.exitcode == 0 for p in procs):
                return out
            last = "exit codes %s" % [p.exitcode for p in procs]
        except Exception as exc:                                   # queue.Empty: a rank never got there
            last = repr(exc)
        for p in procs:
            if p.is_alive():
                p.kill()
            p.join(30)
    raise AssertionError("the ranks did not finish in two attempts: " + str(last))


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
        # two gathers in flight (bench.py overlaps the gather of step i with the kernels of step i+1)
        t1, t2 = torch.from_numpy(loc), torch.from_numpy(loc) * 2.0
        h1 = shard.gather_rows_start(t1, npj)
        h2 = shard.gather_rows_start(t2, npj)
        f1, f2 = shard.gather_rows_finish(h1), shard.gather_rows_finish(h2)
        if rank == 0:
            ref = oracle_lib.moc(e1v, e3v, ib, zv)
            q.put(bool(np.array_equal(full.numpy(), ref) and np.array_equal(f1.numpy(), ref)
                       and np.array_equal(f2.numpy(), 2.0 * ref)))
        else:
            assert full is None and f1 is None and f2 is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("npj", [18, 19])            # even and ragged split
def test_two_rank_gloo_gather_matches_single_process(npj):
    assert _run_ranks(_worker, 2, (24, npj, 7, 5, 2)) == [True]


def _pipeline_worker(rank, world, port, npi, npj, npk, nb, npt, steps, q):
    """bench.py's step loop (shard.StepPipeline: double-buffered results, exact-size gather of every step on rank 0,
    drain) on gloo ranks with the oracle standing in for plan.compute; step s scales the velocity by s+1 so that every
    step has its own answer."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.make_grid(npi, npj, npk, seed=4, nbasins=nb)
        e1v, e3v, ib = synth.moc_inputs(g)
        zv = synth.velocity(g, npt, seed=10)
        j0, j1 = shard.row_block(npj, world, rank)
        state = {"s": 0}

        def compute(out):
            state["s"] += 1
            z = (zv[:, :, j0:j1] * np.float32(state["s"])).astype(np.float32)
            out.copy_(torch.from_numpy(oracle_lib.moc(e1v[j0:j1], e3v[:, j0:j1], ib[j0:j1], z)))
        pipe = shard.StepPipeline(compute, npj, (npt, npk, j1 - j0, nb))
        seen = []
        for _ in range(steps):
            pipe.step()
            seen.append(None if pipe.result is None else pipe.result.clone())
        pipe.drain()
        if rank == 0:
            ref = [oracle_lib.moc(e1v, e3v, ib, (zv * np.float32(k + 1)).astype(np.float32)) for k in range(steps)]
            ok = pipe.gathers == steps and np.array_equal(pipe.result.numpy(), ref[-1])
            for k in range(steps):                       # after step k the newest ASSEMBLED result is step k - nbuf
                want = k - pipe.nbuf
                ok = ok and ((seen[k] is None) if want < 0 else np.array_equal(seen[k].numpy(), ref[want]))
            q.put(bool(ok))
        else:
            assert pipe.result is None and pipe.gathers == steps
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,npj,steps", [(2, 19, 5), (3, 20, 4), (2, 18, 1)])
def test_bench_step_pipeline_on_gloo_ranks_collects_every_step(world, npj, steps):
    assert _run_ranks(_pipeline_worker, world, (24, npj, 6, 5, 2, steps)) == [True]


def test_bench_step_pipeline_single_process():
    g = synth.make_grid(24, 9, 4, seed=4)
    e1v, e3v, ib = synth.moc_inputs(g)
    zv = synth.velocity(g, 2, seed=10)
    ref = oracle_lib.moc(e1v, e3v, ib, zv)
    pipe = shard.StepPipeline(lambda out: out.copy_(torch.from_numpy(ref)), 9, ref.shape)
    for _ in range(3):
        pipe.step()
    pipe.drain()
    assert pipe.nbuf == 1 and pipe.gathers == 0 and np.array_equal(pipe.result.numpy(), ref)


def test_bench_self_launch_builds_a_child_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts a torch.distributed.run CHILD (never an exec of a process
    that may have touched the GPU) and relays its exit code."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


# ------------------------------------------------------------------ the other tools of the path
def test_halo_windows():
    assert shard.halo_window(19, 2, 0) == (0, 11, 0, 10)
    assert shard.halo_window(19, 2, 1) == (10, 19, 0, 9)               # last block: no halo row to read
    assert shard.halo_window(19, 2, 1, south=1) == (9, 19, 1, 10)
    assert shard.halo_window(19, 1, 0, south=1) == (0, 19, 0, 19)


class NumpyPsiStages:
    """The three stages of cdftools_amd.hip.PsiPlan restated with numpy for ONE row block (checker
    for shard.psi_chain; same arithmetic as oracle_psi: r4 products, r8 sums, sequential scans)."""

    def __init__(self, npk, zu=None, e2u=None, e3u=None, zv=None, e1v=None, e3v=None):
        self.lu, self.lv = zu is not None, zv is not None
        ref = e2u if self.lu else e1v
        self.npj, self.npi = ref.shape
        if self.lu:
            self.dtrpu = np.zeros(ref.shape)
            for k in range(npk):
                self.dtrpu += ((zu[k] * e2u) * e3u[k]).astype(np.float64)
        if self.lv:
            dtrpv = np.zeros(ref.shape)
            for k in range(npk):
                dtrpv += ((zv[k] * e1v) * e3v[k]).astype(np.float64)
            self.dpsiv = np.zeros(ref.shape)
            for i in range(self.npi - 2, -1, -1):
                self.dpsiv[:, i] = self.dpsiv[:, i + 1] - dtrpv[:, i]

    def scan_u(self, carry, first_block):
        if not self.lu:
            return np.zeros(self.npi)
        self.dpsiu = np.zeros((self.npj, self.npi))
        acc = np.zeros(self.npi) if carry is None else np.array(carry, np.float64)
        for j in range(self.npj):
            if not (first_block and j == 0):
                acc = acc - self.dtrpu[j]
            self.dpsiu[j] = acc
        return self.dpsiu[-1].copy()

    def _val(self):
        if self.lu and self.lv:
            return 0.5 * (self.dpsiu + self.dpsiv)
        return self.dpsiu if self.lu else self.dpsiv

    def value(self, ii, ij):
        return float(self._val()[ij - 1, ii - 1])

    def finish(self, ref_value=None, iiref=None, ijref=None, fmask=None):
        v = self._val()
        d = v - (v[ijref - 1, iiref - 1] if ref_value is None else ref_value)
        p4 = d.astype(np.float32)
        return (p4 * fmask if fmask is not None else p4), d


def _tools_worker(rank, world, port, npi, npj, npk, q):
    import oracle_tools
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = {}
    try:
        g = synth.make_grid(npi, npj, npk, seed=5)
        rng = np.random.default_rng(11)
        u = (rng.standard_normal((2, npk, npj, npi)) * g["umask"]).astype(np.float32)
        v = (rng.standard_normal((2, npk, npj, npi)) * g["vmask"]).astype(np.float32)
        t = ((10 + rng.standard_normal((2, npk, npj, npi))) * g["tmask"]).astype(np.float32)
        s = ((35 + rng.standard_normal((2, npk, npj, npi))) * g["tmask"]).astype(np.float32)
        j0, j1 = shard.row_block(npj, world, rank)

        # cdfmhst: rows are independent with a vt file; from V,T,S each block reads T/S row j+1 too
        masks = np.stack([g["vmask"][0], g["tmaskatl"], g["tmaskpac"]]).astype(np.float32)
        h, sl = oracle_tools.mhst(g["e1v"][j0:j1], masks[:, j0:j1], e3v=g["e3v_0"][:, j0:j1], vt=t[0][:, j0:j1],
                                  vs=s[0][:, j0:j1])
        full = shard.gather_block_rows(torch.from_numpy(np.stack([h, sl])), npj, axis=1)
        r0, r1, lo, hi = shard.halo_window(npj, world, rank)
        h, sl = oracle_tools.mhst(g["e1v"][r0:r1], masks[:, r0:r1], e3v=g["e3v_0"][:, r0:r1], zv=v[0][:, r0:r1],
                                  zt=t[0][:, r0:r1], zs=s[0][:, r0:r1])
        full2 = shard.gather_block_rows(torch.from_numpy(np.stack([h[lo:hi], sl[lo:hi]])), npj, axis=1)
        if rank == 0:
            rh, rs = oracle_tools.mhst(g["e1v"], masks, e3v=g["e3v_0"], vt=t[0], vs=s[0])
            ok["mhst"] = bool(np.array_equal(full.numpy(), np.stack([rh, rs])))
            rh, rs = oracle_tools.mhst(g["e1v"], masks, e3v=g["e3v_0"], zv=v[0], zt=t[0], zs=s[0])
            ok["mhst"] = ok["mhst"] and bool(np.array_equal(full2.numpy(), np.stack([rh, rs])))

        # cdfvT and cdfcurl: one halo row read by each block but the last
        r0, r1, lo, hi = shard.halo_window(npj, world, rank)
        loc = oracle_tools.vt_level(u[:, 1, r0:r1], v[:, 1, r0:r1], t[:, 1, r0:r1], s[:, 1, r0:r1])
        st = np.stack([loc[k][lo:hi] for k in ("vt", "vs", "ut", "us")])
        full = shard.gather_block_rows(torch.from_numpy(st), npj, axis=1)
        if rank == 0:
            ref = oracle_tools.vt_level(u[:, 1], v[:, 1], t[:, 1], s[:, 1])
            ok["vt"] = bool(np.array_equal(full.numpy(), np.stack([ref[k] for k in ("vt", "vs", "ut", "us")])))
        for name, tp in (("curl", False), ("curlT", True)):
            r0, r1, lo, hi = shard.halo_window(npj, world, rank, south=1 if tp else 0)
            loc = oracle_tools.curl(u[0, 0, r0:r1], v[0, 0, r0:r1], g["e1u"][r0:r1], g["e2v"][r0:r1], g["e1f"][r0:r1],
                                    g["e2f"][r0:r1], True, tp)[lo:hi]
            full = shard.gather_block_rows(torch.from_numpy(loc), npj, axis=0)
            if rank == 0:
                ref = oracle_tools.curl(u[0, 0], v[0, 0], g["e1u"], g["e2v"], g["e1f"], g["e2f"], True, tp)
                ok[name] = bool(np.array_equal(full.numpy(), ref) and np.count_nonzero(ref) > (0 if tp else 10))

        # cdfheatc: per-level sums added in rank order (r8: regrouping moves the last bits only)
        loc = oracle_tools.heatc(g["e1t"][j0:j1], g["e2t"][j0:j1], g["tmask"][:, j0:j1], t[0][:, j0:j1], e3t=g["e3t_0"][:, j0:j1])
        tot = shard.heatc_sums(torch.from_numpy(loc)).numpy()
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], t[0], e3t=g["e3t_0"])
        ok[f"heatc{rank}"] = bool(np.allclose(tot, ref, rtol=1e-13, atol=0))
        both = [torch.empty(tot.shape, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, torch.from_numpy(tot))
        ok[f"heatc_same{rank}"] = all(torch.equal(both[0], b) for b in both)

        # cdfpsi: chained scan, broadcast reference value
        fm = np.minimum(g["fmask"][0], 1).astype(np.float32)
        for name, kw, iiref, ijref in (("psiU", dict(u=True), npi, npj), ("psiV", dict(v=True), 3, 2),
                                       ("psiMean", dict(u=True, v=True), 5, j1 if rank == 0 else j0)):
            sync = torch.tensor([ijref])                                   # rank 0's choice: its own last row
            dist.broadcast(sync, src=0)
            ijref = int(sync.item())
            a, full_a = {}, {}
            if kw.get("u"):
                a.update(zu=u[0][:, j0:j1], e2u=g["e2u"][j0:j1], e3u=g["e3u_0"][:, j0:j1])
                full_a.update(zu=u[0], e2u=g["e2u"], e3u=g["e3u_0"])
            if kw.get("v"):
                a.update(zv=v[0][:, j0:j1], e1v=g["e1v"][j0:j1], e3v=g["e3v_0"][:, j0:j1])
                full_a.update(zv=v[0], e1v=g["e1v"], e3v=g["e3v_0"])
            p4, d8 = shard.psi_chain(NumpyPsiStages(npk, **a), npj, iiref, ijref, fmask=fm[j0:j1])
            full4 = shard.gather_block_rows(torch.from_numpy(np.ascontiguousarray(p4)), npj, axis=0)
            full8 = shard.gather_block_rows(torch.from_numpy(np.ascontiguousarray(d8)), npj, axis=0)
            if rank == 0:
                r4, r8 = oracle_tools.psi(npk, iiref=iiref, ijref=ijref, fmask=fm, **full_a)
                ok[name] = bool(np.array_equal(full4.numpy(), r4) and np.array_equal(full8.numpy(), r8))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,npj", [(2, 19), (3, 20)])
def test_row_block_sharding_of_the_other_tools_matches_single_process(world, npj):
    res = dict(_run_ranks(_tools_worker, world, (23, npj, 5), results=world))
    bad = [k for k in ("mhst", "vt", "curl", "curlT", "psiU", "psiV", "psiMean") if not res[0].get(k)]
    assert not bad, bad
    for r in range(world):
        assert res[r][f"heatc{r}"] and res[r][f"heatc_same{r}"], res[r]
