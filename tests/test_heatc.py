"""cdfheatc level sums (src/cdfheatc.f90:205-240).

Tolerances (stated): the reference evaluates `sum(e1t*e2t*e3t*temp*tmask)` with an r4 SUM, so its
own value depends on the compiler at ~1e-4.  The HIP path forms the same r4
products and accumulates in r8 in a fixed order:
  - vs the "r4 products, r8 sequential sum" restatement: |d| <= 1e-13 * sum|terms|
  - vs the "r4 SUM" restatement (what flang -O1 inlines for the reference): relative 1e-3.
"""
import numpy as np
import pytest

import oracle_tools
import synth

f32 = np.float32


def fields(npi, npj, npk, seed):
    g = synth.make_grid(npi, npj, npk, seed=seed)
    rng = np.random.default_rng(seed + 1)
    temp = (-2.0 + 32.0 * rng.random((npk, npj, npi), dtype=np.float32)).astype(np.float32)
    mxl = (10.0 + 300.0 * rng.random((npj, npi), dtype=np.float32)).astype(np.float32)
    return g, temp, mxl


def numpy_heatc(e1t, e2t, e3t, tmask, temp, r4sum):
    a = (e1t * e2t).astype(np.float32)[None]
    m = tmask.astype(np.float32)
    ps = (a * m).astype(np.float32)
    pv = ((a * e3t).astype(np.float32) * m).astype(np.float32)
    pt = (((a * e3t).astype(np.float32) * temp).astype(np.float32) * m).astype(np.float32)
    acc = np.float32 if r4sum else np.float64
    npk = temp.shape[0]
    return np.stack([np.cumsum(x.reshape(npk, -1).astype(acc), axis=1)[:, -1].astype(np.float64)
                     for x in (ps, pv, pt)], axis=1)


# ----------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("r4sum", [False, True])
def test_oracle_matches_numpy_restatement_bitwise(r4sum):
    g, temp, _ = fields(37, 23, 6, seed=2)
    got = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], r4sum=r4sum)
    assert np.array_equal(got, numpy_heatc(g["e1t"], g["e2t"], g["e3t_0"], g["tmask"], temp, r4sum))


def test_r4_sum_gap_is_what_the_survey_measured():
    g, temp, _ = fields(1442, 120, 3, seed=3)
    a = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], r4sum=False)
    b = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], r4sum=True)
    rel = np.abs(a - b) / np.abs(a).max(axis=0)
    assert rel.max() < 1e-3 and rel.max() > 1e-9      # visible, and inside the stated tolerance


def test_oracle_mixed_layer_options():
    g, temp, mxl = fields(16, 9, 5, seed=4)
    base = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"])
    inside = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], mxl=mxl,
                                gdepw=g["gdepw_1d"], mxlopt=1)
    below = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], mxl=mxl,
                               gdepw=g["gdepw_1d"], mxlopt=-1)
    # the clipped thicknesses partition each cell: volume above + below the mixed layer = total
    assert np.allclose(inside[:, 1] + below[:, 1], base[:, 1], rtol=1e-5)
    assert np.array_equal(inside[:, 0], base[:, 0])   # the surface does not depend on e3t


# ----------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk", [(1, 1, 1), (3, 5, 2), (255, 1, 2), (257, 3, 3), (1442, 33, 2), (182, 149, 31)])
def test_hip_reference_order_sums_equal_the_r4sum_oracle_bitwise(gpu, npi, npj, npk):
    """CDFT_HEATC_SUM_R4SEQ: the sequential r4 SUM of src/cdfheatc.f90:225-229 (pinned to the reference's own statements by
    tests/test_lift.py) -- byte and r4 masks, -full, both -mxloption values, separate plans chained by cdft_heatc_chain."""
    g, temp, mxl = fields(npi, npj, npk, seed=npi + 7)
    for mask in (g["tmask"].astype(np.uint8), (g["tmask"] * np.float32(0.75)).astype(np.float32)):
        for kw_mesh, kw_or in ((dict(e3t=g["e3t_0"]), dict(e3t=g["e3t_0"])), (dict(e31d=g["e3t_1d"]), dict(e31d=g["e3t_1d"]))):
            plan = gpu.HeatcPlan(npi, npj, npk)
            plan.set_mesh(g["e1t"], g["e2t"], mask, **kw_mesh)
            plan.set_sum_mode(True)
            for opt in (0, 1, -1):
                got = plan.compute(temp, rmxldep=mxl if opt else None, gdepw=g["gdepw_1d"], mxloption=opt)
                ref = oracle_tools.heatc(g["e1t"], g["e2t"], mask, temp, mxl=mxl if opt else None, gdepw=g["gdepw_1d"], mxlopt=opt,
                                         r4sum=True, **kw_or)
                assert np.array_equal(got, ref), (mask.dtype, list(kw_mesh), opt)
    if npj >= 3:                                      # two plans, the second continuing the first one's chains
        j = npj // 2
        a, b = gpu.HeatcPlan(npi, j, npk), gpu.HeatcPlan(npi, npj - j, npk)
        m8 = g["tmask"].astype(np.uint8)
        a.set_mesh(g["e1t"][:j], g["e2t"][:j], m8[:, :j], e3t=g["e3t_0"][:, :j])
        b.set_mesh(g["e1t"][j:], g["e2t"][j:], m8[:, j:], e3t=g["e3t_0"][:, j:])
        a.set_sum_mode(True), b.set_sum_mode(True)
        a.compute(np.ascontiguousarray(temp[:, :j]))
        b.chain(carry_in=a.chain(want_last=True))
        got = b.compute(np.ascontiguousarray(temp[:, j:]))
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], r4sum=True)
        assert np.array_equal(got, ref)


@pytest.mark.gpu
def test_hip_reference_order_sums_at_eorca025_size(gpu):
    """BASELINE config 2 in the reference's own arithmetic: 1442 x 1207 x 75, sequential r4 sums of 1.74e6 products per
    level and quantity, equal to the r4sum oracle bit for bit; and how far the r8 sums are from them."""
    import torch
    npi, npj, npk = 1442, 1207, 75
    g, temp, _ = fields(npi, npj, npk, seed=25)
    plan = gpu.HeatcPlan(npi, npj, npk)
    plan.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e3t=g["e3t_0"])
    t = torch.from_numpy(temp).cuda()
    r8 = plan.compute(t)
    plan.set_sum_mode(True)
    got = plan.compute(t)
    ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], r4sum=True)
    assert np.array_equal(got, ref)
    rel = np.abs(got - r8) / np.abs(r8).max(axis=0)
    assert 1e-9 < rel.max() < 1e-3                     # visible (9e-5), and what the r8 mode trades away
    sh = gpu.HeatcPlan(npi, npj, npk, devices=[0, 0, 0, 0])
    sh.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e3t=g["e3t_0"])
    sh.set_sum_mode(True)
    assert np.array_equal(sh.compute(t), ref)


def both_kernels(gpu, plan, temp, monkeypatch, **kw):
    """The level-parallel integer kernel (default) and the plain chain of v_add_f32 (CDFT_HEATC_SEQ=chain), as bits."""
    monkeypatch.delenv("CDFT_HEATC_SEQ", raising=False)
    a = plan.compute(temp, **kw)
    monkeypatch.setenv("CDFT_HEATC_SEQ", "chain")
    b = plan.compute(temp, **kw)
    monkeypatch.delenv("CDFT_HEATC_SEQ")
    return a.view(np.uint64), b.view(np.uint64)


def same_bits(a, b):
    """bit for bit, a NaN being a NaN (its sign and payload depend on which operands met: Inf - Inf here, NaN + x there)"""
    fa, fb = np.asarray(a).view(np.float64), np.asarray(b).view(np.float64)
    na, nb_ = np.isnan(fa), np.isnan(fb)
    return np.array_equal(na, nb_) and np.array_equal(np.asarray(a)[~na], np.asarray(b)[~nb_])


ADVERSARIAL = ["ones", "powers_of_two", "sign_changes", "mostly_land", "huge_range", "nonfinite", "tiny_sum", "negative_mask", "constant_rows"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ADVERSARIAL)
def test_hip_reference_order_kernels_agree_on_adversarial_fields(gpu, case, monkeypatch):
    """heatc_scan_kernel replaces a chain of sequential r4 adds by integer sums inside a binade; everything that can break
    that -- ties on every cell, sums that cross powers of two or zero over and over, a sum smaller than its terms, Inf and
    NaN, negative weights, long stretches of land -- must come out bit-identical to the plain chain and to the oracle."""
    rng = np.random.default_rng(len(case))
    npi, npj, npk = 1442, 211, 3
    shape = (npk, npj, npi)
    e1t = (2000 + 26000 * rng.random((npj, npi))).astype(f32)
    e2t = (2000 + 26000 * rng.random((npj, npi))).astype(f32)
    e3t = (1 + 199 * rng.random(shape)).astype(f32)
    mask = (rng.random(shape) > 0.35).astype(f32)
    temp = (-2 + 32 * rng.random(shape)).astype(f32)
    if case == "ones":                                   # s counts 1, 2, 3, ...: a tie on every cell once s >= 2^24 would need 16.7e6 cells;
        e1t[:], e2t[:], e3t[:], mask[:], temp[:] = 1, 1, 1, 1, 1      # here: every chunk boundary is a power of two early on
    elif case == "powers_of_two":                        # x = 2^k: x / ulp(s) is a half-integer again and again
        e1t[:] = 2.0 ** rng.integers(0, 4, (npj, npi)); e2t[:] = 1
        e3t[:] = 2.0 ** rng.integers(-3, 3, shape); temp[:] = 2.0 ** rng.integers(-2, 3, shape) * rng.choice([-1, 1], shape)
    elif case == "sign_changes":                         # the heat sum hovers around zero
        temp[:] = rng.normal(0, 1, shape)
    elif case == "mostly_land":
        mask[:] = rng.random(shape) > 0.97
        mask[:, : npj // 2] = 0                          # and the first half of every level is land: s = 0 for 150 000 cells
    elif case == "huge_range":
        temp[:] *= 10.0 ** rng.integers(-20, 20, shape)
        e3t[:] *= 10.0 ** rng.integers(-3, 3, shape)
    elif case == "nonfinite":
        temp[0, 100, 7] = np.inf; temp[1, 50, 700] = np.nan; temp[2, 200, 1000] = -np.inf; temp[2, 205, 3] = 3e38
    elif case == "tiny_sum":
        e1t[:] = 1e-18; e2t[:] = 1e-18; e3t[:] = 1e-6    # surfaces of 1e-36: denormal sums
    elif case == "negative_mask":                        # -M with an arbitrary r4 mask: the "surface" chain sees negative terms
        mask[:] = rng.normal(0, 1, shape)
    elif case == "constant_rows":                        # a regular grid: e1t constant along i, the same x thousands of times
        e1t[:] = (1000 * (1 + np.arange(npj)))[:, None]; e2t[:] = 2500; e3t[:] = 10; temp[:] = 4
    plan = gpu.HeatcPlan(npi, npj, npk)
    m = mask if case == "negative_mask" else mask.astype(np.uint8)
    plan.set_mesh(e1t, e2t, m, e3t=e3t)
    plan.set_sum_mode(True)
    a, b = both_kernels(gpu, plan, temp, monkeypatch)
    assert same_bits(a, b), case
    with np.errstate(all="ignore"):
        ref = oracle_tools.heatc(e1t, e2t, mask, temp, e3t=e3t, r4sum=True)
    assert same_bits(a, ref.view(np.uint64)), case
    # chained row blocks: the chains start from a carry in the middle of a binade
    sh = gpu.HeatcPlan(npi, npj, npk, devices=[0, 0, 0])
    sh.set_mesh(e1t, e2t, m, e3t=e3t)
    sh.set_sum_mode(True)
    assert same_bits(sh.compute(temp).view(np.uint64), a), case


@pytest.mark.gpu
def test_hip_reference_order_sum_saturates_like_the_chain(gpu, monkeypatch):
    """One level of 4322 x 3900 ones: the r4 count stops at 2^24 = 16 777 216 of 16 855 800 cells (s + 1 is a tie that goes
    back to the even s, on every cell from there on) -- in both kernels."""
    npi, npj, npk = 4322, 3900, 1
    one = np.ones((npj, npi), f32)
    plan = gpu.HeatcPlan(npi, npj, npk)
    plan.set_mesh(one, one, np.ones((npk, npj, npi), np.uint8), e3t=np.ones((npk, npj, npi), f32))
    plan.set_sum_mode(True)
    a, b = both_kernels(gpu, plan, np.ones((npk, npj, npi), f32), monkeypatch)
    assert np.array_equal(a, b) and np.all(a.view(np.float64) == 16777216.0)


@pytest.mark.gpu
def test_hip_reference_order_kernels_agree_on_eorca12_levels(gpu, monkeypatch):
    """Two levels of 4322 x 3606 = 15.6e6 cells: 119 tickets of 131072 cells per level, sums that reach 2e15 where a term is of
    the order of ulp(s) (the r4 chain then runs per cents away from the r8 sums the hypotheses start from): both kernels and
    the oracle, bit for bit; and with the chains carried over three row blocks."""
    rng = np.random.default_rng(12)
    npi, npj, npk = 4322, 3606, 2
    shape = (npk, npj, npi)
    e1t = (2000 + 26000 * rng.random((npj, npi), dtype=f32)).astype(f32)
    e2t = (2000 + 26000 * rng.random((npj, npi), dtype=f32)).astype(f32)
    e3t = (1 + 199 * rng.random(shape, dtype=f32)).astype(f32)
    mask = (rng.random(shape, dtype=f32) > 0.35).astype(np.uint8)
    temp = (-2 + 32 * rng.random(shape, dtype=f32)).astype(f32)
    plan = gpu.HeatcPlan(npi, npj, npk)
    plan.set_mesh(e1t, e2t, mask, e3t=e3t)
    r8 = plan.compute(temp)
    plan.set_sum_mode(True)
    a, b = both_kernels(gpu, plan, temp, monkeypatch)
    assert np.array_equal(a, b)
    ref = oracle_tools.heatc(e1t, e2t, mask, temp, e3t=e3t, r4sum=True)
    assert np.array_equal(a, ref.view(np.uint64))
    # the tickets dealt from one counter, from three lanes (a lane count the XCDs do not divide into: tickets get stolen) and
    # with no cell read in full by the sampling pass (the first hypotheses all wrong): the same bits
    for var, val in (("CDFT_HEATC_SCAN_LANES", "1"), ("CDFT_HEATC_SCAN_LANES", "3"), ("CDFT_HEATC_SCAN_DENSE", "0")):
        monkeypatch.setenv(var, val)
        assert np.array_equal(plan.compute(temp).view(np.uint64), a), (var, val)
        monkeypatch.delenv(var)
    assert (np.abs(ref - r8) / np.abs(r8)).max() > 1e-3            # the reference's own r4 SUM is this far from the sum here
    sh = gpu.HeatcPlan(npi, npj, npk, devices=[0, 0, 0])
    sh.set_mesh(e1t, e2t, mask, e3t=e3t)
    sh.set_sum_mode(True)
    assert np.array_equal(sh.compute(temp).view(np.uint64), a)


@pytest.mark.gpu
def test_hip_reference_order_scan_that_gives_up_hands_over_to_the_chain(gpu, monkeypatch, capfd):
    """A wait of heatc_scan_kernel that does not end (a preempted queue, a fault) raises a flag instead of hanging; the
    host then forms the sums with the plain chain and says so.  CDFT_HEATC_SCAN_WAIT_US=0: the first wait that is not
    over at once gives up."""
    npi, npj, npk = 1442, 400, 6
    g, temp, _ = fields(npi, npj, npk, seed=31)
    plan = gpu.HeatcPlan(npi, npj, npk)
    plan.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e3t=g["e3t_0"])
    plan.set_sum_mode(True)
    ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], r4sum=True)
    monkeypatch.delenv("CDFT_HEATC_SEQ", raising=False)
    capfd.readouterr()
    monkeypatch.setenv("CDFT_HEATC_SCAN_WAIT_US", "0")
    got = plan.compute(temp)
    monkeypatch.delenv("CDFT_HEATC_SCAN_WAIT_US")
    err = capfd.readouterr().err
    assert np.array_equal(got, ref)
    assert "plain chain" in err, err                   # (18 tickets of 6 levels: some workgroup gets there before its predecessor)
    assert np.array_equal(plan.compute(temp), ref)     # and the plan is as good as before
    assert "plain chain" not in capfd.readouterr().err


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"CDFT_HEATC_SCAN_WALKERS": "0"}, {"CDFT_HEATC_SCAN_RING": "1"}, {"CDFT_HEATC_SCAN_RING": "2"},
                                 {"CDFT_HEATC_SCAN_RING": "5", "CDFT_HEATC_SCAN_CHUNKS": "64"}, {}],
                         ids=["no_walkers", "ring1", "ring2", "ring5_small_tickets", "default"])
def test_hip_reference_order_scan_with_and_without_walkers(gpu, monkeypatch, env):
    """Round 6: the chains of a level are walked by a workgroup of their own from sums the streaming workgroups park in a ring
    of slots per level.  The round-5 form (every workgroup walks its own ticket: CDFT_HEATC_SCAN_WALKERS=0), rings of one,
    two and five slots (a streaming workgroup then waits for its slot to be read) and the default ring: the plain chain's bits,
    twice in a row on one plan."""
    npi, npj, npk = 1442, 400, 9
    g, temp, _ = fields(npi, npj, npk, seed=77)
    plan = gpu.HeatcPlan(npi, npj, npk)
    plan.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e3t=g["e3t_0"])
    plan.set_sum_mode(True)
    monkeypatch.setenv("CDFT_HEATC_SEQ", "chain")
    want = plan.compute(temp).view(np.uint64)
    monkeypatch.delenv("CDFT_HEATC_SEQ")
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    for _ in range(2):
        assert np.array_equal(plan.compute(temp).view(np.uint64), want)
    assert np.array_equal(want.view(np.float64), oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], r4sum=True))


FUZZ_SHAPES = [(1, 1, 1), (511, 1, 2), (512, 1, 1), (513, 1, 3), (4, 127, 2), (4096, 8, 1), (4097, 8, 2), (257, 255, 3), (1023, 33, 2),
               (32768, 1, 2), (32769, 1, 1), (181, 181, 5), (65536, 2, 1), (1442, 91, 2), (977, 269, 4), (3, 3, 7)]


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk", FUZZ_SHAPES)
def test_hip_reference_order_kernels_agree_on_random_shapes_and_options(gpu, npi, npj, npk, monkeypatch):
    """Sizes around the chunk (512), block (4096) and ticket (32768) boundaries, every combination of byte / r4 mask, e3t /
    -full, -mxloption 0 / 1 / -1, magnitudes from 1e-12 to 1e12, and a start value handed in (cdft_heatc_chain)."""
    rng = np.random.default_rng(npi * 7 + npj * 3 + npk)
    shape = (npk, npj, npi)
    for trial in range(4):
        scale = 10.0 ** rng.integers(-6, 7)
        e1t = (scale * (0.5 + rng.random((npj, npi)))).astype(f32)
        e2t = (scale * (0.5 + rng.random((npj, npi)))).astype(f32)
        e3t = (1 + 199 * rng.random(shape)).astype(f32)
        e31d = (1 + 199 * rng.random(npk)).astype(f32)
        mask = (rng.random(shape) > rng.choice([0.1, 0.5, 0.95])).astype(f32)
        temp = (rng.normal(rng.choice([0.0, 10.0]), 8, shape)).astype(f32)
        mxl = (10 + 300 * rng.random((npj, npi))).astype(f32)
        gdepw = np.cumsum(np.concatenate([[0], e31d[:-1]])).astype(f32)
        full, m8, opt = bool(trial & 1), bool(trial & 2), int(rng.integers(-1, 2))
        plan = gpu.HeatcPlan(npi, npj, npk)
        plan.set_mesh(e1t, e2t, mask.astype(np.uint8) if m8 else (mask * f32(0.75)).astype(f32), **(dict(e31d=e31d) if full else dict(e3t=e3t)))
        plan.set_sum_mode(True)
        kw = dict(rmxldep=mxl if opt else None, gdepw=gdepw, mxloption=opt)
        carry = (rng.normal(0, 1, (npk, 3)) * scale * scale * rng.choice([0.0, 1.0, 1e3])).astype(f32)
        monkeypatch.delenv("CDFT_HEATC_SEQ", raising=False)
        plan.chain(carry_in=carry)
        a = plan.compute(temp, **kw).view(np.uint64)
        monkeypatch.setenv("CDFT_HEATC_SEQ", "chain")
        plan.chain(carry_in=carry)
        b = plan.compute(temp, **kw).view(np.uint64)
        monkeypatch.delenv("CDFT_HEATC_SEQ")
        assert np.array_equal(a, b), (trial, full, m8, opt)


@pytest.mark.gpu
def test_hip_reference_order_kernel_next_to_another_kernel_on_another_stream(gpu):
    """tools/stress_scan_concurrent.py for two seconds: heatc_scan_kernel (whose workgroups wait for earlier tickets) on one
    stream, cdfmoc's zonal kernel filling the CUs on another, from two host threads -- same sums every time, no hang."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_scan_concurrent.py")], capture_output=True, text=True,
                       env=dict(os.environ, STRESS_SECONDS="2"), timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1500:] + r.stderr[-1500:]


def abs_terms(g, temp, e3t=None, e31d=None, **kw):
    return oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], np.abs(temp), e3t=e3t, e31d=e31d, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("npi,npj,npk", [(24, 18, 7), (182, 149, 31), (1, 1, 1), (3, 5, 2), (33, 7, 4),
                                         (128, 64, 3), (1442, 33, 2)])
def test_hip_matches_oracle(gpu, npi, npj, npk):
    g, temp, _ = fields(npi, npj, npk, seed=npi)
    plan = gpu.HeatcPlan(npi, npj, npk)
    plan.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e3t=g["e3t_0"])
    got = plan.compute(temp)
    ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"])
    assert np.all(np.abs(got - ref) <= 1e-13 * abs_terms(g, temp, e3t=g["e3t_0"]) + 1e-300)
    ref4 = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"], r4sum=True)
    scale = np.abs(abs_terms(g, temp, e3t=g["e3t_0"])).max(axis=0) + 1e-300
    assert np.all(np.abs(got - ref4) / scale <= 1e-3)


@pytest.mark.gpu
def test_hip_options_full_step_mxl_and_r4_mask(gpu):
    import torch

    npi, npj, npk = 60, 21, 6
    g, temp, mxl = fields(npi, npj, npk, seed=9)
    plan = gpu.HeatcPlan(npi, npj, npk)
    # -full: e31d(k) replaces e3t
    plan.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e31d=g["e3t_1d"])
    ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e31d=g["e3t_1d"])
    lim = 1e-13 * abs_terms(g, temp, e31d=g["e3t_1d"]) + 1e-300
    assert np.all(np.abs(plan.compute(temp) - ref) <= lim)
    # -mxloption +1 / -1, arrays resident in HBM, r4 mask with non-0/1 values (-M)
    fm = (g["tmask"].astype(np.float32) * np.float32(0.5))
    dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (g["e1t"], g["e2t"], fm, g["e3t_0"])]
    plan.set_mesh(dev[0], dev[1], dev[2], e3t=dev[3])
    g2 = dict(g, tmask=fm)
    for opt in (1, -1):
        got = plan.compute(torch.from_numpy(temp).cuda(), rmxldep=torch.from_numpy(mxl).cuda(),
                           gdepw=g["gdepw_1d"], mxloption=opt)
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], fm, temp, e3t=g["e3t_0"], mxl=mxl, gdepw=g["gdepw_1d"],
                                 mxlopt=opt)
        lim = 1e-13 * abs_terms(g2, temp, e3t=g["e3t_0"], mxl=mxl, gdepw=g["gdepw_1d"], mxlopt=opt) + 1e-300
        assert np.all(np.abs(got - ref) <= lim)
    with pytest.raises(gpu.CdftError):
        plan.compute(temp, mxloption=1)                # rmxldep missing


@pytest.mark.gpu
def test_hip_properties_at_eorca025_size(gpu):
    """BASELINE config 1: cdfheatc on eORCA025 (1442x1207x75), one record, 1 GPU."""
    import torch

    npi, npj, npk = 1442, 1207, 75
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(11)
    e1t = 2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen)
    e2t = 2000.0 + 26000.0 * torch.rand((npj, npi), device=dev, generator=gen)
    mb = torch.where(torch.rand((npj, npi), device=dev, generator=gen) < 0.35, 0.0,
                     1.0 + torch.floor((npk - 1) * torch.rand((npj, npi), device=dev, generator=gen)))
    k = torch.arange(npk, device=dev, dtype=torch.float32)[:, None, None]
    tmask = (k < mb[None]).to(torch.uint8)
    e3t = torch.from_numpy(synth.e3_profile(npk)).to(dev)[:, None, None] * (0.2 + 0.8 * torch.rand((npk, npj, npi), device=dev, generator=gen))
    temp = -2.0 + 32.0 * torch.rand((npk, npj, npi), device=dev, generator=gen)
    plan = gpu.HeatcPlan(npi, npj, npk)
    plan.set_mesh(e1t, e2t, tmask, e3t=e3t.contiguous())
    a = plan.compute(temp)
    assert np.array_equal(a, plan.compute(temp))                          # deterministic
    b = plan.compute(temp * 2.0)                                          # exact power-of-two scaling
    assert np.array_equal(b[:, 2], 2.0 * a[:, 2]) and np.array_equal(b[:, :2], a[:, :2])
    # surface/volume do not depend on temperature; temperature == 1 => dsum2d == dvol2d
    c = plan.compute(torch.ones_like(temp))
    assert np.array_equal(c[:, 2], c[:, 1])
    # three levels against the oracle
    for lev in (0, 40, 74):
        ref = oracle_tools.heatc(e1t.cpu().numpy(), e2t.cpu().numpy(), tmask[lev:lev + 1].cpu().numpy(),
                                 temp[lev:lev + 1].cpu().numpy(), e3t=e3t[lev:lev + 1].cpu().numpy())
        assert np.all(np.abs(a[lev] - ref[0]) <= 1e-13 * np.abs(ref[0]) * 40 + 1e-300)
