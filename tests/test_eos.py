"""sigma0 / sigmai (src/eos.f90): the oracle is PINNED against the real reference here --
(a) the reference's own known-answer values (src/eos.f90:817,:820),
(b) tests/golden/eos.npz, outputs of the unmodified eos.f90 built by oracle/ref_build.sh
    (generator: tests/make_golden_eos.py),
(c) oracle/_ref itself when present (build container).
GPU: the HIP kernels must equal the oracle BIT FOR BIT in the exact mode; the FMA-contracted mode
is held to 16 ulp of r8 at the magnitude of rho (and equality after the r4 store of the tool form to 1 ulp r4).
"""
import os

import numpy as np
import pytest

import oracle_lib

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "eos.npz"))
REF_PROFILE_03 = 13.486253711156655      # ref(h = 0.3): R00..R05 polynomial at 3000 m


# ----------------------------------------------------------------------------- CPU: pin the oracle
def test_known_answers_of_the_reference():
    t = np.array([3.0], np.float32)
    s = np.array([35.5], np.float32)
    # src/eos.f90:820  EOS-80 : rho = 1028.35011066567 at z=3000, pt=3, sp=35.5
    # src/eos.f90:817  TEOS-10: rho = 1028.21993233072 at z=3000, ct=3, sa=35.5
    # (in-situ density; sigmai returns rho + ref(h) - 1000)
    for teos, rho in ((False, 1028.35011066567), (True, 1028.21993233072)):
        sig = oracle_lib.sigmai(t, s, 3000.0, teos)[0]
        assert abs(sig + 1000.0 - REF_PROFILE_03 - rho) < 1e-10
    # values measured with the flang-built reference
    assert oracle_lib.sigmai(t, s, 3000.0, False)[0] == 41.83636437682958
    assert oracle_lib.sigmai(t, s, 3000.0, True)[0] == 41.706186041879164
    assert oracle_lib.sigmai(t, s, 0.0, False)[0] == 28.284481618400832
    assert oracle_lib.sigmai_jmd94(t, s, 3000.0)[0] == 41.83326188211663


@pytest.mark.parametrize("tag,teos", [("eos80", False), ("teos10", True)])
def test_oracle_equals_golden_reference_outputs_bitwise(tag, teos):
    T, S, D = GOLD["T"], GOLD["S"], GOLD["D"]
    assert np.array_equal(oracle_lib.sigmai(T, S, 0.0, teos), GOLD[f"sigma0_{tag}"])
    assert np.array_equal(oracle_lib.sigmai(T, S, 2000.0, teos), GOLD[f"sigmai2000_{tag}"])
    assert np.array_equal(oracle_lib.sigmai(T, S, 3000.0, teos), GOLD[f"sigmai3000_{tag}"])
    assert np.array_equal(oracle_lib.sigmai(T, S, 0.0, teos, pref2d=D), GOLD[f"sigmai2d_{tag}"])


def test_oracle_jmd94_equals_golden_reference_outputs_bitwise():
    T, S, D = GOLD["T"], GOLD["S"], GOLD["D"]
    assert np.array_equal(oracle_lib.sigmai_jmd94(T, S, 2000.0), GOLD["sigmai2000_jmd94"])
    assert np.array_equal(oracle_lib.sigmai_jmd94(T, S, 0.0, pref2d=D), GOLD["sigmai2d_jmd94"])


def test_neutral_density_known_answer_and_golden_reference_outputs_bitwise():
    # src/eos.f90:646  rho(20,35) = 1024.59416751197 kg/m3; sigmantr returns rho - 1000 and does not mask
    assert abs(oracle_lib.sigmantr(np.float32([20.0]), np.float32([35.0]))[0] - 24.59416751197) < 1e-11
    T, S = GOLD["T"], GOLD["S"]
    got = oracle_lib.sigmantr(T, S)
    assert np.array_equal(got, GOLD["sigmantr"]) and np.all(got[S == 0] != 0.0)


def test_land_points_are_masked_on_salinity_zero():
    T, S = GOLD["T"], GOLD["S"]
    sig = oracle_lib.sigmai(T, S, 1000.0)
    assert np.all(sig[S == 0] == 0.0) and np.all(sig[S != 0] != 0.0)


@pytest.mark.skipif(not oracle_lib.ref_eos_available(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_equals_live_reference_on_fresh_random_fields():
    ref = oracle_lib.RefEos()
    rng = np.random.default_rng(5)
    T = (-2 + 34 * rng.random((331, 577))).astype(np.float32)          # 190 987 points
    S = (38 * rng.random((331, 577))).astype(np.float32)
    S[rng.random(S.shape) < 0.2] = 0
    D = (6000 * rng.random(S.shape)).astype(np.float32)
    for teos in (False, True):
        assert np.array_equal(ref.sigma0(T, S, teos), oracle_lib.sigmai(T, S, 0.0, teos))
        assert np.array_equal(ref.sigmai(T, S, 1234.5, teos), oracle_lib.sigmai(T, S, 1234.5, teos))
        assert np.array_equal(ref.sigmai2d(T, S, D, teos), oracle_lib.sigmai(T, S, 0.0, teos, pref2d=D))
    assert np.array_equal(ref.sigmai_jmd94(T, S, 1234.5), oracle_lib.sigmai_jmd94(T, S, 1234.5))
    assert np.array_equal(ref.sigmai2d_jmd94(T, S, D), oracle_lib.sigmai_jmd94(T, S, 0.0, pref2d=D))
    assert np.array_equal(ref.sigmantr(T, S), oracle_lib.sigmantr(T, S))


def test_tool_form_matches_its_definition():
    T, S = GOLD["T"], GOLD["S"].copy()
    spval = np.float32(0.0)
    got = oracle_lib.sig_tool(T, S, 2000.0, spval)
    want = (GOLD["sigmai2000_eos80"] * (S != spval)).astype(np.float32)
    assert np.array_equal(got, want)


# ----------------------------------------------------------------------------- GPU: parity
def fields(n, seed):
    rng = np.random.default_rng(seed)
    T = (-2 + 34 * rng.random(n)).astype(np.float32)
    S = (38 * rng.random(n)).astype(np.float32)
    S[rng.random(n) < 0.3] = 0
    D = (6000 * rng.random(n)).astype(np.float32)
    return T, S, D


@pytest.mark.gpu
def test_hip_neutral_density_equals_golden_reference_outputs_bitwise(gpu):
    T, S = GOLD["T"], GOLD["S"]
    assert np.array_equal(gpu.sigmai(T, S, 0.0, gpu.EOS_NTR), GOLD["sigmantr"])
    assert np.array_equal(gpu.sigmai(T, S, 2000.0, gpu.EOS_NTR), GOLD["sigmantr"])          # no reference depth in it
    rng = np.random.default_rng(7)
    t, s = (-2 + 32 * rng.random(100_003)).astype(np.float32), (38 * rng.random(100_003)).astype(np.float32)
    assert np.array_equal(gpu.sigmai(t, s, 0.0, gpu.EOS_NTR), oracle_lib.sigmantr(t, s))


@pytest.mark.gpu
@pytest.mark.parametrize("eos", [0, 1, 2])
def test_hip_equals_golden_reference_outputs_bitwise(gpu, eos):
    T, S, D = GOLD["T"], GOLD["S"], GOLD["D"]
    tag = {0: "eos80", 1: "teos10", 2: "jmd94"}[eos]
    assert np.array_equal(gpu.sigmai(T, S, 2000.0, eos), GOLD[f"sigmai2000_{tag}"])
    assert np.array_equal(gpu.sigmai(T, S, 0.0, eos, pref2d=D), GOLD[f"sigmai2d_{tag}"])
    if eos < 2:
        assert np.array_equal(gpu.sigma0(T, S, eos), GOLD[f"sigma0_{tag}"])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 257, 27118, 1_000_003])
@pytest.mark.parametrize("eos", [0, 1, 2])
def test_hip_equals_oracle_bitwise_host_and_device(gpu, n, eos):
    import torch

    T, S, D = fields(n, seed=n + eos)
    ora = oracle_lib.sigmai_jmd94 if eos == 2 else (lambda t, s, p, pref2d=None: oracle_lib.sigmai(t, s, p, eos == 1, pref2d))
    want = ora(T, S, 1500.0)
    want2 = ora(T, S, 0.0, pref2d=D)
    assert np.array_equal(gpu.sigmai(T, S, 1500.0, eos), want)                       # host arrays
    if n:
        dT, dS, dD = (torch.from_numpy(x).cuda() for x in (T, S, D))
        assert np.array_equal(gpu.sigmai(dT, dS, 1500.0, eos).cpu().numpy(), want)    # resident in HBM
        assert np.array_equal(gpu.sigmai(dT, dS, 0.0, eos, pref2d=dD).cpu().numpy(), want2)
        # unaligned views take the scalar variant
        assert np.array_equal(gpu.sigmai(dT[1:], dS[1:], 1500.0, eos).cpu().numpy(), want[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("eos", [0, 1])
def test_hip_tool_form_and_fast_mode(gpu, eos):
    T, S, _ = fields(182 * 149, seed=3)
    S[::7] = np.float32(9999.0)                                  # a non-zero fill value
    want = oracle_lib.sig_tool(T, S, 1000.0, 9999.0, eos == 1)
    got = gpu.sig_tool(T, S, 1000.0, 9999.0, eos)
    assert np.array_equal(got, want)                             # exact mode: bit-identical r4
    fast8 = gpu.sigmai(T, S, 1000.0, eos, flags=gpu.EOS_FAST_FMA)
    ref8 = oracle_lib.sigmai(T, S, 1000.0, eos == 1)
    ulp = np.spacing(np.abs(ref8) + 1000.0)                      # ulp of rho ~ 1e3, the summed quantity
    phys = S != np.float32(9999.0)                               # fill points are masked by the tool form
    assert np.all(np.abs(fast8 - ref8)[phys] <= 16 * ulp[phys])
    fast4 = gpu.sig_tool(T, S, 1000.0, 9999.0, eos, flags=gpu.EOS_FAST_FMA)
    assert np.all(np.abs(fast4.astype(np.float64) - want) <= np.spacing(np.abs(want)).astype(np.float64))


@pytest.mark.gpu
def test_hip_full_level_properties_eorca12_slab(gpu):
    """One eORCA12 level (15.6 M points) resident in HBM: pointwise => any permutation of the points
    permutes the result bit for bit; land (S == 0) is exactly 0."""
    import torch

    n = 4322 * 3606
    g = torch.Generator(device="cuda").manual_seed(7)
    T = -2 + 32 * torch.rand(n, device="cuda", generator=g)
    S = (30 + 8 * torch.rand(n, device="cuda", generator=g)) * (torch.rand(n, device="cuda", generator=g) > 0.35)
    a = gpu.sigmai(T, S, 2000.0)
    perm = torch.randperm(n, device="cuda", generator=g)
    b = gpu.sigmai(T[perm].contiguous(), S[perm].contiguous(), 2000.0)
    torch.cuda.synchronize()
    assert torch.equal(b, a[perm])
    assert torch.all(a[S == 0] == 0)
    idx = torch.randint(0, n, (4096,), device="cuda", generator=g)
    want = oracle_lib.sigmai(T[idx].cpu().numpy(), S[idx].cpu().numpy(), 2000.0)
    assert np.array_equal(a[idx].cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("eos,pref", [(0, 0.0), (0, 2000.0), (1, 1000.0), (2, 0.0), (3, 0.0)])
def test_hip_land_shortcut_stores_the_bits_of_the_full_evaluation(gpu, monkeypatch, eos, pref):
    """A wave that finds T = S = +0 on all its 256 points stores sigma(0,0,pref) without evaluating the polynomials.  Fields
    with land in large pieces -- and, inside the land, everything that must NOT take the shortcut: -0.0, a NaN, a fill
    value, a single ocean point, pieces shorter than a wave and pieces that end in the middle of one -- give the bits of the
    run that evaluates every point (CDFT_EOS_NOLAND=1), which are the oracle's."""
    import torch
    rng = np.random.default_rng(5)
    n = 64 * 4 * 40 + 3
    t = (-2 + 32 * rng.random(n)).astype(np.float32)
    s = (30 + 8 * rng.random(n)).astype(np.float32)
    land = np.zeros(n, bool)
    for a, b in ((0, 1024), (1500, 1500 + 256), (2048 + 13, 2048 + 13 + 700), (4096, 4096 + 255), (6000, 8200), (n - 300, n)):
        land[a:b] = True
    t[land] = 0
    s[land] = 0
    t[100] = -0.0                     # not +0: no shortcut for its wave
    t[6100] = np.nan
    s[6500] = -0.0
    t[7000] = 1.0e20                  # a fill value on land
    t[7600], s[7600] = 3.0, 35.0      # one ocean point in the middle of land
    for spval in (0.0, -999.0):
        monkeypatch.delenv("CDFT_EOS_NOLAND", raising=False)
        got = gpu.sigmai(t, s, pref, eos=eos)
        tool = gpu.sig_tool(t, s, pref=pref, spval=spval, eos=eos)
        dev = gpu.sig_tool(torch.from_numpy(t).cuda(), torch.from_numpy(s).cuda(), pref=pref, spval=spval, eos=eos).cpu().numpy()
        monkeypatch.setenv("CDFT_EOS_NOLAND", "1")
        full = gpu.sigmai(t, s, pref, eos=eos)
        tool_full = gpu.sig_tool(t, s, pref=pref, spval=spval, eos=eos)
        for x, y in ((got, full), (tool, tool_full), (dev, tool_full)):
            assert x.tobytes() == y.tobytes()          # the sign of a zero included
    if eos in (0, 1):
        ref = oracle_lib.sigmai(t, s, pref, teos10=eos == 1)
        assert np.array_equal(got, ref, equal_nan=True)
        assert np.signbit(got[5]) == np.signbit(ref[5])
