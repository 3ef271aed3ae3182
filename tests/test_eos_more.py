"""The rest of the eos module's public list (src/eos.f90:24-56): spice, alpha, beta, albet, eosbn2 -- every member of
each generic.  The oracle (oracle/cdf_oracle_eos2.c) is PINNED here against
(a) the check value the source states for spiciness (src/eos.f90:700; good to 1e-7, see the test),
(b) tests/golden/eos_more.npz: outputs of the unmodified eos.f90 built by oracle/ref_build.sh (tests/make_golden_eos_more.py),
(c) oracle/_ref itself on fresh random points when present (build container).
GPU: the kernels of cdftools_amd/csrc/eos_more.hip equal the oracle and the golden BIT FOR BIT (r8 results; r4 for
eosbn2), through the C ABI and through the Fortran module's generics."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from make_golden_eos_more import POINTS, inputs

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "eos_more.npz"))
f32 = np.float32
COEF = [(0, "alpha"), (1, "beta"), (2, "albet"), (3, "beta_mcd87")]
SETS = [("eos80", False), ("teos10", True)]


def gold_inputs():
    return [GOLD[k] for k in ("T", "S", "D", "T2", "S2", "E")]


# ----------------------------------------------------------------------------- CPU: pin the oracle
def test_the_golden_inputs_are_the_seeded_ones():
    for a, b in zip(inputs(), gold_inputs()):
        assert np.array_equal(a, b)


def test_spiciness_check_value_of_the_source():
    # src/eos.f90:700-701: spice(15,33) = 0.544586321373410 "calcul en double", 0.5445864 in single precision (Flament
    # 2002).  The module as it stands today gives 0.5445864137500002 (compiled reference and oracle alike, bitwise): the
    # comment predates the coefficients' present spelling, so it is a 1e-7 check, the golden file is the exact one.
    got = ol.spice(f32([15.0]), f32([33.0]), flament=True)[0]
    assert abs(got - 0.544586321373410) < 1e-7 and got == 0.5445864137500002
    assert GOLD["spice_f02"][0, 0] == ol.spice(f32([15.0]), f32([33.0]), flament=True)[0]


def test_oracle_spice_equals_golden_reference_outputs_bitwise():
    T, S = GOLD["T"], GOLD["S"]
    assert np.array_equal(ol.spice(T, S), GOLD["spice_pol"]) and np.array_equal(ol.spice(T, S, flament=True), GOLD["spice_f02"])


@pytest.mark.parametrize("tag,teos", SETS)
def test_oracle_alpha_beta_albet_equal_golden_reference_outputs_bitwise(tag, teos):
    T, S, D = GOLD["T"], GOLD["S"], GOLD["D"]
    for which, nm in COEF:
        for dep in (0.0, 1234.5):
            assert np.array_equal(ol.eos_coef(which, T, S, dep, teos), GOLD[f"{nm}_{dep:g}_{tag}"]), (nm, dep)
        assert np.array_equal(ol.eos_coef(which, T, S, 0.0, teos, pdep2d=D), GOLD[f"{nm}_z_{tag}"]), nm
    # alpha and beta are masked on S == 0 (src/eos.f90:1419-1423), the McDougall forms are not
    land = S == 0
    assert land.any() and np.all(GOLD[f"alpha_0_{tag}"][land] == 0) and np.all(GOLD[f"beta_0_{tag}"][land] == 0)
    assert np.all(GOLD[f"albet_0_{tag}"][land] != 0) and np.all(GOLD[f"beta_mcd87_0_{tag}"][land] != 0)
    # McDougall 1987 does not depend on the coefficient set
    assert np.array_equal(GOLD["albet_z_eos80"], GOLD["albet_z_teos10"]) and np.array_equal(GOLD["beta_mcd87_z_eos80"], GOLD["beta_mcd87_z_teos10"])


@pytest.mark.parametrize("tag,teos", SETS)
def test_oracle_bn2_equals_golden_reference_outputs_bitwise(tag, teos):
    T, S, D, T2, S2, E = gold_inputs()
    for mcd, nm in ((False, "pol"), (True, "mcd87")):
        got = ol.bn2(T, T2, S, S2, E, 777.0, teos, mcd)
        assert got.dtype == f32
        for up in (1, 2):                                      # which plane of (kpi,kpj,2) holds the upper level does not matter
            assert np.array_equal(got, GOLD[f"bn2_{nm}_up{up}_{tag}"]), (nm, up)
    # the quirk kept from src/eos.f90:1231-1235: alpha_0d / beta_0d get the anomaly s-35, so a mean salinity of exactly
    # 35 is what they mask on, not 0
    assert GOLD[f"bn2_pol_up1_{tag}"][1, 0] == 0.0 and S[1, 0] != 0 and 0.5 * (S[1, 0] + S2[1, 0]) == 35.0


@pytest.mark.skipif(not os.path.exists(ol.REF_EOS), reason="oracle/_ref is built in the build container only")
def test_oracle_equals_the_compiled_reference_on_fresh_points():
    ref = ol.RefEos()
    rng = np.random.default_rng(99)
    shape = (41, 67)
    T = (-3 + 36 * rng.random(shape)).astype(f32)
    S = (-2 + 44 * rng.random(shape)).astype(f32)
    S[rng.random(shape) < 0.1] = 0
    D = (6000 * rng.random(shape)).astype(f32)
    T2, S2 = (T + rng.normal(0, 2, shape)).astype(f32), (S + rng.normal(0, 1, shape)).astype(f32)
    E = (0.5 + 300 * rng.random(shape)).astype(f32)
    for fl in (False, True):
        assert np.array_equal(ol.spice(T, S, fl), ref.spice(T, S, fl))
    for teos in (False, True):
        for which, nm in COEF:
            assert np.array_equal(ol.eos_coef(which, T, S, 432.1, teos), ref.coef(which, T, S, 432.1, teos)), nm
            assert np.array_equal(ol.eos_coef(which, T, S, 0, teos, pdep2d=D), ref.coef(which, T, S, D, teos)), nm
        for mcd in (False, True):
            assert np.array_equal(ol.bn2(T, T2, S, S2, E, 250.0, teos, mcd), ref.bn2(T, T2, S, S2, E, 250.0, teos, mcd))
        for w in (0, 1):
            for p in POINTS:                                  # the r8 one-point members against the array oracle on r4-exact points
                got = ol.eos_coef(w, f32([p[0]]), f32([p[1]]), p[2], teos)[0]
                assert got == ref.coef0d(w, *p, teos10=teos)


# ----------------------------------------------------------------------------- GPU: the kernels
@pytest.mark.gpu
def test_hip_spice_is_bit_identical(gpu):
    import torch
    T, S = GOLD["T"], GOLD["S"]
    assert np.array_equal(gpu.spice(T, S), GOLD["spice_pol"]) and np.array_equal(gpu.spice(T, S, flament=True), GOLD["spice_f02"])
    got = gpu.spice(torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda())
    assert got.dtype == torch.float64 and np.array_equal(got.cpu().numpy(), GOLD["spice_pol"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,teos", SETS)
def test_hip_alpha_beta_albet_are_bit_identical(gpu, tag, teos):
    import torch
    T, S, D = GOLD["T"], GOLD["S"], GOLD["D"]
    eos = gpu.EOS_TEOS10 if teos else gpu.EOS_80
    fn = {"alpha": lambda *a, **k: gpu.alpha(*a, eos=eos, **k), "beta": lambda *a, **k: gpu.beta(*a, eos=eos, **k),
          "albet": gpu.albet, "beta_mcd87": lambda *a, **k: gpu.beta(*a, eos=gpu.EOS_MCD87, **k)}
    for _, nm in COEF:
        for dep in (0.0, 1234.5):
            assert np.array_equal(fn[nm](T, S, dep), GOLD[f"{nm}_{dep:g}_{tag}"]), (nm, dep)
        assert np.array_equal(fn[nm](T, S, D), GOLD[f"{nm}_z_{tag}"]), nm
        # device-resident arrays, an odd length and an address that is not 16-byte aligned (the scalar kernel)
        t, s, d = (torch.from_numpy(np.concatenate([[0], x.ravel()]).astype(f32)).cuda()[1:] for x in (T, S, D))
        assert np.array_equal(fn[nm](t, s, d).cpu().numpy(), GOLD[f"{nm}_z_{tag}"].ravel()), nm
    pts = np.array([[gpu.eos_point(w, *p, eos=eos) for p in POINTS] for w in (gpu.EOSQ_ALPHA, gpu.EOSQ_BETA)])
    assert np.array_equal(pts, GOLD[f"points_{tag}"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,teos", SETS)
def test_hip_bn2_is_bit_identical(gpu, tag, teos):
    import torch
    T, S, D, T2, S2, E = gold_inputs()
    for eos, nm in ((gpu.EOS_TEOS10 if teos else gpu.EOS_80, "pol"), (gpu.EOS_MCD87, "mcd87")):
        got = gpu.eosbn2(T, T2, S, S2, E, 777.0, eos=eos)
        assert got.dtype == f32 and np.array_equal(got, GOLD[f"bn2_{nm}_up1_{tag}"]), nm
        dev = [torch.from_numpy(x).cuda() for x in (T, T2, S, S2, E)]
        assert np.array_equal(gpu.eosbn2(*dev, 777.0, eos=eos).cpu().numpy(), GOLD[f"bn2_{nm}_up1_{tag}"])


@pytest.mark.gpu
def test_hip_eos_more_against_the_oracle_at_a_level_of_eORCA025(gpu):
    """One 1442 x 1207 level of seeded fields: every quantity equals the oracle bit for bit (both memory spaces)."""
    import torch
    rng = np.random.default_rng(5)
    n = 1442 * 1207
    T = (-2 + 32 * rng.random(n)).astype(f32)
    S = (30 + 8 * rng.random(n)).astype(f32)
    S[rng.random(n) < 0.3] = 0
    D = (5800 * rng.random(n)).astype(f32)
    T2, S2, E = (T + f32(0.5)).astype(f32), (S * f32(1.01)).astype(f32), (1 + 250 * rng.random(n)).astype(f32)
    dT, dS, dD = (torch.from_numpy(x).cuda() for x in (T, S, D))
    assert np.array_equal(gpu.spice(dT, dS).cpu().numpy(), ol.spice(T, S))
    assert np.array_equal(gpu.spice(dT, dS, flament=True).cpu().numpy(), ol.spice(T, S, True))
    for teos in (False, True):
        eos = gpu.EOS_TEOS10 if teos else gpu.EOS_80
        assert np.array_equal(gpu.alpha(dT, dS, dD, eos=eos).cpu().numpy(), ol.eos_coef(0, T, S, 0, teos, pdep2d=D))
        assert np.array_equal(gpu.beta(dT, dS, 1000.0, eos=eos).cpu().numpy(), ol.eos_coef(1, T, S, 1000.0, teos))
        assert np.array_equal(gpu.eosbn2(T, T2, S, S2, E, 55.0, eos=eos), ol.bn2(T, T2, S, S2, E, 55.0, teos))
    assert np.array_equal(gpu.albet(dT, dS, 10.0).cpu().numpy(), ol.eos_coef(2, T, S, 10.0))
    assert np.array_equal(gpu.beta(dT, dS, dD, eos=gpu.EOS_MCD87).cpu().numpy(), ol.eos_coef(3, T, S, 0, pdep2d=D))
    assert np.array_equal(gpu.eosbn2(T, T2, S, S2, E, 55.0, eos=gpu.EOS_MCD87), ol.bn2(T, T2, S, S2, E, 55.0, mcd87=True))
    # device arrays: four points per lane when every address is 16-byte aligned (with a ragged tail), one point per lane when
    # one is not -- the same bits either way
    dev = [torch.from_numpy(x).cuda() for x in (T, T2, S, S2, E)]
    for lo, hi in ((0, n - 3), (1, n), (1, 4099), (0, 3)):
        want = ol.bn2(*(x[lo:hi] for x in (T, T2, S, S2, E)), 55.0, False)
        assert np.array_equal(gpu.eosbn2(*(x[lo:hi] for x in dev), 55.0).cpu().numpy(), want), (lo, hi)


@pytest.mark.gpu
def test_hip_eos_more_argument_errors(gpu):
    T, S = GOLD["T"], GOLD["S"]
    with pytest.raises(gpu.CdftError):
        gpu.eos_field(9, T, S)
    with pytest.raises(gpu.CdftError):
        gpu.alpha(T, S, eos=gpu.EOS_JMD94)
    with pytest.raises(gpu.CdftError):
        gpu.eosbn2(T, T, S, S, T, 1.0, eos=gpu.EOS_JMD94)
    with pytest.raises(gpu.CdftError):
        gpu.eos_point(gpu.EOSQ_SPICE_POL, 1.0, 35.0, 0.0)
    with pytest.raises(ValueError):
        gpu.alpha(T, S[:3])
    e = np.empty(0, f32)
    assert gpu.alpha(e, e).shape == (0,) and gpu.eosbn2(e, e, e, e, e, 0.0).shape == (0,)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("amdflang") is None, reason="needs amdflang")
@pytest.mark.parametrize("tag,teos", SETS)
def test_fortran_eos_module_generics_on_the_gpu(gpu, tmp_path, tag, teos):
    """tests/fortran/eos_public.f90 calls every member of every generic through `USE eos`; results equal the golden."""
    from test_cdfio_public import BUILD, FFLAGS, link_args
    d = str(tmp_path)
    exe = os.path.join(d, "eos_public")
    r = subprocess.run(["amdflang", *FFLAGS, f"-I{BUILD}", "-module-dir", d, os.path.join(HERE, "fortran", "eos_public.f90"), "-o", exe, *link_args()],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    T, S, D, T2, S2, E = gold_inputs()
    npj, npi = T.shape
    with open(os.path.join(d, "in.bin"), "wb") as f:
        f.write(np.array([npi, npj], np.int32).tobytes())
        for a in (T, S, D, T2, S2, E):
            f.write(a.tobytes())
        f.write(np.array(POINTS, np.float64).tobytes())       # (4,3) C order = Fortran dpt(3,4)
    r = subprocess.run([exe, d, "1" if teos else "0"], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0 and "done" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    out = np.fromfile(os.path.join(d, "out.bin"), np.float64)
    pos = [0]

    def take(shape):
        n = int(np.prod(shape))
        a = out[pos[0]:pos[0] + n].reshape(shape)
        pos[0] += n
        return a

    gs = np.load(os.path.join(HERE, "golden", "eos.npz"))
    full = (npj, npi)
    assert np.array_equal(take(full), ol.sigmantr(T, S))
    assert np.array_equal(take(full), GOLD["spice_pol"]) and np.array_equal(take(full), GOLD["spice_f02"])
    for nm in ("alpha", "beta", "albet", "beta_mcd87"):
        assert np.array_equal(take(full), GOLD[f"{nm}_1234.5_{tag}"]), nm
    for nm in ("alpha", "beta", "albet", "beta_mcd87"):
        assert np.array_equal(take((npi,)), GOLD[f"{nm}_z_{tag}"][0]), nm
    pts = take((4, 2))
    assert np.array_equal(pts.T, GOLD[f"points_{tag}"])
    assert np.array_equal(take(full).astype(f32), GOLD[f"bn2_pol_up1_{tag}"])
    assert np.array_equal(take(full).astype(f32), GOLD[f"bn2_mcd87_up1_{tag}"])
    assert np.array_equal(take(full).astype(f32), GOLD[f"bn2_pol_up2_{tag}"])
    assert pos[0] == out.size and gs is not None
