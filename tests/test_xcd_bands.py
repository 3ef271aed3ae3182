"""The XCD band mappings (csrc/common.hpp xcd_band, csrc/moc_zonal.hpp, csrc/zonal.hip): a permutation of which workgroup
does which piece of the work, for the L2s' sake -- the results must not know.  Every kernel that has one, on a grid where it
is active (131 rows: more than the 64 the row maps ask for, not a multiple of 8; workgroup counts that are no multiple of
8 either), with the hooks CDFT_XCD_BANDS / CDFT_MOC_XCDMAP / CDFT_ZONAL_XCDMAP off and on: the same bits, and the oracle's."""
import numpy as np
import pytest

import oracle_lib
import oracle_tools
import synth

f32 = np.float32
NPI, NPJ, NPK = 1301, 131, 9


@pytest.fixture(scope="module")
def gpu():
    from cdftools_amd import hip
    return hip


def both(monkeypatch, fn, hooks=("CDFT_XCD_BANDS",)):
    out = []
    for val in ("0", "1"):
        for h in hooks:
            monkeypatch.setenv(h, val)
        out.append(fn())
    for h in hooks:
        monkeypatch.delenv(h)
    return out


@pytest.mark.gpu
def test_moc_zonal_kernel_rows_in_bands(gpu, monkeypatch):
    g = synth.make_grid(NPI, NPJ, NPK, seed=5, nbasins=5)
    e1v, e3v, ib = synth.moc_inputs(g)
    zv = synth.velocity(g, 3, 6)
    plan = gpu.MocPlan(NPI, NPJ, NPK, 5)
    plan.set_mesh(e1v, e3v, ib)
    for hook in ("CDFT_XCD_BANDS", "CDFT_MOC_XCDMAP"):
        off, on = both(monkeypatch, lambda: plan.compute(zv), (hook,))
        assert np.array_equal(off, on), hook
    assert "xcd_bands=1" in plan.describe()
    ref = oracle_lib.moc(e1v, e3v, ib, zv)
    bound = -oracle_lib.moc(e1v, e3v, ib, np.abs(zv))
    assert np.all(np.abs(on - ref) <= 1e-13 * bound + 1e-300)


@pytest.mark.gpu
def test_level_and_row_kernels_in_bands(gpu, monkeypatch):
    g = synth.make_grid(NPI, NPJ, NPK, seed=7, nbasins=5)
    rng = np.random.default_rng(8)
    temp = ((-2 + 32 * rng.random((NPK, NPJ, NPI), dtype=f32)) * g["tmask"]).astype(f32)
    # cdfheatc (heatc_level_kernel / _u4_: 1301 * 131 is odd, the unaligned kernel)
    hp = gpu.HeatcPlan(NPI, NPJ, NPK)
    hp.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e3t=g["e3t_0"])
    off, on = both(monkeypatch, lambda: hp.compute(temp))
    assert np.array_equal(off, on)
    ref = oracle_tools.heatc(g["e1t"], g["e2t"], g["tmask"], temp, e3t=g["e3t_0"])
    assert np.all(np.abs(on - ref) <= 1e-13 * np.abs(ref) + 1e-300)
    # cdfmean (mean_level_kernel)
    mp = gpu.MeanPlan(NPI, NPJ, NPK, 1)
    mp.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(f32), np.ones((NPJ, NPI, 1), np.int16), e3=g["e3t_0"])
    off, on = both(monkeypatch, lambda: np.array(mp.compute(temp)))
    assert np.array_equal(off, on)
    # cdfmhst -vt and cdfpsi (vint_kernel)
    vt = ((rng.random((NPK, NPJ, NPI), dtype=f32) - f32(0.5)) * g["vmask"]).astype(f32)
    masks = np.stack([g["tmask"][0]] * 5).astype(f32)
    hs = gpu.MhstPlan(NPI, NPJ, NPK, 5)
    hs.set_mesh(g["e1v"], masks, e3v=g["e3v_0"])
    off, on = both(monkeypatch, lambda: [np.array(x) for x in hs.compute(vt=vt, vs=vt)])
    assert all(np.array_equal(a, b) for a, b in zip(off, on))
    pp = gpu.PsiPlan(NPI, NPJ, NPK)
    off, on = both(monkeypatch, lambda: np.array(pp.compute(zu=vt, e2u=g["e2u"], e3u=g["e3u_0"])[0]))
    assert np.array_equal(off, on)
    # cdfzonalsum / cdfzonalmean (zonal_row_class_kernel)
    zm = np.stack([g["tmask"][0]] * 5, axis=2).astype(f32)
    zp = gpu.ZonalPlan(NPI, NPJ, NPK, 5)
    zp.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(f32), zm)
    for mean in (False, True):
        off, on = both(monkeypatch, lambda: np.array(zp.compute(temp, mean_form=mean)["sum"]), ("CDFT_XCD_BANDS", "CDFT_ZONAL_XCDMAP"))
        assert np.array_equal(off, on), mean
