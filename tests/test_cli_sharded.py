"""The drop-in programs on several row blocks: CDFT_GPU_MAP / CDFT_GPUS (read by hip_shard in cdftools_hip.f90 through
cdft_shard_from_env) make cdfmoc, cdfheatc, cdfpsi, cdfvT and cdfmhst create sharded plans.  On this 1-GPU box the
blocks are mapped to device 0 (CDFT_GPU_MAP=0,0 and 0,0,0).  Every output file must be IDENTICAL to the one-device run (bytes of every
variable; the whole file where the command line is not recorded in it) -- cdfheatc included, whose reference-order r4
sums are chained through the blocks; only its optional r8 sums (CDFT_HEATC_SUM=r8) regroup their additions: 1e-13."""
import filecmp
import os

import numpy as np
import pytest

import nemo_files
from test_cli import read_nc, run, tools  # noqa: F401  (tools is a fixture)


@pytest.fixture(scope="module")
def case(tools, tmp_path_factory):  # noqa: F811
    d = tmp_path_factory.mktemp("nemo_sharded")
    g, rec = nemo_files.make_case(str(d), npi=26, npj=19, npk=6, npt=2)
    return d, g, rec


def same_file(a, b):
    """byte-identical, or -- for the programs that record their command line (with its -o name) in the global `history`
    attribute -- identical in every variable (bytes), attribute and dimension"""
    if filecmp.cmp(a, b, shallow=False):
        return True
    va, aa, da = read_nc(a)
    vb, ab, db = read_nc(b)
    if da != db or list(va) != list(vb) or aa.keys() != ab.keys():
        return False
    for k in va:
        if va[k].dtype != vb[k].dtype or va[k].tobytes() != vb[k].tobytes():
            return False
        if {x: np.asarray(y).tobytes() for x, y in aa[k].items()} != {x: np.asarray(y).tobytes() for x, y in ab[k].items()}:
            return False
    return True


def pair(tool, args, d, out, gmap, same_stdout=True):
    """run `tool args -o <out>` on one device and on the row blocks of `gmap`; -> the two output paths"""
    a, b = f"{out}_1.nc", f"{out}_n.nc"
    rc, o1, e1 = run(tool, args + ["-o", a], d)
    assert rc == 0, o1 + e1
    rc, o2, e2 = run(tool, args + ["-o", b], d, env={"CDFT_GPU_MAP": gmap})
    assert rc == 0, o2 + e2
    if same_stdout:
        assert o1.replace(a, "X") == o2.replace(b, "X")                  # the same standard output
    return os.path.join(d, a), os.path.join(d, b)


@pytest.mark.gpu
@pytest.mark.parametrize("gmap", ["0,0", "0,0,0"])
def test_programs_on_row_blocks_write_identical_files(case, gmap):
    d, g, rec = case
    uv = ["-u", "SYN_y0001_gridU.nc", "-v", "SYN_y0001_gridV.nc"]
    for tool, args, out in (("cdfmoc", ["-v", "SYN_y0001_gridV.nc"], "moc"),
                            ("cdfpsi", uv, "psi_u"), ("cdfpsi", uv + ["-V"], "psi_v"),
                            ("cdfpsi", uv + ["-mean", "-mask", "-ref", "5", "4"], "psi_m"),
                            ("cdfpsi", uv + ["-open"], "psi_open"),               # stays on one device
                            ("cdfvT", ["-c", "SYN", "-l", "y0001"], "vt")):
        a, b = pair(tool, args, d, out, gmap)
        assert same_file(a, b), (tool, args)
    # cdfmhst on the vt file just written, with and without -Zdim, and from V,T,S (halo row between the blocks)
    for args, out in ((["-vt", "vt_1.nc", "-MST"], "mhst"), (["-vt", "vt_1.nc", "-MST", "-Zdim"], "mhst_z"),
                      (["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-MST"], "mhst_sep")):
        a, b = pair("cdfmhst", args, d, out, gmap)
        assert same_file(a, b), args
    a, b = pair("cdfheatc", ["-f", "SYN_y0001_gridT.nc"], d, "heatc", gmap)       # the reference's r4 sums, chained through the blocks
    assert same_file(a, b)
    os.environ["CDFT_HEATC_SUM"] = "r8"
    try:
        # r8 sums: block sums added in block order -- 1e-13, so the last printed digit may differ (it does at 1443 x 37 x 4)
        a, b = pair("cdfheatc", ["-f", "SYN_y0001_gridT.nc"], d, "heatc8", gmap, same_stdout=False)
    finally:
        del os.environ["CDFT_HEATC_SUM"]
    va, _, _ = read_nc(a)
    vb, _, _ = read_nc(b)
    for k in ("heatc2d", "heatc3d", "heatc3dpervol"):
        assert np.allclose(vb[k], va[k], rtol=1e-13, atol=0), k


@pytest.mark.gpu
def test_cdft_gpus_count_form_and_bad_values(case):
    d, g, rec = case
    rc, out, err = run("cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-o", "moc_g1.nc"], d, env={"CDFT_GPUS": "1"})
    assert rc == 0, out + err
    rc, out, err = run("cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-o", "moc_bad.nc"], d, env={"CDFT_GPUS": "9"})
    assert rc == 98 and "HIP device" in out                              # more GPUs than the box has: message + STOP 98
    rc, out, err = run("cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-o", "moc_bad.nc"], d, env={"CDFT_GPU_MAP": "zero"})
    assert rc == 98 and "CDFT_GPU_MAP" in out


@pytest.mark.gpu
def test_timing_totals_count_a_sharded_call_once(case):
    """CDFT_TIMING=1 with the plan on three row blocks: the blocks run the same entry points on their own threads, inside the
    caller's call -- the totals count the caller's calls, as many as on one device (they used to count the blocks' too)."""
    import re
    d, g, rec = case

    def calls(tool, args, env):
        rc, out, err = run(tool, args, d, env=dict(env, CDFT_TIMING="1"))
        assert rc == 0, out + err
        return {m.group(1): int(m.group(2)) for m in re.finditer(r"CDFT_TIMING library: (\S+)\s+(\d+) call", err)}
    for tool, args in (("cdfheatc", ["-f", "SYN_y0001_gridT.nc", "-o", "ht_timing.nc"]), ("cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-o", "moc_timing.nc"]),
                       ("cdfpsi", ["-u", "SYN_y0001_gridU.nc", "-v", "SYN_y0001_gridV.nc", "-o", "psi_timing.nc"]),
                       ("cdfmhst", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-o", "mhst_timing.nc"]),
                       ("cdfvT", ["-c", "SYN", "-l", "y0001", "-o", "vt_timing.nc"])):
        one, three = calls(tool, args, {}), calls(tool, args, {"CDFT_GPU_MAP": "0,0,0"})
        assert one, tool
        three = {k.replace("_sharded", ""): v for k, v in three.items()}        # (cdfvT's sharded entry has its own name)
        assert {k: three.get(k) for k in one} == one, (tool, one, three)
        if tool == "cdfheatc":
            assert one.get("cdft_heatc_compute") == 2               # two records
