"""bench.py's two CPU legs without a GPU: the oracle port with its thread-count pick, and the reference's own statements
(oracle/_ref/liblift_ref.so, built where /root/reference exists) run in a child process whose stdout is closed -- on the
bench's own synthetic inputs at a small size, both against each other."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench            # noqa: E402
import oracle_lib       # noqa: E402
import ref_lift_lib     # noqa: E402


@pytest.fixture(scope="module")
def inputs():
    npi, npj, npk, npt, nb = 182, 40, 31, 3, 5
    return (npi, npj, npk, npt, nb) + tuple(bench.build_inputs(npi, npj, npk, npt, nb, 0, npj, torch.device("cpu")))


def test_inputs_do_not_depend_on_the_row_block(inputs):
    npi, npj, npk, npt, nb, e1v, e3v, ib, zv = inputs
    a = bench.build_inputs(npi, npj, npk, npt, nb, 11, 29, torch.device("cpu"))
    assert torch.equal(a[0], e1v[11:29]) and torch.equal(a[1], e3v[:, 11:29]) and torch.equal(a[3], zv[:, :, 11:29])
    assert torch.equal(a[2][:, 1:-1], ib[11:29][:, 1:-1])            # (basin 0 loses the first and last column of every block alike)


def test_port_leg_picks_a_thread_count_and_returns_the_oracle_result(inputs):
    npi, npj, npk, npt, nb, e1v, e3v, ib, zv = inputs
    cpu, rows, out = bench.cpu_baseline(e1v, e3v, ib, zv, npi, npk, npt, target_s=0.2)
    assert cpu["kind"] == "port" and cpu["value"] > 0 and 1 <= cpu["cores"] <= (os.cpu_count() or 1) and rows == npj
    assert np.array_equal(out, oracle_lib.moc(e1v.numpy(), e3v.numpy(), ib.numpy(), zv.numpy()))


@pytest.mark.skipif(not ref_lift_lib.available(), reason="oracle/_ref/liblift_ref.so is only built where /root/reference exists")
def test_reference_leg_runs_in_a_child_and_equals_the_oracle(inputs, capfd):
    npi, npj, npk, npt, nb, e1v, e3v, ib, zv = inputs
    ref = oracle_lib.moc(e1v.numpy(), e3v.numpy(), ib.numpy(), zv.numpy())
    r = bench.reference_baseline(e1v, e3v, ib, zv, npi, npk, npt, ref, rows=16, target_s=0.2)
    assert r["kind"] == "reference" and r["cores"] == 1 and r["value"] > 0
    assert r["hip_vs_reference_rel_err"] == 0.0                        # (here "hip" is the oracle: the two agree bit for bit)
    assert "Equation of state" not in capfd.readouterr().out          # the reference's eos_init prints stay in the child
