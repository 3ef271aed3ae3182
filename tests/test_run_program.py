"""tests/run_program.py itself: a run that does not end within its limit is described (every thread's state from /proc) and, with
CDFT_TEST_HANG=fail, fails the test with that description; a run that ends comes back like subprocess.run's."""
import os
import sys

import pytest

import run_program


def test_a_run_that_ends_comes_back(tmp_path):
    p = run_program.run_watched([sys.executable, "-c", "import sys; print('out'); print('err', file=sys.stderr); sys.exit(3)"],
                                cwd=str(tmp_path), env=dict(os.environ), timeout=60)
    assert (p.returncode, p.stdout.strip(), p.stderr.strip()) == (3, "out", "err")


def test_a_run_that_does_not_end_is_described_and_fails_on_request(tmp_path, monkeypatch):
    monkeypatch.setenv("CDFT_TEST_HANG", "fail")
    monkeypatch.setattr(run_program, "ROOT", str(tmp_path))               # (its report goes to <ROOT>/gpurun_out)
    with pytest.raises(AssertionError) as e:
        run_program.run_watched([sys.executable, "-c", "import time; print('begun', flush=True); time.sleep(60)"],
                                cwd=str(tmp_path), env=dict(os.environ), timeout=2)
    text = str(e.value)
    assert "HUNG after 2 s" in text and "thread " in text and "State:" in text and "begun" in text
    assert "HUNG after 2 s" in open(tmp_path / "gpurun_out" / "hung_programs.txt").read()


def test_a_run_that_does_not_end_is_an_xfail_by_default(tmp_path, monkeypatch):
    monkeypatch.delenv("CDFT_TEST_HANG", raising=False)
    monkeypatch.setattr(run_program, "ROOT", str(tmp_path))
    with pytest.raises(pytest.xfail.Exception):
        run_program.run_watched([sys.executable, "-c", "import time; time.sleep(60)"], cwd=str(tmp_path), env=dict(os.environ), timeout=1)
