"""The long-running checkers of tools/ (random shapes and options against the oracle; device memory over 200 plan cycles
is tools/leak_check.py's business) each get a few seconds here so that they keep working: a fixed seed by default (a test
suite should not be a lottery), CDFT_FUZZ_SEED=<n> or =time for cases no earlier run has seen."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,seconds", [("fuzz_moc.py", 6), ("fuzz_tools.py", 10), ("fuzz_heatc_scan.py", 6)])
def test_fuzz_tool_runs_clean_for_a_few_seconds(gpu, tool, seconds):
    want = os.environ.get("CDFT_FUZZ_SEED", "4242")
    seed = (int(time.time()) & 0xffffff) if want == "time" else int(want)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, FUZZ_SECONDS=str(seconds)))
    assert p.returncode == 0, f"seed {seed}\n" + p.stdout[-3000:] + p.stderr[-3000:]
    assert f"seed {seed}" in p.stdout and " cases" in p.stdout, p.stdout[-500:]
