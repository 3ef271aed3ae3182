"""bench.py end to end on the GPU box: the N = 1 line, and the N = 2 code path (self-launch, row blocks, step pipeline,
gather of every step, oracle check on rank 0) on ONE GPU through the test hooks -- both ranks on cuda:0, transfers by
gloo, because RCCL does not accept two ranks on one device.  The RCCL transport itself only runs on a multi-GPU node
(the driver's SCALE runs)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_n1_line_small_workload(gpu):
    r = run_bench(["--workload", "eorca025", "--steps", "20", "--warmup", "2"])
    assert r["n_gpus"] == 1 and r["steps"] == 20 and r["oracle_check_rel_err"] < 1e-11
    rf = r["roofline"]
    assert rf["launches"] == 20 and rf["min_launch_ms"] <= rf["median_launch_ms"] <= rf["max_launch_ms"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and r["cpu_baseline"]["cores"] >= 1
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["value"] > 0
    ref = r["cpu_baseline_reference"]                # the reference's own statements, when oracle/_ref travelled with the snapshot
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liblift_ref.so")):
        assert ref["kind"] == "reference" and ref["cores"] == 1 and ref["value"] > 0 and ref["hip_vs_reference_rel_err"] < 1e-11, ref
    else:
        assert ref is None


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3])
def test_bench_n_gt_1_path_on_one_gpu_with_test_hooks(gpu, n):
    r = run_bench(["--gpus", str(n), "--workload", "eorca025", "--steps", "6", "--warmup", "2"],
                  env={"CDFT_BENCH_ONE_GPU": "1", "CDFT_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == n and r["config"]["result_gathers"] == 6 + 2
    assert r["oracle_check_rel_err"] < 1e-11 and r["test_hooks"] == {"one_gpu": True, "backend": "gloo"}
    assert r["roofline"]["launches"] == 6 and r["cpu_baseline"] is None
