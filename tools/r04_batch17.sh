#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_heatc.py tests/test_lift.py tests/test_cli.py tests/test_sharded_plans.py -m gpu -x -q 2>&1 | tail -3
(bash tools/ktrace_seq.sh heatc12 14 tools/bench_tools.py --grid eorca12 --tools heatc --reps 10; bash tools/ktrace_seq.sh heatc025 14 tools/bench_tools.py --grid eorca025 --tools heatc --reps 10) > gpurun_out/r04/kseq_heatc.txt 2>&1; cat gpurun_out/r04/kseq_heatc.txt
python3 tools/bench_tools.py --grid eorca025 --tools heatc,mean --reps 20 > gpurun_out/r04/heatc_mean_025.jsonl 2>&1
python3 tools/bench_tools.py --grid eorca12 --tools heatc,mean --reps 20 > gpurun_out/r04/heatc_mean_12.jsonl 2>&1
python3 - <<'PY'
import json
for f in ("gpurun_out/r04/heatc_mean_025.jsonl","gpurun_out/r04/heatc_mean_12.jsonl"):
    for l in open(f):
        try: d=json.loads(l)
        except Exception: print(l.strip()[:200]); continue
        print(d["grid"], d["tool"], d["ms"], d["frac_of_hbm_peak"], d.get("kernels_only_ms"), d.get("kernels_only_frac"))
PY
