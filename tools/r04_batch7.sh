#!/bin/bash
# heatc with DPP scans: bits, fuzz, time; cdfmoc bench at eORCA025 (pool against dense)
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_heatc.py tests/test_lift.py tests/test_lift_vvl.py -m gpu -x -q 2>&1 | tail -2
FUZZ_SECONDS=120 python tools/fuzz_heatc_scan.py 777 2>&1 | tail -1
for g in eorca12 eorca025; do for d in 0 4 5; do
  ms=$(CDFT_HEATC_SCAN_DEBUG=$d python tools/bench_tools.py --grid $g --tools heatc --reps 20 2>/dev/null | grep "reference-order" | sed 's/.*"ms": \([0-9.]*\).*/\1/')
  echo "$g debug=$d ms $ms" | tee -a $O/heatc_debug_dpp.txt
done; done
python3 bench.py --workload eorca025 --steps 200 --warmup 10 --no-inprocess > $O/bench_eorca025.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/bench_eorca025.json")); r=d["roofline"]
print("eorca025 cdfmoc: pool avg %.4f ms frac %.3f kernel %s | dense %.4f ms frac %.3f kernel %s | value %.3e" % (r["avg_launch_ms"], r["frac"], r["kernel"], r["dense_arrays"]["avg_launch_ms"], r["dense_arrays"]["frac"], r["dense_arrays"]["kernel"], d["value"]))
PY
python tools/leak_check.py 2>&1 | tail -3
