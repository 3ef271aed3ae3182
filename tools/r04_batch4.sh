#!/bin/bash
# round-4 GPU batch 4: lease g, the whole GPU suite, per-tool lines + rocprof stats at both grids, heatc ticket sizes at eORCA025, cdfcurl PMC
O=gpurun_out/r04; mkdir -p $O
bash tools/lease_probe.sh g > /dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/lease_g/bench.json")); r=d["roofline"]
print("lease g avg %.3f min %.3f max %.3f frac %.4f dense %.3f inproc %s" % (r["avg_launch_ms"], r["min_launch_ms"], r["max_launch_ms"], r["frac"], r["dense_arrays"]["avg_launch_ms"], d["inprocess"].get("roofline",{}).get("avg_launch_ms", d["inprocess"].get("error"))))
PY
python -m pytest tests -m gpu -q -x > $O/pytest_all_g.txt 2>&1; tail -3 $O/pytest_all_g.txt
for c in 64 128 256; do echo "== eorca025 SCAN_CHUNKS=$c"; CDFT_HEATC_SCAN_CHUNKS=$c python tools/bench_tools.py --grid eorca025 --tools heatc --reps 40 2>/dev/null | grep "reference-order" | sed 's/.*"ms": \([0-9.]*\).*/   ms \1/'; done
for c in 64 128 256; do echo "== eorca12 SCAN_CHUNKS=$c"; CDFT_HEATC_SCAN_CHUNKS=$c python tools/bench_tools.py --grid eorca12 --tools heatc --reps 20 2>/dev/null | grep "reference-order" | sed 's/.*"ms": \([0-9.]*\).*/   ms \1/'; done
bash tools/profile_tools.sh eorca12 > /dev/null 2>&1
bash tools/profile_tools.sh eorca025 > /dev/null 2>&1
grep -h '"tool"' gpurun_out/tools_eorca12/lines.jsonl | sed 's/{"tool": "\([^"]*\)".*"ms": \([0-9.]*\).*"frac_of_hbm_peak": \([0-9.]*\).*/\1 | \2 ms | \3/' | cut -c1-150
echo ---- eorca025
grep -h '"tool"' gpurun_out/tools_eorca025/lines.jsonl | sed 's/{"tool": "\([^"]*\)".*"ms": \([0-9.]*\).*"frac_of_hbm_peak": \([0-9.]*\).*/\1 | \2 ms | \3/' | cut -c1-150
bash tools/pmc_curl.sh eorca12 r04 > $O/pmc_curl_eorca12.txt 2>&1; tail -4 $O/pmc_curl_eorca12.txt
