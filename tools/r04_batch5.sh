#!/bin/bash
# round-4 GPU batch 5: lease h, the whole GPU suite, cdfmocsig and cdfcurl after their round-4 changes, PMC bytes of cdfcurl
O=gpurun_out/r04; mkdir -p $O
bash tools/lease_probe.sh h > /dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/lease_h/bench.json")); r=d["roofline"]
print("lease h avg %.3f min %.3f max %.3f frac %.4f dense %.3f inproc %s" % (r["avg_launch_ms"], r["min_launch_ms"], r["max_launch_ms"], r["frac"], r["dense_arrays"]["avg_launch_ms"], d["inprocess"].get("roofline",{}).get("avg_launch_ms", d["inprocess"].get("error"))))
best=0
for l in open("gpurun_out/r04/lease_h/hbm_probe.jsonl"):
    try: j=json.loads(l)
    except Exception: continue
    best=max(best, j.get("read_GBps",0))
print("read ceiling GB/s", best)
PY
python -m pytest tests -m gpu -q > $O/pytest_all_h.txt 2>&1; tail -4 $O/pytest_all_h.txt
for g in eorca12 eorca025; do python tools/bench_tools.py --grid $g --tools mocsig,curl --reps 10 --land noise 2>/dev/null > $O/mocsig_curl_${g}_h.jsonl; grep -h '"tool"' $O/mocsig_curl_${g}_h.jsonl | sed 's/{"tool": "\([^"]*\)".*"ms": \([0-9.]*\).*"frac_of_hbm_peak": \([0-9.]*\).*/\1 | \2 ms | \3/' | cut -c1-150; done
CDFT_MOCSIG_SPVAL_GENERAL=1 python tools/bench_tools.py --grid eorca12 --tools mocsig --reps 10 --land noise 2>/dev/null | grep -h '"tool"' | head -1 | sed 's/{"tool": "\([^"]*\)".*"ms": \([0-9.]*\).*/general form: \1 | \2 ms/' | cut -c1-150
bash tools/pmc_kernels.sh eorca12 curl curl_r04 2>&1 | tail -2
CDFT_CURL_BAND=0 bash tools/pmc_kernels.sh eorca12 curl curl_r04_nobands 2>&1 | tail -2
