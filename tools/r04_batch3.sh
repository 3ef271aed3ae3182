#!/bin/bash
# round-4 GPU batch 3: lease f (+ read ceiling, reserve A/B), heatc occupancy 4, curl bands at both grids, vint VPT at eORCA025, row-block workloads
O=gpurun_out/r04; mkdir -p $O
bash tools/lease_probe.sh f > /dev/null
CDFT_BENCH_RESERVE=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-inprocess > $O/lease_f/bench_noreserve.json 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-inprocess > $O/lease_f/bench_reserve_again.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench","bench_noreserve","bench_reserve_again"):
    d=json.load(open(f"gpurun_out/r04/lease_f/{f}.json")); r=d["roofline"]
    print(f, "avg %.3f min %.3f max %.3f frac %.4f dense %.3f stage_s %.3f" % (r["avg_launch_ms"], r["min_launch_ms"], r["max_launch_ms"], r["frac"], r["dense_arrays"]["avg_launch_ms"], d["config"]["stage_s"]))
best=0
for l in open("gpurun_out/r04/lease_f/hbm_probe.jsonl"):
    try: j=json.loads(l)
    except Exception: continue
    best=max(best, j.get("read_GBps",0))
print("read ceiling GB/s", best)
PY
python -m pytest tests/test_heatc.py -m gpu -x -q 2>&1 | tail -1
for g in eorca12 eorca025; do python tools/bench_tools.py --grid $g --tools heatc --reps 20 2>/dev/null | grep "reference-order" | cut -c1-200; done
for w in eorca12_half eorca12_quarter eorca12_eighth; do python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-inprocess --no-dense 2>/dev/null > $O/bench_$w.json; python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
for g in eorca12 eorca025; do for b in 0 16 32 64 128; do for lc in 0 1 3 5 8 15 25; do
  e="CDFT_CURL_BAND=$b"; [ $lc -gt 0 ] && e="$e CDFT_CURL_CHUNKS=$lc"
  ms=$(env $e python tools/bench_tools.py --grid $g --tools curl --reps 10 2>/dev/null | grep '"tool": "cdfcurl' | head -1 | sed 's/.*"ms": \([0-9.]*\).*/\1/')
  echo "$g band $b chunks $lc ms $ms" >> $O/curl_band_f.txt
done; done; done
sort -k7 -n $O/curl_band_f.txt | awk '{print}' | head -40
for v in 0 1 2 4; do echo "== eorca025 VINT_VPT=$v"; CDFT_VINT_VPT=$v python tools/bench_tools.py --grid eorca025 --tools mhst,psi --reps 20 2>/dev/null | cut -c1-150; done > $O/vint_vpt_f.txt
grep -E "==|\"ms\"" $O/vint_vpt_f.txt | sed 's/{"tool": "\([^"]*\)".*"ms": \([0-9.]*\).*/   \1 \2/'
