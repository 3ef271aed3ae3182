#!/bin/bash
# heatc scan timing experiments (results invalid with DEBUG != 0): where do the 2.2 ms go?
O=gpurun_out/r04; mkdir -p $O
for g in eorca12 eorca025; do for d in 0 4 5 7; do
  ms=$(CDFT_HEATC_SCAN_DEBUG=$d python tools/bench_tools.py --grid $g --tools heatc --reps 20 2>/dev/null | grep "reference-order" | sed 's/.*"ms": \([0-9.]*\).*/\1/')
  echo "$g debug=$d ms $ms" | tee -a $O/heatc_debug.txt
done; done
CDFT_HEATC_SCAN_STATS=1 python tools/bench_tools.py --grid eorca025 --tools heatc --reps 3 2>&1 | grep -i "scan\|fallback\|resolved" | head -5
python tools/bench_tools.py --grid eorca025 --tools mean,mhst --reps 30 2>/dev/null | cut -c1-420
python tools/bench_tools.py --grid eorca12 --tools mean,mhst --reps 20 2>/dev/null | cut -c1-420
