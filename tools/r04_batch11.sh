#!/bin/bash
# heatc scan: plain (cached) loads for the chunks the walk reads again, against nontemporal for all (debug bit 8)
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_heatc.py tests/test_lift.py -m gpu -x -q 2>&1 | tail -1
FUZZ_SECONDS=60 python tools/fuzz_heatc_scan.py 99 2>&1 | tail -1
for g in eorca025 eorca12; do for d in 0 8 0 8; do
  ms=$(CDFT_HEATC_SCAN_DEBUG=$d python tools/bench_tools.py --grid $g --tools heatc --reps 40 2>/dev/null | grep "reference-order" | sed 's/.*"ms": \([0-9.]*\).*/\1/')
  echo "$g debug=$d ms $ms" | tee -a $O/heatc_keep.txt
done; done
