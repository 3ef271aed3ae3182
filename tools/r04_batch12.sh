#!/bin/bash
# heatc scan with three sub-blocks of loads ahead: bits, fuzz, time (and the skeleton alone)
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_heatc.py tests/test_lift.py tests/test_lift_vvl.py -m gpu -x -q 2>&1 | tail -1
FUZZ_SECONDS=90 python tools/fuzz_heatc_scan.py 31337 2>&1 | tail -1
for g in eorca025 eorca12; do for d in 0 7 0; do
  ms=$(CDFT_HEATC_SCAN_DEBUG=$d python tools/bench_tools.py --grid $g --tools heatc --reps 40 2>/dev/null | grep "reference-order" | sed 's/.*"ms": \([0-9.]*\).*/\1/')
  echo "$g ahead=3 debug=$d ms $ms" | tee -a $O/heatc_ahead3.txt
done; done
