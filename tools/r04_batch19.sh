#!/bin/bash
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_transport.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
for d in 0 3 27; do
  CDFT_PSI_SCAN_DEBUG=$d bash tools/kstat.sh psi_scan dbg$d tools/bench_tools.py --grid eorca025 --tools psi --reps 20
done 2>&1 | tee gpurun_out/r04/psi_scan_debug.txt
for d in 0 3; do
  CDFT_PSI_SCAN_DEBUG=$d bash tools/kstat.sh psi_scan dbg12_$d tools/bench_tools.py --grid eorca12 --tools psi --reps 10
done 2>&1 | tee -a gpurun_out/r04/psi_scan_debug.txt
