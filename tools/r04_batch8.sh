#!/bin/bash
# cdfmean with the next level in flight: bits and time; heatc skeleton experiments
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_lift_next.py tests/test_fuzz_tools.py tests/test_cli_next.py tests/test_nonfinite.py -m gpu -x -q -k "mean or sum or nonfinite or fuzz" 2>&1 | tail -2
for g in eorca025 eorca12; do python tools/bench_tools.py --grid $g --tools mean --reps 30 2>/dev/null | sed 's/{"tool": "\([^"]*\)".*"ms": \([0-9.]*\).*"frac_of_hbm_peak": \([0-9.]*\).*kernels_only_ms": \([0-9.]*\).*kernels_only_frac": \([0-9.]*\).*/\1 | call \2 ms \3 | kernels \4 ms \5/' | cut -c1-160; done
for e in "" "CDFT_HEATC_SCAN_LANES=1" "CDFT_HEATC_SCAN_NOALIGN=1" "CDFT_HEATC_SCAN_CHUNKS=128" "CDFT_XCD_BANDS=0"; do
  ms=$(env CDFT_HEATC_SCAN_DEBUG=7 $e python tools/bench_tools.py --grid eorca12 --tools heatc --reps 20 2>/dev/null | grep "reference-order" | sed 's/.*"ms": \([0-9.]*\).*/\1/')
  echo "eorca12 skeleton (debug=7) [$e] ms $ms" | tee -a $O/heatc_skeleton.txt
done
