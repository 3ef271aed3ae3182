#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_heatc.py -m gpu -x -q 2>&1 | tail -1
for g in eorca025 eorca12; do for d in 0 16 7 23; do
  ms=$(CDFT_HEATC_SCAN_DEBUG=$d python tools/bench_tools.py --grid $g --tools heatc --reps 40 2>/dev/null | grep "reference-order" | sed 's/.*"ms": \([0-9.]*\).*/\1/')
  echo "$g debug=$d ms $ms" | tee -a $O/heatc_contig.txt
done; done
