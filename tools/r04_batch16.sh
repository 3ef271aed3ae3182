#!/bin/bash
mkdir -p gpurun_out/r04
for v in 2 4; do
  CDFT_MEAN_VPT=$v timeout 900 python -m pytest tests/test_cli_next.py tests/test_lift_next.py tests/test_nonfinite.py tests/test_cli_vvl.py tests/test_xcd_bands.py tests/test_fuzz_tools.py -m gpu -x -q -k "mean or sum" 2>&1 | tail -3
done
for g in eorca025 eorca12; do for v in 2 4; do
  CDFT_MEAN_VPT=$v bash tools/kstat.sh mean_level,mean_final ${g}_vpt$v tools/bench_tools.py --grid $g --tools mean --reps 20
done; done 2>&1 | tee gpurun_out/r04/mean_vpt.txt
