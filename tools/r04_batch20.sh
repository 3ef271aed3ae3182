#!/bin/bash
mkdir -p gpurun_out/r04
for v in 4 8 6; do
  CDFT_VINT_VPT=$v bash tools/kstat.sh vint_kernel psi025_vpt$v tools/bench_tools.py --grid eorca025 --tools psi --reps 20
done 2>&1 | tee gpurun_out/r04/vint_vpt8_eorca025.txt
