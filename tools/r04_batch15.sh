#!/bin/bash
mkdir -p gpurun_out/r04
for d in 0 1 2 3; do
  CDFT_VINT_DEBUG=$d bash tools/kstat.sh vint_kernel,psi_scan,mhst_zonal dbg$d tools/bench_tools.py --grid eorca025 --tools mhst,psi --reps 20
done 2>&1 | tee gpurun_out/r04/vint_debug_eorca025.txt
