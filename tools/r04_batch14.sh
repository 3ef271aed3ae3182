#!/bin/bash
# counters of the level-walking kernels at eORCA025 (verdict item 6): what do they wait for
mkdir -p gpurun_out/r04
R=$(pwd)
for t in "mean_level_kernel mean025 mean" "vint_kernel mhst025 mhst" "vint_kernel psi025 psi" "heatc_level heatc025 heatc" "vt_kernel vt025 vt"; do
  set -- $t
  bash tools/pmc_stalls.sh $1 $2 tools/bench_tools.py --grid eorca025 --tools $3 --reps 3 > gpurun_out/r04/stalls_$2.txt 2>&1
done
cd $R; tail -n 25 gpurun_out/r04/stalls_*025.txt
