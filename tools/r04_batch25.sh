#!/bin/bash
mkdir -p gpurun_out/r04
run() { for g in eorca025 eorca12; do for d in 0 0; do echo "$1 $g"; python3 tools/bench_tools.py --grid $g --tools heatc --reps 20 2>/dev/null | grep "r4 sums" | cut -c100-170; done; done; }
run prefetch
cp cdftools_amd/csrc/heatc.hip /tmp/heatc_new.hip
cp gpurun_in_heatc_baseline.txt cdftools_amd/csrc/heatc.hip
make -C cdftools_amd/csrc -j8 2>&1 | grep -E "error" | head -3
run baseline
cp /tmp/heatc_new.hip cdftools_amd/csrc/heatc.hip
make -C cdftools_amd/csrc -j8 2>&1 | grep -E "error" | head -3
run prefetch
