#!/bin/bash
# vint: vectors per lane against the number of rounds (tail of the last round); fuzz_moc with the staged path
O=gpurun_out/r04; mkdir -p $O
for v in 0 4 6 8; do for t in psi mhst; do ms=$(CDFT_VINT_VPT=$v python tools/bench_tools.py --grid eorca12 --tools $t --reps 30 2>/dev/null | grep '"tool"' | sed 's/.*"ms": \([0-9.]*\).*/\1/'); echo "eorca12 $t VPT=$v ms $ms" | tee -a $O/vint_vpt_eorca12.txt; done; done
FUZZ_SECONDS=150 python tools/fuzz_moc.py 4711 2>&1 | tail -1
