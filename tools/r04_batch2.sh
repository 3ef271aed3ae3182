#!/bin/bash
# round-4 GPU batch 2: lease e, program differential, heatc scan (bits, time, counters), curl row bands
O=gpurun_out/r04; mkdir -p $O
bash tools/lease_probe.sh e > /dev/null
python -m pytest tests/test_prog_diff.py -m gpu -x -q > $O/pytest_prog_e.txt 2>&1; tail -3 $O/pytest_prog_e.txt
python -m pytest tests/test_heatc.py tests/test_lift.py -m gpu -x -q > $O/pytest_heatc_e.txt 2>&1; tail -3 $O/pytest_heatc_e.txt
FUZZ_SECONDS=90 python tools/fuzz_heatc_scan.py 4242 > $O/fuzz_heatc_e.txt 2>&1; tail -2 $O/fuzz_heatc_e.txt
for g in eorca12 eorca025; do python tools/bench_tools.py --grid $g --tools heatc --reps 20 > $O/heatc_${g}_e.jsonl 2>&1; done
grep -h "reference-order" $O/heatc_eorca12_e.jsonl $O/heatc_eorca025_e.jsonl | cut -c1-260
bash tools/pmc_stalls.sh heatc_scan_kernel heatc_r04 tools/bench_tools.py --grid eorca12 --tools heatc --reps 3 > $O/heatc_stalls_e.txt 2>&1
grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|_valu_busy|_waves" $O/heatc_stalls_e.txt
for b in 0 32 64 128 256; do for lc in 0 3 5 15; do
  e="CDFT_CURL_BAND=$b"; [ $lc -gt 0 ] && e="$e CDFT_CURL_CHUNKS=$lc"
  echo "== band $b chunks $lc" >> $O/curl_band_e.txt
  env $e python tools/bench_tools.py --grid eorca12 --tools curl --reps 10 2>&1 | grep '"tool"' | cut -c1-230 >> $O/curl_band_e.txt
done; done
cat $O/curl_band_e.txt | grep -E "==|\"ms\"" | sed 's/.*"ms": \([0-9.]*\).*/   ms \1/' | paste - - | head -30
