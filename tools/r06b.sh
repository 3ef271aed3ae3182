mkdir -p gpurun_out/r06b; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_heatc.py -m gpu -x -q > gpurun_out/r06b/pytest_heatc.log 2>&1; echo "rc $?" >> gpurun_out/r06b/pytest_heatc.log; tail -5 gpurun_out/r06b/pytest_heatc.log
for g in eorca025 eorca12; do
  for m in 0 1; do
    echo "== $g walkers=$m" >> gpurun_out/r06b/ab.txt
    CDFT_HEATC_SCAN_WALKERS=$m timeout 300 python tools/bench_tools.py --grid $g --tools heatc 2>&1 | grep "reference-order" | cut -c1-260 >> gpurun_out/r06b/ab.txt
  done
done
cat gpurun_out/r06b/ab.txt
FUZZ_SECONDS=40 timeout 400 python tools/fuzz_heatc_scan.py 606 > gpurun_out/r06b/fuzz.txt 2>&1; tail -2 gpurun_out/r06b/fuzz.txt
