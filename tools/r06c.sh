mkdir -p gpurun_out/r06c; cd $GRAFT_REPO_ROOT
out=gpurun_out/r06c/ab2.txt; : > $out
run() { echo "== $*" >> $out; env "$@" timeout 300 python tools/bench_tools.py --grid $G --tools heatc 2>gpurun_out/r06c/err.txt | grep "reference-order" | python -c "import sys,json; [print(json.loads(l)['ms']) for l in sys.stdin]" >> $out; grep "timing" gpurun_out/r06c/err.txt | tail -2 >> $out; }
for G in eorca12 eorca025; do
echo "#### $G" >> $out
run CDFT_HEATC_SCAN_WALKERS=0
run CDFT_HEATC_SCAN_WALKERS=1
run CDFT_HEATC_SCAN_WALKERS=1 CDFT_HEATC_SCAN_DEBUG=8 CDFT_HEATC_SCAN_STATS=1
run CDFT_HEATC_SCAN_WALKERS=1 CDFT_HEATC_SCAN_DEBUG=12 CDFT_HEATC_SCAN_STATS=1
done
cat $out
timeout 600 python -m pytest tests/test_heatc.py -m gpu -x -q 2>&1 | tail -2
