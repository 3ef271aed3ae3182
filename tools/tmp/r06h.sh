mkdir -p gpurun_out/r06h; cd $GRAFT_REPO_ROOT
t0=$(date +%s); n=0
while [ $(( $(date +%s) - t0 )) -lt 1100 ]; do
  n=$((n+1))
  timeout 1200 python -m pytest tests/test_cli.py tests/test_cli_next.py tests/test_cli_feed.py tests/test_cli_cdf5.py tests/test_cli_rapid.py tests/test_prog_diff.py tests/test_moc_pool_feed.py tests/test_cli_stack.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r06h/pass_$n.log 2>&1
  rc=$?
  echo "pass $n rc $rc $(grep -E 'passed|failed' gpurun_out/r06h/pass_$n.log | tail -1)" >> gpurun_out/r06h/summary.txt
  if [ $rc -ne 0 ]; then break; fi
done
cat gpurun_out/r06h/summary.txt
STRESS_SECONDS=420 python tools/stress_feed.py > gpurun_out/r06h/stress_feed.txt 2>&1; tail -1 gpurun_out/r06h/stress_feed.txt
STRESS_SECONDS=240 STRESS_FEED=0 python tools/stress_feed.py > gpurun_out/r06h/stress_feed0.txt 2>&1; tail -1 gpurun_out/r06h/stress_feed0.txt
ls gpurun_out/hung_programs.txt 2>/dev/null && cp gpurun_out/hung_programs.txt gpurun_out/r06h/
