#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_heatc.py tests/test_lift.py tests/test_sharded_plans.py tests/test_nonfinite.py -m gpu -x -q > gpurun_out/r04/pytest_part.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r04/pytest_part.txt | tail -5
(bash tools/ktrace_seq.sh heatc12 10 tools/bench_tools.py --grid eorca12 --tools heatc --reps 10; bash tools/ktrace_seq.sh heatc025 10 tools/bench_tools.py --grid eorca025 --tools heatc --reps 10) > gpurun_out/r04/kseq_heatc.txt 2>&1; cat gpurun_out/r04/kseq_heatc.txt
for g in eorca025 eorca12; do CDFT_HEATC_SCAN_STATS=1 python3 tools/bench_tools.py --grid $g --tools heatc --reps 20 2>gpurun_out/r04/heatc_stats_$g.err | cut -c1-180; tail -2 gpurun_out/r04/heatc_stats_$g.err; done
