#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_transport.py -m gpu -x -q --durations=5 > gpurun_out/r04/pytest_part.txt 2>&1; grep -E "passed|failed|error|s call|s setup" gpurun_out/r04/pytest_part.txt | tail -8
(bash tools/ktrace_seq.sh psi025 6 tools/bench_tools.py --grid eorca025 --tools psi --reps 10; bash tools/ktrace_seq.sh psi12 6 tools/bench_tools.py --grid eorca12 --tools psi --reps 10) 2>&1 | tee gpurun_out/r04/kseq_psi.txt
python3 tools/bench_tools.py --grid eorca025 --tools psi --reps 20 2>/dev/null | cut -c1-200
python3 tools/bench_tools.py --grid eorca12 --tools psi --reps 20 2>/dev/null | cut -c1-200
