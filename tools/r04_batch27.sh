#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_transport.py tests/test_cli.py tests/test_fuzz_tools.py tests/test_sharded_plans.py -m gpu -x -q -k "psi" 2>&1 | grep -E "passed|failed|rror" | tail -3
python3 tools/bench_options.py --grid eorca025 2>/dev/null | grep -E "cdfpsi" | cut -c1-150
bash tools/ktrace_seq.sh psiv 8 tools/bench_options.py --grid eorca025 2>&1 | grep -E "psi|vint" | tail -8
