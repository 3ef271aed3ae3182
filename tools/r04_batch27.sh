#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_stencil.py tests/test_cli.py tests/test_fuzz_tools.py tests/test_prog_diff.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python3 tools/bench_options.py --grid eorca025 2>/dev/null | grep curl | cut -c1-140
python3 tools/bench_options.py --grid eorca12 --reps 5 2>/dev/null | grep curl | cut -c1-140
