#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_stencil.py tests/test_cli.py tests/test_lift.py tests/test_sharded_plans.py tests/test_xcd_bands.py tests/test_nonfinite.py -m gpu -x -q -k "vt or vT or VT" > gpurun_out/r04/pytest_part.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r04/pytest_part.txt | tail -3
python3 tools/bench_tools.py --grid eorca025 --tools vt --reps 20 2>/dev/null | cut -c1-220
python3 tools/bench_tools.py --grid eorca12 --tools vt --reps 10 2>/dev/null | cut -c1-220
