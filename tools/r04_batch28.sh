#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_lift.py tests/test_transport.py tests/test_cli.py tests/test_cli_next.py tests/test_lift_next.py tests/test_fuzz_tools.py tests/test_sharded_plans.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python3 tools/bench_options.py --grid eorca025 2>/dev/null | grep -E "zdim" | cut -c1-140
python3 tools/bench_options.py --grid eorca12 --reps 5 2>/dev/null | grep -E "zdim" | cut -c1-140
python3 tools/mean_nb_vpt.py eorca025 2>/dev/null | grep '"vectors_per_lane": 4'
