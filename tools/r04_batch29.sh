#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_moc_decomp.py tests/test_lift.py tests/test_cli.py tests/test_prog_diff.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
bash tools/kstat.sh _kernel decomp tools/decomp_kernels.py eorca025 2>&1 | grep -E "decomp " | cut -c1-150
