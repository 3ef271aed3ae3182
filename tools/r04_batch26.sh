#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_cli_next.py tests/test_lift_next.py tests/test_nonfinite.py tests/test_cli_vvl.py tests/test_fuzz_tools.py tests/test_xcd_bands.py -m gpu -x -q -k "zonal" 2>&1 | grep -E "passed|failed|rror" | tail -3
python3 tools/zonal_nb.py eorca025 2>/dev/null | grep '"basins": 1' | tee gpurun_out/r04/zonal_nb_eorca025.jsonl
python3 tools/zonal_nb.py eorca12 2>/dev/null | grep '"basins": 1' | tee gpurun_out/r04/zonal_nb_eorca12.jsonl
