#!/usr/bin/env python3
"""Print 'tool ms frac' from bench_tools JSON lines on stdin (helper for gpurun one-liners)."""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(d.get("tool"), d.get("ms"), d.get("frac_of_hbm_peak"))
