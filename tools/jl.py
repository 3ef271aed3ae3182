#!/usr/bin/env python3
"""Print 'tool ms frac' from bench_tools JSON lines in the files named, or on stdin when none is (helper for gpurun
one-liners)."""
import json
import sys

import fileinput

for line in fileinput.input():
    if line.startswith("{"):
        d = json.loads(line)
        print(d.get("tool"), d.get("ms"), d.get("frac_of_hbm_peak"))
