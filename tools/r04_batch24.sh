#!/bin/bash
for g in eorca12 eorca025; do echo "$g"; python3 tools/bench_tools.py --grid $g --tools vt --reps 5 2>/dev/null | cut -c1-160; done
