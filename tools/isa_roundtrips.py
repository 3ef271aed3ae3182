#!/usr/bin/env python3
"""For every kernel of cdftools_amd/csrc/*.hip: how many times the innermost loops wait for ALL outstanding loads
(s_waitcnt vmcnt(0)) between two global loads of one iteration -- each is a full memory round trip inside the iteration
(what a run-time branch around a load does: mean_level_kernel's mask, found in round 4).
    python3 tools/isa_roundtrips.py [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_ops


def main():
    files = sys.argv[1:] or sorted(os.path.basename(f) for f in glob.glob(os.path.join(isa_ops.CSRC, "*.hip")))
    for src in files:
        asm = isa_ops.device_asm(src)
        fns = isa_ops.functions(asm)
        names = list(fns)
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        for name, d in zip(names, dem):
            lines = fns[name]
            # the blocks of every loop, by the compiler's own annotation of the block labels
            loops, cur = {}, None
            for ln in lines:
                m = re.match(r"^\.LBB(\d+_\d+):(.*)", ln)
                if m:
                    h = re.search(r"in Loop: Header=BB(\d+_\d+)", m.group(2))
                    cur = m.group(1) if "Loop Header" in m.group(2) else (h.group(1) if h else None)
                    continue
                if cur is not None:
                    loops.setdefault(cur, []).append(ln)
            worst = None
            for body in loops.values():
                trips, state, loads = 0, 0, 0          # state 1: a load is out; 2: and all loads have been waited for since
                for b in body:
                    if re.match(r"^\s+(global|buffer|flat)_load", b):
                        if state == 2:
                            trips += 1
                        state = 1
                        loads += 1
                    elif re.match(r"^\s+s_waitcnt\s+vmcnt\(0\)", b) and state == 1:
                        state = 2
                if loads >= 2 and (worst is None or trips > worst[0]):
                    worst = (trips, loads, len(body))
            if worst and worst[0] > 0:
                k = re.search(r"\w+_kernel(<[^>]*>)?", d)
                print("%-16s %-70s round trips inside one iteration: %d (+1)   loads %d, body %d lines" % (
                    src, (k.group(0) if k else d)[:70], worst[0], worst[1], worst[2]))


if __name__ == "__main__":
    main()
