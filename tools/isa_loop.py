#!/usr/bin/env python3
"""Mnemonic histogram of the main loop (the backward branch with the longest body) of one kernel, from the gfx950 assembly
the compiler emits with the flags of cdftools_amd/csrc/Makefile.
    python3 tools/isa_loop.py mean.hip 'mean_level_kernel<4, false, 1, true>' [cells per iteration]"""
import collections
import os
import re
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_ops


def main():
    src, pat = sys.argv[1], sys.argv[2]
    cells = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    asm = isa_ops.device_asm(src)
    for name, lines in isa_ops.functions(asm).items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pat not in dem:
            continue
        label_at = {}
        for i, ln in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                label_at[m.group(1)] = i
        best = None
        for i, ln in enumerate(lines):
            m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in label_at and label_at[m.group(1)] < i:
                body = lines[label_at[m.group(1)]:i]
                n = sum(1 for b in body if re.match(r"^\s+[vs]_|^\s+(global|ds|buffer|flat|scratch)_", b))
                if best is None or n > best[0]:
                    best = (n, body)
        h = collections.Counter()
        for b in best[1]:
            m = re.match(r"^\s+([a-z][a-z0-9_]+)", b)
            if m and not m.group(1).startswith("s_nop"):
                h[m.group(1)] += 1
        tot_v = sum(v for k, v in h.items() if k.startswith("v_"))
        print(dem[:120])
        print("  loop body: %d instructions, %d vector (%.2f per cell), %d scalar, %d memory" % (
            sum(h.values()), tot_v, tot_v / cells, sum(v for k, v in h.items() if k.startswith("s_")),
            sum(v for k, v in h.items() if re.match(r"global|ds|buffer|flat|scratch", k))))
        for k, v in h.most_common():
            print("    %-28s %4d  %.2f" % (k, v, v / cells))
        d = re.search(re.escape(name) + r"\n.*?\.end_amdhsa_kernel", asm, re.S)
        d = re.search(r"\.amdhsa_kernel " + re.escape(name) + r".*?\.end_amdhsa_kernel", asm, re.S)
        if d:
            for key in ("next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size", "accum_offset"):
                mm = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", d.group(0))
                if mm:
                    print("  %s = %s" % (key, mm.group(1)))


if __name__ == "__main__":
    main()
