#!/usr/bin/env python3
"""Share of a file's code lines that occur verbatim in the same-named reference file (whitespace and case normalised,
comments dropped) -- the measure the round-2 review used on the Fortran host layer.  Build container only (reads
/root/reference).   python3 tools/linematch.py [-v FILE]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"


def norm(path):
    out = []
    for ln in open(path, errors="replace"):
        s = ln.split("!")[0] if "'" not in ln and '"' not in ln else re.sub(r"!(?=[^'\"]*$).*", "", ln)
        s = re.sub(r"\s+", "", s).lower()
        if s and not s.startswith("#"):
            out.append(s)
    return out


def main():
    d = os.path.join(ROOT, "cdftools_amd", "fortran")
    rows = []
    for f in sorted(os.listdir(d)):
        r = os.path.join(REF, f)
        if not os.path.isfile(os.path.join(d, f)) or not os.path.exists(r):
            continue
        mine, ref = norm(os.path.join(d, f)), set(norm(r))
        hit = [x for x in mine if x in ref]
        rows.append((f, len(hit), len(mine)))
        if "-v" in sys.argv and f == sys.argv[sys.argv.index("-v") + 1]:
            for x in hit:
                print("   ", x)
    for f, h, n in rows:
        print(f"{f:22s} {h:5d} / {n:5d} = {100.0 * h / max(n, 1):5.1f} %")


if __name__ == "__main__":
    main()
