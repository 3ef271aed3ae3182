#!/usr/bin/env python3
"""Writes tests/golden/prog_<case>.npz: what the reference's OWN PROGRAMS leave behind on the files of nemo_files.make_case.

Build container only (needs /root/reference): `make -C oracle refprog` compiles the unmodified reference sources into
oracle/_ref/prog/ over this repository's NetCDF layer (oracle/ref_prog_build.sh says what that is worth: a differential of
the program level -- option parsers, CreateOutput, cdfio's packing and headers, stdout -- not a parity pin), this script runs
the sixteen cases of tests/prog_cases.py and stores exit status, stdout, and every output file's dimensions, variables,
attributes (in file order) and data.  Only inputs' seeds and the programs' OUTPUTS are committed; no reference text.

    python3 tests/make_golden_prog.py            (re)write the goldens
    python3 tests/make_golden_prog.py --check    compare a fresh run with the committed files
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import nemo_files  # noqa: E402
import prog_cases as pc  # noqa: E402


def main():
    check = "--check" in sys.argv
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "refprog"], stdout=subprocess.DEVNULL)
    bindir = os.path.join(ROOT, "oracle", "_ref", "prog")
    gold = os.path.join(HERE, "golden")
    bad = 0
    with tempfile.TemporaryDirectory(prefix="cdft_prog_") as d:
        nemo_files.make_case(d)
        for case in pc.CASES:
            meta, arrays = pc.run_case(bindir, case, d)
            path = os.path.join(gold, f"prog_{case[0]}.npz")
            missing = [o for o, s in meta["files"].items() if s is None]
            print(f"{case[0]:28s} rc={meta['rc']} files={list(meta['files'])} missing={missing} stdout={len(meta['stdout'].splitlines())} lines")
            if meta["rc"] != 0 or missing:
                print(meta["stdout"][-2000:])
                raise SystemExit(f"the reference program failed on case {case[0]}")
            if check:
                m0, a0 = pc.load_golden(path)
                same = m0 == meta and sorted(a0) == sorted(arrays) and all(np.array_equal(a0[k], arrays[k], equal_nan=True) for k in a0)
                print("   ", "equal" if same else "DIFFERENT")
                bad += not same
            else:
                pc.save_golden(path, meta, arrays)
    if check:
        print("all equal" if not bad else f"{bad} case(s) differ")
        raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
