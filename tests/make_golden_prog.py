#!/usr/bin/env python3
"""Writes tests/golden/prog_<case>.npz: what the reference's OWN PROGRAMS leave behind on the files of nemo_files.make_case.

Build container only (needs /root/reference): `make -C oracle refprog` compiles the unmodified reference sources into
oracle/_ref/prog/ over this repository's NetCDF layer (oracle/ref_prog_build.sh says what that is worth: a differential of
the program level -- option parsers, CreateOutput, cdfio's packing and headers, stdout -- not a parity pin), this script runs
the cases of tests/prog_cases.py and stores exit status, stdout, and every output file's dimensions, variables,
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
    skipped = {}
    with tempfile.TemporaryDirectory(prefix="cdft_prog_") as d:
        g_, rec_ = nemo_files.make_case(d)
        pc.extra_files(d, g_, rec_)
        for case in pc.CASES + pc.USAGE_CASES:
            meta, arrays = pc.run_case(bindir, case, d)
            path = os.path.join(gold, f"prog_{case[0]}.npz")
            missing = [o for o, s in meta["files"].items() if s is None]
            print(f"{case[0]:28s} rc={meta['rc']} files={list(meta['files'])} missing={missing} stdout={len(meta['stdout'].splitlines())} lines")
            if (meta["rc"] != 0 and not (meta["rc"] in (97, 98, 99) and not case[3])) or missing:      # (a STOP code a case expects is an outcome)
                # a reference program that does not run here AS IT IS (e.g. cdfmean opens unit 6 with RECL=, an ifort idiom
                # its own comment flags, which the flang runtime refuses) is left out of the differential, with the reason
                skipped[case[0]] = {"rc": meta["rc"], "missing": missing, "stderr_tail": meta.get("stderr", "")[-400:],
                                    "stdout_tail": meta["stdout"][-400:]}
                print("    -> left out:", (meta.get("stderr", "") or meta["stdout"]).strip().splitlines()[-1:] )
                if os.path.exists(path) and not check:
                    os.remove(path)
                continue
            if check:
                m0, a0 = pc.load_golden(path)
                same = m0 == meta and sorted(a0) == sorted(arrays) and all(np.array_equal(a0[k], arrays[k], equal_nan=True) for k in a0)
                print("   ", "equal" if same else "DIFFERENT")
                bad += not same
            else:
                pc.save_golden(path, meta, arrays)
    with tempfile.TemporaryDirectory(prefix="cdft_prog_rapid_") as d:          # cdfmoc -rapid: its own grid (tests/test_cli_rapid.py)
        import test_cli_rapid as tr
        tr.make_rapid_case(d)
        for case in pc.RAPID_CASES:
            meta, arrays = pc.run_case(bindir, case, d)
            path = os.path.join(gold, f"prog_{case[0]}.npz")
            missing = [o for o, s in meta["files"].items() if s is None]
            print(f"{case[0]:28s} rc={meta['rc']} files={list(meta['files'])} missing={missing} stdout={len(meta['stdout'].splitlines())} lines")
            if meta["rc"] != 99 or missing:
                print("    -> NOT the outcome the reference's -rapid has (STOP 99 after its file):", meta.get("stderr", "")[-300:])
                bad += 1
                continue
            meta.pop("stderr", None)                            # ("Fortran STOP: code 99": the runtime's line, not the program's)
            if check:
                m0, a0 = pc.load_golden(path)
                same = m0 == meta and sorted(a0) == sorted(arrays) and all(np.array_equal(a0[k], arrays[k], equal_nan=True) for k in a0)
                print("   ", "equal" if same else "DIFFERENT")
                bad += not same
            else:
                pc.save_golden(path, meta, arrays)
    import json
    spath = os.path.join(gold, "prog_left_out.json")
    if check:
        same = json.load(open(spath)) == skipped if os.path.exists(spath) else not skipped
        print("all equal" if not bad and same else f"{bad} case(s) differ" + ("" if same else "; the list of cases left out differs"))
        raise SystemExit(1 if bad or not same else 0)
    json.dump(skipped, open(spath, "w"), indent=1)
    print(f"{len(pc.CASES) + len(pc.USAGE_CASES) - len(skipped)} goldens written, {len(skipped)} cases left out ({', '.join(skipped) or 'none'})")


if __name__ == "__main__":
    main()
