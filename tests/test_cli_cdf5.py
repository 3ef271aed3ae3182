"""The programs on CDF-5 input (the format of files whose fields pass 4 GiB: what tools/e2e_all.py --grid eorca12 runs on), here on a
small grid so that it is part of the suite: the file set of tools/big_files.make_big_case -- written through the bundled backend level by
level, with its r8 sums of what it wrote -- read by three programs with and without the host feed (the byte masks then travel as bytes
and are widened in HBM): the same stdout and files either way, and the printed integrals against the generator's sums."""
import os
import re
import sys

import pytest

from test_cli import ROOT, run, tools  # noqa: F401  (tools is a fixture)

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _number_after(text, key):
    for ln in text.splitlines():
        if key in ln:
            m = re.search(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", ln.split(key, 1)[1])
            if m:
                return float(m.group(0))
    return None


@pytest.mark.gpu
def test_programs_on_cdf5_files_with_and_without_the_feed(tools, tmp_path):  # noqa: F811
    import big_files
    d = str(tmp_path)
    stats = big_files.make_big_case(d, 26, 19, 6, npt=2)
    assert open(os.path.join(d, "SYN_y0001_gridT.nc"), "rb").read(4) == b"CDF\x05"
    runs = [("cdfmoc", ["-v", "SYN_y0001_gridV.nc"], ["moc.nc"]),
            ("cdfheatc", ["-f", "SYN_y0001_gridT.nc"], ["heatc.nc"]),
            ("cdfmean", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "T"], ["cdfmean.nc", "cdfmean.txt"]),
            ("cdfzonalmean", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-b", "new_maskglo.nc", "-max"], ["zonalmean.nc"])]
    for tool, args, files in runs:
        res = {}
        for feed in ("0", "1"):
            rc, out, err = run(tool, args, d, env={"CDFT_FEED": feed, "CDFT_MOC_ROUTE": "record"})
            assert rc == 0, (tool, out[-400:] + err[-400:])
            res[feed] = (out, {f: open(os.path.join(d, f), "rb").read() for f in files})
            for f in files:
                os.remove(os.path.join(d, f))
        assert res["0"][0] == res["1"][0], tool
        assert res["0"][1] == res["1"][1], tool
        if tool == "cdfheatc":                                     # rau0 * rcp * sum of T e1t e2t e3t tmask (its default sums add in r4)
            got = _number_after(res["1"][0], "Total Heat content        :")
            assert abs(got - 1020.0 * 4000.0 * stats["heat"]) <= 2e-3 * abs(1020.0 * 4000.0 * stats["heat"])
        if tool == "cdfmean":
            got = _number_after(res["1"][0], "Mean value over the ocean:")
            assert abs(got - stats["tvol"] / stats["vol"]) <= 1e-6 * abs(stats["tvol"] / stats["vol"])
