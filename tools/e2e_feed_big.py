#!/usr/bin/env python3
"""cdfpsi, cdfmhst, cdfvT or cdfmocsig (the programs) on FILES of the eORCA12 grid with and without the host feed: the two routes to the GPU
(whole records in pageable memory, read THEN copied; slabs through the page-locked ring) must print the same and write the
same file, at sizes where a record is 4.68 GB -- slab sizes, offsets and counts beyond 2^31.  CDF-5 files written level by
level through the bundled backend; only what the program reads.  (cdfmoc and cdfheatc have their own tools, which also hold
the result against the oracle.)

    python tools/e2e_feed_big.py --tool psi|mhst|vt|mocsig|sig0|sigi|curl|mean|zonal|vint|vertmean|vhst|vtrp [--grid eorca12] [--npt 2]      # on the GPU box; 20-30 GB of scratch files
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
from e2e_cdfmoc_big import BYTE, DOUBLE, FLOAT, GRIDS, Writer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tool", required=True, choices=["psi", "mhst", "vt", "mocsig", "sig0", "sigi", "curl", "mean", "zonal", "vint", "vertmean", "vhst", "vtrp"])
    ap.add_argument("--runs", type=int, default=2, help="runs per route (the first warms the page cache)")
    ap.add_argument("--grid", default="eorca12", choices=sorted(GRIDS))
    ap.add_argument("--npt", type=int, default=2)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    npi, npj, npk = GRIDS[args.grid]
    npt = args.npt
    d = tempfile.mkdtemp(prefix="cdf%s_big_" % args.tool, dir=os.environ.get("TMPDIR", "/tmp"))
    rng = np.random.default_rng(20240603)
    f32 = np.float32
    t0 = time.perf_counter()
    e31d = synth.e3_profile(npk)
    gdepw = np.concatenate([[0.0], np.cumsum(e31d[:-1], dtype=np.float64)]).astype(f32)
    mbathy = np.where(rng.random((npj, npi)) < 0.35, 0, 1 + np.floor((npk - 1) * rng.random((npj, npi))).astype(np.int32))
    mbathy = np.minimum(mbathy, npk - 1).astype(np.int32)
    lon = np.broadcast_to(np.linspace(-180, 180, npi, dtype=f32)[None, :], (npj, npi)).copy()
    lat = np.broadcast_to(np.linspace(-78, 89.9, npj, dtype=f32)[:, None], (npj, npi)).copy()
    metric = lambda: (2000 + 26000 * rng.random((npj, npi), dtype=f32)).astype(f32)   # noqa: E731

    w = Writer(os.path.join(d, "mesh_hgr.nc"), [("t", None), ("y", npj), ("x", npi)])
    names = ("e1v", "e2u", "glamf", "gphif", "gphiv") + (("e1u", "e2v", "e1f", "e2f") if args.tool == "curl" else ()) + (("e1t", "e2t") if args.tool in ("mean", "zonal", "vint", "vertmean") else ()) + (("gphit",) if args.tool == "zonal" else ())
    for nm in names:
        w.define(nm, FLOAT, ("t", "y", "x"))
    w.enddef()
    for nm in names:
        w.put(nm, (0, 0, 0), lon if nm == "glamf" else lat if nm.startswith("gphi") else metric())
    w.close()
    e3names = (["e3u_0"] if args.tool == "psi" else ["e3t_0"] if args.tool in ("mean", "zonal", "vint", "vertmean") else
               ["e3u_0", "e3v_0"] if args.tool in ("vhst", "vtrp") else ["e3v_0"])
    wz = Writer(os.path.join(d, "mesh_zgr.nc"), [("t", None), ("z", npk), ("y", npj), ("x", npi)])
    wz.define("e3t_0", FLOAT, ("t", "z", "y", "x"))              # present (it selects the v3.6 layout), never read here
    for e3name in e3names:
        if e3name != "e3t_0":
            wz.define(e3name, FLOAT, ("t", "z", "y", "x"))
    wz.define("e3t_1d", FLOAT, ("t", "z"))
    wz.define("gdepw_1d", FLOAT, ("t", "z"))
    wz.define("gdept_1d", FLOAT, ("t", "z"))
    wz.enddef()
    wz.put("e3t_1d", (0, 0), e31d.astype(f32))
    wz.put("gdepw_1d", (0, 0), gdepw)
    wz.put("gdept_1d", (0, 0), (gdepw + 0.5 * e31d).astype(f32))
    wm = Writer(os.path.join(d, "mask.nc"), [("t", None), ("z", npk), ("y", npj), ("x", npi)])
    masks = ("vmask", "fmask") + (("tmask",) if args.tool in ("mean", "zonal", "vint", "vertmean") else ())
    for nm in masks:
        wm.define(nm, BYTE, ("t", "z", "y", "x"))
    wm.enddef()

    def data_file(name, depname, fields):
        wf = Writer(os.path.join(d, name), [("time_counter", None), (depname, npk), ("y", npj), ("x", npi)])
        for v in fields:
            wf.define(v, FLOAT, ("time_counter", depname, "y", "x"))
        wf.define("time_counter", DOUBLE, ("time_counter",), units="seconds since 1950-01-01 00:00:00", calendar="noleap",
                  time_origin="1950-JAN-01 00:00:00")
        wf.define(depname, FLOAT, (depname,), units="m")
        wf.define("nav_lon", FLOAT, ("y", "x"))
        wf.define("nav_lat", FLOAT, ("y", "x"))
        wf.enddef()
        wf.put(depname, (0,), (gdepw + 0.5 * e31d).astype(f32))
        wf.put("time_counter", (0,), np.arange(1, npt + 1, dtype=np.float64) * 86400.0 * 30)
        wf.put("nav_lon", (0, 0), lon)
        wf.put("nav_lat", (0, 0), lat)
        return wf

    if args.tool == "vhst":                                       # the four products of cdfvT in one file
        files = {"vt.nc": data_file("vt.nc", "depthv", ["vomevt", "vomevs", "vozout", "vozous"])}
        fields = [("vt.nc", v) for v in ("vomevt", "vomevs", "vozout", "vozous")]
        cmd = ["cdfvhst", "-f", "vt.nc"]
    elif args.tool == "vtrp":
        files = {"U.nc": data_file("U.nc", "depthu", ["vozocrtx"]), "V.nc": data_file("V.nc", "depthv", ["vomecrty"])}
        fields = [("U.nc", "vozocrtx"), ("V.nc", "vomecrty")]
        cmd = ["cdfvtrp", "-u", "U.nc", "-v", "V.nc"]
    elif args.tool == "psi":
        files = {"U.nc": data_file("U.nc", "depthu", ["vozocrtx"]), "V.nc": data_file("V.nc", "depthv", ["vomecrty"])}
        fields = [("U.nc", "vozocrtx")]                          # the default path reads U alone; V.nc has to exist
        cmd = ["cdfpsi", "-u", "U.nc", "-v", "V.nc"]
    elif args.tool == "vt":                                       # one frame per file, --npt records: cdfvT averages over them
        files = {"SYN_y0001_gridT.nc": data_file("SYN_y0001_gridT.nc", "deptht", ["votemper", "vosaline"]),
                 "SYN_y0001_gridU.nc": data_file("SYN_y0001_gridU.nc", "depthu", ["vozocrtx"]),
                 "SYN_y0001_gridV.nc": data_file("SYN_y0001_gridV.nc", "depthv", ["vomecrty"])}
        fields = [("SYN_y0001_gridT.nc", "votemper"), ("SYN_y0001_gridT.nc", "vosaline"), ("SYN_y0001_gridU.nc", "vozocrtx"),
                  ("SYN_y0001_gridV.nc", "vomecrty")]
        cmd = ["cdfvT", "-c", "SYN", "-l", "y0001"]
    elif args.tool in ("mean", "vint", "vertmean"):
        files = {"T.nc": data_file("T.nc", "deptht", ["votemper"])}
        fields = [("T.nc", "votemper")]
        cmd = {"mean": ["cdfmean", "-f", "T.nc", "-v", "votemper", "-p", "T", "-var"], "vint": ["cdfvint", "-f", "T.nc"],
               "vertmean": ["cdfvertmean", "-f", "T.nc", "-l", "votemper", "-p", "T", "-zlim", "100", "1500"]}[args.tool]
    elif args.tool == "curl":                                     # every level of the record
        files = {"U.nc": data_file("U.nc", "depthu", ["vozocrtx"]), "V.nc": data_file("V.nc", "depthv", ["vomecrty"])}
        fields = [("U.nc", "vozocrtx"), ("V.nc", "vomecrty")]
        cmd = ["cdfcurl", "-u", "U.nc", "vozocrtx", "-v", "V.nc", "vomecrty", "-l", "1-%d" % (npk - 1)]
    elif args.tool in ("sig0", "sigi"):
        files = {"T.nc": data_file("T.nc", "deptht", ["votemper", "vosaline"])}
        fields = []
        cmd = ["cdfsig0", "-t", "T.nc"] if args.tool == "sig0" else ["cdfsigi", "-t", "T.nc", "-r", "2000"]
    elif args.tool in ("mocsig", "zonal"):                        # (mocsig -isodep: the three-histogram kernel); 5 basins in i-bands
        if args.tool == "zonal":
            files = {"T.nc": data_file("T.nc", "deptht", ["votemper"])}
            fields = [("T.nc", "votemper")]
            cmd = ["cdfzonalmean", "-f", "T.nc", "-p", "T", "-b", "new_maskglo.nc"]
        else:
            files = {"T.nc": data_file("T.nc", "deptht", ["votemper", "vosaline"]), "V.nc": data_file("V.nc", "depthv", ["vomecrty"])}
            fields = [("V.nc", "vomecrty")]
            cmd = ["cdfmocsig", "-v", "V.nc", "-t", "T.nc", "-r", "2000", "-isodep"]
        band = (np.arange(npi) * 3 // npi)[None, :]
        sea1 = mbathy > 0
        wb = Writer(os.path.join(d, "new_maskglo.nc"), [("y", npj), ("x", npi)])
        for nm in ("tmaskatl", "tmaskind", "tmaskpac"):
            wb.define(nm, BYTE, ("y", "x"))
        wb.enddef()
        for b, nm in enumerate(("tmaskatl", "tmaskind", "tmaskpac")):
            wb.put(nm, (0, 0), (sea1 & (band == b)).astype(np.int8))
        wb.close()
    else:
        files = {"vt.nc": data_file("vt.nc", "depthv", ["vomevt", "vomevs"])}
        fields = [("vt.nc", "vomevt"), ("vt.nc", "vomevs")]
        cmd = ["cdfmhst", "-vt", "vt.nc"]
    for k in range(npk):
        wet = (k < mbathy)
        m = wet.astype(np.int8)
        e3 = np.full((npj, npi), e31d[k], f32)
        bottom = mbathy == k + 1
        e3[bottom] *= (f32(0.2) + f32(0.8) * rng.random(int(bottom.sum()), dtype=f32)).astype(f32)
        for e3name in e3names:
            wz.put(e3name, (0, k, 0, 0), e3)
        for nm in masks:
            wm.put(nm, (0, k, 0, 0), m)
        for fn, v in fields:
            for t in range(npt):
                files[fn].put(v, (t, k, 0, 0), ((rng.random((npj, npi), dtype=f32) - f32(0.5)) * m).astype(f32))
        if args.tool in ("mocsig", "sig0", "sigi"):               # T, S smooth in i and z, as an ocean's
            x = np.linspace(0, 1, npi, dtype=f32)[None, :]
            z = f32(k / max(npk - 1, 1))
            for t in range(npt):
                files["T.nc"].put("votemper", (t, k, 0, 0), ((f32(25) * (1 - z) ** 2 + f32(2) * x + f32(0.1 * t)) * m).astype(f32))
                files["T.nc"].put("vosaline", (t, k, 0, 0), ((f32(34) + z + f32(0.5) * x) * m).astype(f32))
    wz.close(); wm.close()
    for wf in files.values():
        wf.close()
    gen = time.perf_counter() - t0
    sizes = {f: round(os.path.getsize(os.path.join(d, f)) / 1e9, 2) for f in sorted(os.listdir(d))}

    def run_program(feed, out):
        env = dict(os.environ, CDFT_TIMING="1", CDFT_FEED="1" if feed else "0")
        before = set(os.listdir(d))
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "cdftools_amd", "bin", cmd[0])] + cmd[1:] + ["-o", out], cwd=d, env=env, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            print(cmd[0], "failed with code", r.returncode)
            print(r.stdout[-3000:], r.stderr[-2000:])
            sys.exit(1)
        wrote = {}
        for f in sorted(set(os.listdir(d)) - before):            # the output file and the program's text files
            wrote[f.replace(out, "OUT")] = hashlib.sha1(open(os.path.join(d, f), "rb").read()).hexdigest()
            os.remove(os.path.join(d, f))
        return round(wall, 3), r.stdout.replace(out, "OUT"), [ln.strip() for ln in r.stderr.splitlines() if "CDFT_TIMING" in ln], wrote

    e2e, outs, wrote = {}, {}, {}
    for feed in (False, True):
        best = None
        for i in range(args.runs):                                # (the second run has every file in the page cache)
            wall, stdout, timing, w_ = run_program(feed, "out.nc")      # (the same name both ways: cdfpsi writes its command line into the file)
            if best is None or wall < best[0]:
                best = (wall, timing)
        outs[feed], wrote[feed] = stdout, w_
        e2e["feed" if feed else "whole_records"] = {"wall_s": best[0], "grid_cells_per_s": npi * npj * npk * npt / best[0], "timing": best[1]}
    e2e["stdout_identical"] = outs[False] == outs[True]
    e2e["files_identical"] = wrote[False] == wrote[True] and len(wrote[True]) > 0
    print(json.dumps({"tool": cmd[0] + " (program, CDF-5 files)", "args": " ".join(cmd[1:]), "grid": args.grid, "npt": npt, "files_GB": sizes,
                      "generate_s": round(gen, 1), "e2e": e2e, "files_written": sorted(wrote[True])}))
    if not args.keep:
        shutil.rmtree(d, ignore_errors=True)
    sys.exit(0 if e2e["stdout_identical"] and e2e["files_identical"] else 2)


if __name__ == "__main__":
    main()
