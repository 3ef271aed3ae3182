#!/usr/bin/env python3
"""cdfmoc (the program) on FILES of the headline grid, eORCA12 4322 x 3606 x 75: one 3-D field is 4.68 GB, more than
the 64-bit-offset format holds per record, so the files are CDF-5 written through the bundled backend (ncc_* over
ctypes; scipy cannot write them), level by level to keep the generator's memory small.  Only what cdfmoc reads is
written: mesh_hgr.nc (e1v, gphiv), mesh_zgr.nc (e3v_0, gdepw_1d; e3t_0 defined, not written), mask.nc (vmask),
new_maskglo.nc, V.nc (--npt records).  Sampled rows of moc.nc are checked against the oracle.

    python tools/e2e_cdfmoc_big.py [--grid eorca12] [--npt 2]      # on the GPU box; ~25 GB of scratch files
"""
import argparse
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
import oracle_lib  # noqa: E402
import synth  # noqa: E402
from test_cli import read_nc  # noqa: E402

GRIDS = {"tiny": (64, 40, 7), "eorca025": (1442, 1207, 75), "eorca12": (4322, 3606, 75)}
sys.path.insert(0, os.path.join(ROOT, "tools"))
from big_files import BYTE, CHAR, FLOAT, DOUBLE, Writer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="eorca12", choices=sorted(GRIDS))
    ap.add_argument("--npt", type=int, default=2)
    ap.add_argument("--rows", type=int, default=5)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--rocprof", default="", help="directory: also run the program once under rocprofv3 --kernel-trace --stats")
    args = ap.parse_args()
    npi, npj, npk = GRIDS[args.grid]
    npt = args.npt
    d = tempfile.mkdtemp(prefix="cdfmoc_big_", dir=os.environ.get("TMPDIR", "/tmp"))
    rng = np.random.default_rng(20240601)
    f32 = np.float32
    t0 = time.perf_counter()
    rows = np.sort(rng.choice(np.arange(1, npj - 1), args.rows, replace=False))

    e1v = (2000 + 26000 * rng.random((npj, npi), dtype=f32)).astype(f32)
    gphiv = np.broadcast_to(np.linspace(-78.0, 89.9, npj, dtype=f32)[:, None], (npj, npi)).copy()
    e31d = synth.e3_profile(npk)
    gdepw = np.concatenate([[0.0], np.cumsum(e31d[:-1], dtype=np.float64)]).astype(f32)
    mbathy = np.where(rng.random((npj, npi)) < 0.35, 0, 1 + np.floor((npk - 1) * rng.random((npj, npi))).astype(np.int32))
    mbathy = np.minimum(mbathy, npk - 1).astype(np.int32)
    mbathy[0, :] = 0
    band = (np.arange(npi) * 3) // npi

    def tmask(k):
        return (k < mbathy).astype(np.int8)

    def vmask(k):
        t = tmask(k)
        v = np.zeros_like(t)
        v[:-1] = t[:-1] * t[1:]
        return v

    w = Writer(os.path.join(d, "mesh_hgr.nc"), [("t", None), ("y", npj), ("x", npi)])
    for nm in ("e1v", "gphiv"):
        w.define(nm, FLOAT, ("t", "y", "x"))
    w.enddef()
    w.put("e1v", (0, 0, 0), e1v)
    w.put("gphiv", (0, 0, 0), gphiv)
    w.close()

    t1 = tmask(0)
    w = Writer(os.path.join(d, "new_maskglo.nc"), [("y", npj), ("x", npi)])
    for nm in ("tmaskatl", "tmaskpac", "tmaskind"):
        w.define(nm, BYTE, ("y", "x"))
    w.enddef()
    basin = {"tmaskatl": t1 * (band == 0)[None], "tmaskpac": t1 * (band == 1)[None], "tmaskind": t1 * (band == 2)[None]}
    for nm, a in basin.items():
        w.put(nm, (0, 0), a.astype(np.int8))
    w.close()

    wz = Writer(os.path.join(d, "mesh_zgr.nc"), [("t", None), ("z", npk), ("y", npj), ("x", npi)])
    wz.define("e3t_0", FLOAT, ("t", "z", "y", "x"))          # present (it selects the v3.6 layout), never read by cdfmoc
    wz.define("e3v_0", FLOAT, ("t", "z", "y", "x"))
    wz.define("gdepw_1d", FLOAT, ("t", "z"))
    wz.enddef()
    wz.put("gdepw_1d", (0, 0), gdepw)
    wm = Writer(os.path.join(d, "mask.nc"), [("t", None), ("z", npk), ("y", npj), ("x", npi)])
    wm.define("vmask", BYTE, ("t", "z", "y", "x"))
    wm.enddef()
    wv = Writer(os.path.join(d, "V.nc"), [("time_counter", None), ("depthv", npk), ("y", npj), ("x", npi)])
    wv.define("vomecrty", FLOAT, ("time_counter", "depthv", "y", "x"), units="m/s")
    wv.define("time_counter", DOUBLE, ("time_counter",), units="seconds since 1950-01-01 00:00:00", calendar="noleap",
              time_origin="1950-JAN-01 00:00:00")
    wv.define("depthv", FLOAT, ("depthv",), units="m")
    wv.enddef()
    wv.put("depthv", (0,), (gdepw + 0.5 * e31d).astype(f32))
    wv.put("time_counter", (0,), np.arange(1, npt + 1, dtype=np.float64) * 86400.0 * 30)
    e3v_rows = np.zeros((npk, args.rows, npi), f32)
    v_rows = np.zeros((npt, npk, args.rows, npi), f32)
    vm0 = None
    for k in range(npk):
        vm = vmask(k)
        if k == 0:
            vm0 = vm
        e3 = np.full((npj, npi), e31d[k], f32)
        bottom = mbathy == k + 1                                  # partial step: the last wet cell is thinner
        e3[bottom] *= (f32(0.2) + f32(0.8) * rng.random(int(bottom.sum()), dtype=f32)).astype(f32)
        e3v = e3.copy()
        e3v[:-1] = np.minimum(e3[:-1], e3[1:])
        wz.put("e3v_0", (0, k, 0, 0), e3v)
        wm.put("vmask", (0, k, 0, 0), vm)
        e3v_rows[k] = (e3v[rows] * vm[rows].astype(f32)).astype(f32)          # get_e3v: e3v * vmask
        for t in range(npt):
            v = ((rng.random((npj, npi), dtype=f32) - f32(0.5)) * vm).astype(f32)
            wv.put("vomecrty", (t, k, 0, 0), v)
            v_rows[t, k] = v[rows]
    wz.close(); wm.close(); wv.close()
    gen = time.perf_counter() - t0
    sizes = {f: round(os.path.getsize(os.path.join(d, f)) / 1e9, 2) for f in sorted(os.listdir(d))}

    exe = os.path.join(ROOT, "cdftools_amd", "bin", "cdfmoc")

    plan_lines = {}

    def run_program(feed, out, more=None, key=None):
        """-> wall time, the phases line, stdout without it, the phases as numbers"""
        env = dict(os.environ, CDFT_TIMING="1", CDFT_FEED="1" if feed else "0", **(more or {}))
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-v", "V.nc", "-o", out], cwd=d, env=env, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        plan_lines[key or ("feed" if feed else "whole_records")] = [ln.strip() for ln in r.stderr.splitlines() if "CDFT_PLAN" in ln or "cdft_moc" in ln]
        if r.returncode != 0:
            print("cdfmoc failed with code", r.returncode)
            print(r.stdout[-3000:], r.stderr[-2000:])
            sys.exit(1)
        line = [ln.strip() for ln in r.stdout.splitlines() if "CDFT_TIMING" in ln][0]
        rest = "\n".join(ln for ln in r.stdout.splitlines() if "CDFT_TIMING" not in ln)
        nums = [float(x) for x in line.replace("(H2D + kernels + D2H)", "").split() if x.replace(".", "", 1).isdigit()]
        return round(wall, 3), line, rest, dict(zip(("setup", "mesh_reads", "data_reads", "gpu_calls", "output"), nums))

    # the old way (whole records in pageable memory, read THEN transferred) and the feed (slabs through the page-locked ring,
    # the transfer of one under the read of the next), each twice (the second run has every file in the page cache)
    e2e = {}
    outs = {}
    for feed in (False, True):
        best = None
        for i in range(2):
            wall, line, rest, ph = run_program(feed, "moc_feed.nc" if feed else "moc.nc")
            if best is None or wall < best[0]:
                best = (wall, line, ph)
        outs[feed] = rest
        wall, line, ph = best
        e2e["feed" if feed else "whole_records"] = {
            "wall_s": wall, "phases_s": ph, "per_record_s": round((ph["data_reads"] + ph["gpu_calls"] - (0 if feed else 0)) / npt, 4),
            "grid_cells_per_s": npi * npj * npk * npt / wall, "phases_line": line}
    # the feed's other routes beside the default (stream: windows of levels, every record of the run per launch, nothing 3-D
    # resident): record by record (a launch per record on a dense record, the weights read again every time) and the resident
    # pool (all records in HBM, ONE launch: bench.py's call -- and a 61 GB allocation at eORCA12 x 12)
    for key, more in (("feed_record_by_record", {"CDFT_MOC_ROUTE": "record"}), ("feed_resident_pool", {"CDFT_MOC_ROUTE": "pool"})):
        best = None
        for i in range(2):
            wall, line, rest, ph = run_program(True, f"moc_{key}.nc", more, key)
            if best is None or wall < best[0]:
                best = (wall, line, ph)
        e2e[key] = {"wall_s": best[0], "phases_s": best[2], "phases_line": best[1], "grid_cells_per_s": npi * npj * npk * npt / best[0],
                    "file_identical_to_feed": open(os.path.join(d, f"moc_{key}.nc"), "rb").read() == open(os.path.join(d, "moc_feed.nc"), "rb").read()}
    wall, line, rest, ph = run_program(True, "moc_feed_2d.nc", {"CDFT_FEED_ROWS": "2d"}, "feed_pitched_copies")   # hipMemcpy2DAsync per slab
    e2e["feed_pitched_copies"] = {"wall_s": wall, "phases_s": ph, "phases_line": line,
                                  "file_identical_to_feed": open(os.path.join(d, "moc_feed_2d.nc"), "rb").read() == open(os.path.join(d, "moc_feed.nc"), "rb").read()}
    e2e["plan_lines"] = plan_lines
    if args.rocprof:                                       # the PROGRAM under rocprofv3: the zonal kernel's launches and durations
        import csv
        import glob
        os.makedirs(args.rocprof, exist_ok=True)
        e2e["rocprof_of_the_program"] = {}
        touched = float(npi) * npj * (npk - 1) * npt
        for route in ("stream", "pool", "record"):
            out_dir = os.path.join(args.rocprof, route)
            env = dict(os.environ, CDFT_FEED="1", TMPDIR="/tmp", CDFT_MOC_ROUTE=route)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "--", exe, "-v", "V.nc", "-o",
                                "moc_prof.nc"], cwd=d, env=env, capture_output=True, text=True)
            ks = {}
            for f in glob.glob(os.path.join(out_dir, "**", "*kernel_stats.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    ks[row["Name"][:120]] = {"calls": int(row["Calls"]), "total_ms": float(row["TotalDurationNs"]) / 1e6,
                                             "avg_ms": float(row["AverageNs"]) / 1e6}
            zon = [v for k, v in ks.items() if "moc_zonal" in k]
            zms = sum(v["total_ms"] for v in zon)
            # algorithmic bytes of the run's zonal launches: the records once, the weights once per launch that reads them
            wreads = 1 if route != "record" else npt
            alg = 4.0 * touched * (1.0 + wreads / npt)
            e2e["rocprof_of_the_program"][route] = {"rc": r.returncode, "kernels": ks, "zonal_launches": sum(v["calls"] for v in zon),
                                                    "zonal_total_ms": zms, "algorithmic_GB_of_the_launches": alg / 1e9,
                                                    "zonal_frac_of_8TBps": (alg / (zms * 1e-3) / 8e12) if zms > 0 else None}
    e2e["stdout_identical"] = outs[False] == outs[True].replace("moc_feed.nc", "moc.nc")
    e2e["moc_files_identical"] = open(os.path.join(d, "moc.nc"), "rb").read() == open(os.path.join(d, "moc_feed.nc"), "rb").read()
    walls, phases = [e2e["whole_records"]["wall_s"], e2e["feed"]["wall_s"]], e2e["feed"]["phases_line"]

    ib = np.zeros((args.rows, npi, 5), np.int16)                  # src/cdfmoc.f90:325-336
    ib[:, :, 0] = vm0[rows]
    ib[:, :, 1], ib[:, :, 3], ib[:, :, 4] = basin["tmaskatl"][rows], basin["tmaskind"][rows], basin["tmaskpac"][rows]
    ib[:, :, 2] = np.minimum(ib[:, :, 4] + ib[:, :, 3], 1)
    ib[:, 0, 0] = 0
    ib[:, npi - 1, 0] = 0
    ref = oracle_lib.moc(e1v[rows], e3v_rows, ib, v_rows)         # (npt,npk,rows,nb)
    var, _, dims = read_nc(os.path.join(d, "moc_feed.nc"))
    worst, ident = 0.0, 0
    for b, nm in enumerate(("zomsfglo", "zomsfatl", "zomsfinp", "zomsfind", "zomsfpac")):
        got = var[nm][:, :, rows, 0]
        want = ref[..., b].astype(f32)
        ident += int(np.array_equal(got, want))
        worst = max(worst, float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30)))
    print(json.dumps({"tool": "cdfmoc (program, CDF-5 files)", "grid": args.grid, "npt": npt, "files_GB": sizes, "generate_s": round(gen, 1),
                      "wall_s": walls, "phases_last": phases, "grid_cells_per_s": npi * npj * npk * npt / walls[-1], "e2e": e2e,
                      "rows_checked": [int(x) for x in rows], "basins_bit_identical_after_r4_store": ident,
                      "max_rel_diff_vs_oracle": worst, "moc_dims": {k: (int(v) if v is not None else None) for k, v in dims.items()}}))
    if not args.keep:
        shutil.rmtree(d, ignore_errors=True)
    sys.exit(0 if worst < 1e-6 else 2)


if __name__ == "__main__":
    main()
