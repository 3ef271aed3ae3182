#!/usr/bin/env python3
"""Turns the rocprofv3 PMC passes of tools/profile_moc.sh into profiles/moc_zonal_traffic.json, the file bench.py takes
`roofline.traffic` from -- with the IDENTITY of what was measured (plan.describe() and the block shape, read from the bench
line of the same call), so that a later kernel change or another block shape gets `traffic: null` instead of an old number.

    python3 tools/pmc_traffic.py gpurun_out/prof_<tag> [--out profiles/moc_zonal_traffic.json] [--source "text"]

Counters as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in separate passes, KB -> bytes x 1024,
FETCH_SIZE doubled on gfx950 for 16-byte-per-lane streaming reads, WRITE_SIZE as it is; per launch of the zonal kernel."""
import csv
import glob
import json
import os
import sys


def counter_per_launch(d, name):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name and "moc_zonal" in row["Kernel_Name"]:
                tot += float(row["Counter_Value"])
                n += 1
    if n == 0:
        raise SystemExit(f"no {name} rows of a moc_zonal kernel under {d}")
    return tot / n, n


def bench_line(log):
    for ln in open(log):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"no JSON line in {log}")


def main():
    d = sys.argv[1]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                           "profiles", "moc_zonal_traffic.json")
    source = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else d
    fetch_kb, nf = counter_per_launch(os.path.join(d, "pmc_fetch"), "FETCH_SIZE")
    write_kb, nw = counter_per_launch(os.path.join(d, "pmc_write"), "WRITE_SIZE")
    line = bench_line(os.path.join(d, "pmc_fetch.log"))
    rf = line["roofline"]
    wl = line["config"]["workload"].split()                   # "cdfmoc eorca12 4322x3606x75, 12 records, 5 basins"
    npi, npj, npk = (int(x) for x in wl[2].rstrip(",").split("x"))
    npt = int(wl[3])
    fetch = 2.0 * 1024.0 * fetch_kb
    write = 1024.0 * write_kb
    j = {"workload": wl[1], "n_gpus": line["n_gpus"], "describe": rf["kernel"], "block_shape": [npi, rf["rows"], npk, npt],
         "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb, "launches_in_the_passes": [nf, nw],
         "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
         "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
         "ratio_to_algorithmic": (fetch + write) / rf["algorithmic_bytes_per_launch"],
         "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_moc.sh); KB -> bytes x1024; "
                   "FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for 16-B-per-lane streaming reads on gfx950; "
                   "WRITE_SIZE taken as is; mean over the launches of the moc_zonal kernel in each pass",
         "source": source}
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j, indent=1))


if __name__ == "__main__":
    main()
