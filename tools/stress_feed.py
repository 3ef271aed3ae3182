#!/usr/bin/env python3
"""Every program run of tests/test_cli_feed.py (the feed on, a ring of two one-plane slots: every slot wraps many times per field), over
and over for STRESS_SECONDS (default 240), each with a 30 s limit (STRESS_FEED=0: the same calls on whole records in pageable memory).  A run that does not end is described -- the state and the
system call of each of its threads from /proc, a backtrace when rocgdb may attach -- before it is killed: a rare hang (one program
run in about a thousand, seen once in round 5) names itself here.       python tools/stress_feed.py        (on the GPU box)"""
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lift_cases as lc  # noqa: E402
import nemo_files  # noqa: E402
import test_cli_feed as tf  # noqa: E402

BIN = os.path.join(ROOT, "cdftools_amd", "bin")


from run_program import describe  # noqa: E402  (what every thread of a live process is doing)


def main():
    seconds = float(os.environ.get("STRESS_SECONDS", "240"))
    ring = {"CDFT_FEED_MB": "0.0001", "CDFT_FEED_SLOTS": "2"} if os.environ.get("STRESS_RING", "small") == "small" else {}
    d = tempfile.mkdtemp(prefix="cdft_stress_")
    g, rec = nemo_files.make_case(d, npi=26, npj=19, npk=6, npt=3)
    e3 = {p: lc.e3_records(g[f"e3{p}_0"], 3, 8100 + i) for i, p in enumerate("tuv")}
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridV.nc"), g, "depthv", {"vomecrty": rec["v"], "e3v": e3["v"]})
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridU.nc"), g, "depthu", {"vozocrtx": rec["u"], "e3u": e3["u"]})
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridT.nc"), g, "deptht", {"votemper": rec["t"], "vosaline": rec["s"], "e3t": e3["t"]}, fill=0.0)
    nemo_files.write_data(os.path.join(d, "EIV_y0001_gridV.nc"), g, "depthv", {"vomecrty": rec["v"], "vomeeivv": (rec["v"][::-1] * 0.3).astype("float32")})
    nemo_files.write_data(os.path.join(d, "mxl.nc"), g, "deptht", {"somxl010": (5 + 200 * abs(rec["t"][:, 0]) / 30).astype("float32")})
    nemo_files.write_data(os.path.join(d, "ssh.nc"), g, "deptht", {"sossheig": (rec["t"][:, 0] / 30 - 0.5).astype("float32")})
    subprocess.run([os.path.join(BIN, "cdfvT"), "-c", "SYN", "-l", "y0001", "-o", "vt_keep.nc"], cwd=d, env=dict(os.environ, CDFT_FEED="0"), capture_output=True)
    t0, runs, hung, slowest = time.time(), 0, [], (0.0, None)
    while time.time() - t0 < seconds:
        for tool, args, files, *more in tf.RUNS:
            env = dict(os.environ, CDFT_FEED=os.environ.get("STRESS_FEED", "1"), **ring, **(more[0] if more else {}))
            p = subprocess.Popen([os.path.join(BIN, tool)] + args, cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t1 = time.time()
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                what = describe(p.pid)
                p.kill(); p.wait()
                hung.append({"tool": tool, "args": args, "after_runs": runs})
                print("HUNG:", tool, " ".join(args), "\n" + what, flush=True)
            dt = time.time() - t1
            if dt > slowest[0]:
                slowest = (dt, tool + " " + " ".join(args))
            runs += 1
            if time.time() - t0 >= seconds:
                break
    print(json.dumps({"program_runs": runs, "seconds": round(time.time() - t0, 1), "hung": hung, "slowest_s": round(slowest[0], 2), "slowest": slowest[1],
                      "ring": ring, "feed": os.environ.get("STRESS_FEED", "1")}))
    sys.exit(1 if hung else 0)


if __name__ == "__main__":
    main()
