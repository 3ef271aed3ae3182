#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of one command, filtered:  bash tools/kstat.sh <pattern> <tag> <python script and arguments ...>
set -u
PAT=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kstat_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/$1" "${@:2}" > "$OUT/log.txt" 2>&1
python3 - "$OUT" "$PAT" "$TAG" <<'PY'
import csv, glob, re, sys, statistics
out, pat, tag = sys.argv[1:4]
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    d = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(p in n for p in pat.split(",")):
            d.setdefault((re.search(r"\w+_kernel(<[^>]*>)?", n) or re.search(r".*", n)).group(0), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, v in d.items():
        v = v[len(v) // 3:]            # the first third: warm-up
        print("%-22s %-62s n=%3d  median %8.1f  min %8.1f us" % (tag, n, len(v), statistics.median(v), min(v)))
PY
