#!/bin/bash
# the last launches of a command in time order with the gaps between them:  bash tools/ktrace_seq.sh <tag> <n> <python script and arguments ...>
set -u
TAG=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kseq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -- python3 "$R/$1" "${@:2}" > "$OUT/log.txt" 2>&1
python3 - "$OUT" "$N" <<'PY'
import csv, glob, re, sys
out, n = sys.argv[1], int(sys.argv[2])
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\w+_kernel(<[^>]*>)?", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:70]))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
prev = None
for s, e, name in ev[-n:]:
    print("%10.1f us  gap %8.1f  %s" % ((e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, name))
    prev = e
PY
