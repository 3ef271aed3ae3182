#!/bin/bash
# HBM bytes of the cdfcurl kernel variants (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes as
# MI355X_MICROARCH.md prescribes); run on the GPU box from the repo root:  bash tools/pmc_curl.sh <grid> <tag> [ENV=VAL ...]
set -u
GRID=${1:-eorca12}; TAG=${2:-default}; shift 2 || true
for kv in "$@"; do export "$kv"; done
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_curl_${GRID}_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/tools/bench_tools.py --grid $GRID --tools curl --reps 3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ARGS > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for which in ("fetch", "write"):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{which}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "curl" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0][-60:]
                tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
    for k, (v, n) in tot.items():
        # FETCH_SIZE / WRITE_SIZE count 32-byte units... the guide's correction: x64 B for FETCH on gfx950 -> report raw and both
        print(which, k, "launches", n, "raw/launch", v / n)
PY
