#!/bin/bash
# rocprofv3 passes for the bench workload; run on the GPU box from the repo root:
#   bash tools/profile_moc.sh <tag>          (outputs under gpurun_out/prof_<tag>/)
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2,3: PMC counters alone
# (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots, FETCH_SIZE takes 3).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# pass 1 runs bench.py exactly as the driver does (default steps: a timed region of about 2.5 s), so that the trace's
# average kernel duration and the line's roofline.avg_launch_ms describe the same sustained run; the counter passes
# only need a few launches
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --no-cpu-baseline --no-inprocess --no-dense --no-tools ${BENCH_ARGS:-} > "$OUT/trace.log" 2>&1
ARGS="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inprocess --no-dense --no-tools ${BENCH_ARGS:-}"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
# then, in the build container:  python3 tools/pmc_traffic.py gpurun_out/prof_$TAG     (writes profiles/moc_zonal_traffic.json
# with the identity -- plan.describe(), block shape -- of what these passes measured; bench.py checks it)
