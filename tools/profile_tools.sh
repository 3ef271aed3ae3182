#!/bin/bash
# Per-tool lines (HIP events) and the rocprofv3 kernel statistics of the same command for every kernel
# other than cdfmoc's; run on the GPU box from the repo root:  bash tools/profile_tools.sh <grid>
set -u
GRID=${1:-eorca12}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/tools_$GRID
mkdir -p "$OUT"
TOOLS=heatc,mean,zonal,mhst,psi,vt,curl,eos,mocsig
python3 "$REPO/tools/bench_tools.py" --grid "$GRID" --tools $TOOLS > "$OUT/lines.jsonl" 2> "$OUT/lines.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/bench_tools.py" --grid "$GRID" --tools $TOOLS --reps 5 > "$OUT/trace.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" | head
