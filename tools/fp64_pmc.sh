#!/bin/bash
# Counter passes for the fp64-bound kernels; run on the GPU box from the repo root:   bash tools/fp64_pmc.sh [grid]
# One rocprofv3 run per case (counters alone: no trace domains), SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 per dispatch;
# then, in the build container:  python3 tools/fp64_pmc_parse.py gpurun_out/fp64_pmc   (writes profiles/r03/fp64_ops_pmc.json)
set -u
GRID=${1:-eorca12}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/fp64_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in sigma0:none sigma0_fma:none sigma0_teos10:none sigma2:none bn2:none mocsig:noise sigma0:noise sigma2:noise sigma2:contiguous mocsig:contiguous; do
  c=${spec%%:*}; l=${spec##*:}
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv \
      -d "$OUT/${c}_${l}" -- python3 $REPO/tools/fp64_cases.py --case $c --land $l --grid $GRID > "$OUT/${c}_${l}.log" 2>&1
  # how busy the vector pipeline was: instructions of every kind, the cycles spent executing them, the cycles the kernel took
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv \
      -d "$OUT/${c}_${l}" -- python3 $REPO/tools/fp64_cases.py --case $c --land $l --grid $GRID >> "$OUT/${c}_${l}.log" 2>&1
done
find "$OUT" -name "*counter_collection.csv" | head -20
