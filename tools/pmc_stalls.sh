#!/bin/bash
# Issue / stall counters of one kernel (rocprofv3 --pmc, counters alone, a few passes of four counters).  On the GPU box:
#     bash tools/pmc_stalls.sh <kernel-name-substring> <tag> <python script and its arguments ...>
# e.g. bash tools/pmc_stalls.sh heatc_scan_kernel heatc tools/bench_tools.py --grid eorca12 --tools heatc --reps 3
# Prints the mean per dispatch of every counter and writes gpurun_out/stalls_<tag>.json.
set -u
PAT=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stalls_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-48)
  rocprofv3 --pmc $set --output-format csv -d "$OUT/$tag" -- python3 "$R/$1" "${@:2}" > "$OUT/$tag.log" 2>&1
done
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, collections, json, sys
out, pat = sys.argv[1:3]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
if "SQ_BUSY_CYCLES" in res:
    cyc = res["SQ_BUSY_CYCLES"] / 32.0
    res["_kernel_cycles"] = cyc
    if "SQ_ACTIVE_INST_VALU" in res: res["_valu_busy"] = res["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0)
    if "SQ_WAVE_CYCLES" in res: res["_waves_per_simd"] = res["SQ_WAVE_CYCLES"] * 4.0 / (cyc * 1024.0)
json.dump({"kernel": pat, "per_dispatch": res}, open(out + ".json", "w"), indent=1)
for k, v in res.items(): print(k, "%.4g" % v)
PY
