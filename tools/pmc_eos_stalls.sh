cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc2/$tag -- python3 $R/tools/fp64_cases.py --case sigma2 --land none --grid eorca12 --reps 2 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eos_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v))
PY
