cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tcc
rm -rf $OUT; mkdir -p $OUT
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_TAG_STALL_sum TCC_EA_RDREQ_LEVEL_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_BUBBLE_sum TCC_EA_RD_UNCACHED_32B_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/tools/bench_tools.py --grid eorca12 --tools heatc,mhst,psi --reps 3 > $OUT/$tag.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        name = "heatc_level" if "heatc_level" in k else ("vint<4,1>" if "vint_kernel<4, 1" in k else ("vint<4,0>" if "vint_kernel<4, 0" in k else None))
        if name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,c in acc.items():
    print(n, {k: "%.4g"%(sum(v)/len(v)) for k,v in sorted(c.items())})
PY
grep -l "rror" $OUT/*.log | head
