#!/bin/bash
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r04/pytest_gpu_full.txt | tail -3
python3 tools/bench_tools.py --grid eorca025 --reps 20 --tools heatc,mean,zonal,mhst,psi,vt,curl,eos,mocsig > gpurun_out/r04/tools_eorca025_n1.jsonl 2> gpurun_out/r04/tools_eorca025.err
python3 tools/bench_tools.py --grid eorca12 --reps 10 --tools heatc,mean,zonal,mhst,psi,vt,curl,eos,mocsig > gpurun_out/r04/tools_eorca12_n1.jsonl 2> gpurun_out/r04/tools_eorca12.err
python3 - <<'PY'
import json
for f in ("gpurun_out/r04/tools_eorca025_n1.jsonl","gpurun_out/r04/tools_eorca12_n1.jsonl"):
    for l in open(f):
        try: d=json.loads(l)
        except Exception: continue
        print(d["grid"], "|", d["tool"][:60], "|", d["ms"], d["frac_of_hbm_peak"], d.get("kernels_only_ms"), d.get("kernels_only_frac"))
PY
