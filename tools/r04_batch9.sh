#!/bin/bash
# final check of the tree: lease i (driver's command), the whole GPU suite, smoke()
O=gpurun_out/r04; mkdir -p $O
bash tools/lease_probe.sh i > /dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/lease_i/bench.json")); r=d["roofline"]
print("lease i avg %.3f min %.3f max %.3f frac %.4f dense %.3f inproc %s value %.4e" % (r["avg_launch_ms"], r["min_launch_ms"], r["max_launch_ms"], r["frac"], r["dense_arrays"]["avg_launch_ms"], d["inprocess"].get("roofline",{}).get("avg_launch_ms", d["inprocess"].get("error")), d["value"]))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -q > $O/pytest_all_i.txt 2>&1; tail -4 $O/pytest_all_i.txt | head -2
