#!/bin/bash
# One fresh lease: what the box is (clocks, power cap), the driver's exact bench command, then optional extras.
# usage (through gpurun): bash tools/lease_probe.sh <tag> [extra command ...]
tag=${1:-lease}; shift
d=gpurun_out/r04/lease_$tag
mkdir -p $d
{ rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showtemp --showmemuse 2>&1 || true; } > $d/smi_before.txt
# what the box streams: the read-only ceiling of tools/hbm_probe.hip (best of its shapes), beside the kernel's own figure
if [ -x tools/bin/hbm_probe ]; then tools/bin/hbm_probe 16 > $d/hbm_probe.jsonl 2>&1; fi
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $d/bench.json 2> $d/bench.err
echo "bench rc=$?" >> $d/bench.err
{ rocm-smi --showclocks --showpower --showtemp 2>&1 || true; } > $d/smi_after.txt
if [ $# -gt 0 ]; then "$@" > $d/extra.out 2> $d/extra.err; echo "extra rc=$?" >> $d/extra.err; fi
tail -c 600 $d/bench.json
