#!/bin/bash
# cdfcurl: the order of the tile kernel's workgroups, measured (round-4 verdict, item 5): the row bands of round 4 (default at
# eORCA12), the plain (level chunk, row, tile) order with 1 .. nlev chunks -- CDFT_CURL_CHUNKS=75 is the flat LEVEL-MAJOR sweep,
# one level per workgroup, the metric planes asked of L2 / Infinity Cache by every workgroup (plain loads; u, v nontemporal) --
# on one grid.  Usage (GPU box, repo root):  bash tools/curl_sweep.sh eorca025 [reps]   -> gpurun_out/r05/curl_sweep_<grid>.txt
set -u
GRID=${1:-eorca025}; REPS=${2:-50}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05; mkdir -p "$OUT"
F=$OUT/curl_sweep_$GRID.txt; : > "$F"
run() {  # label, env...
  local label=$1; shift
  local ms
  ms=$(env "$@" python3 "$R/tools/bench_tools.py" --grid "$GRID" --tools curl --reps "$REPS" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms'], d['frac_of_hbm_peak'])")
  echo "$label  $ms" | tee -a "$F"
}
run "default" CDFT_NOOP=1
for lc in 1 3 5 8 15 25 38 75; do run "band=0 chunks=$lc" CDFT_CURL_BAND=0 CDFT_CURL_CHUNKS=$lc; done
for b in 32 64 128 256; do for lc in 5 15 25 75; do run "band=$b chunks=$lc" CDFT_CURL_BAND=$b CDFT_CURL_CHUNKS=$lc; done; done
# the persistent order: n workgroups per CU (0: what the occupancy calculator says fits)
for n in 0 2 4 6 8; do run "persist=$n" CDFT_CURL_PERSIST=$n; done
