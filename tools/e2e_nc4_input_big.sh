#!/bin/bash
# NetCDF-4 input at the headline grid: cdfsigi -nc4 writes a deflated 4.7 GB field (2.1 GB in the file), cdfmean and cdfzonalmean read it back through the
# feed (chunks inflated by the I/O threads) and must answer what they answer for the classic file of the same field.  On the GPU box:  bash tools/e2e_nc4_input_big.sh
D=/tmp/big
R=$(pwd)
python tools/e2e_all.py --grid eorca12 --dir $D --only cdfmaxmoc > /dev/null 2>&1
cd $D
B=$R/cdftools_amd/bin
$B/cdfsigi -t SYN_y0001_gridT.nc -r 2000 -nc4 -o sigi4.nc > /dev/null 2>&1; echo "cdfsigi -nc4 rc=$? $(ls -la sigi4.nc | awk '{print $5}') bytes"
$B/cdfsigi -t SYN_y0001_gridT.nc -r 2000 -o sigi3.nc > /dev/null 2>&1; echo "cdfsigi rc=$? $(ls -la sigi3.nc | awk '{print $5}') bytes"
for f in sigi4.nc sigi3.nc; do
  t0=$(date +%s%N); CDFT_TIMING=1 $B/cdfmean -f $f -v vosigmai -p T > /tmp/o.out 2> /tmp/o.err; rc=$?; t1=$(date +%s%N)
  echo "== cdfmean -f $f rc=$rc wall $(( (t1 - t0) / 1000000 )) ms  $(grep 'Mean value over the ocean' /tmp/o.out | head -1)  $(grep cdfio /tmp/o.err | tr -s ' ')"
  t0=$(date +%s%N); CDFT_TIMING=1 $B/cdfzonalmean -f $f -p T -l vosigmai -o zm_$f > /tmp/o.out 2> /tmp/o.err; rc=$?; t1=$(date +%s%N)
  echo "== cdfzonalmean -f $f rc=$rc wall $(( (t1 - t0) / 1000000 )) ms $(grep cdfio /tmp/o.err | tr -s ' ')"
done
cmp zm_sigi4.nc zm_sigi3.nc && echo "zonal means of the NetCDF-4 and the classic input: identical files"
