#!/bin/sh
# Compiles the parts of the reference that build from their own sources alone, where they lie
# under /root/reference, into oracle/_ref/ (git-ignored; travels to the GPU box with gpurun).
# Only src/eos.f90 qualifies: every cdf*.f90 program needs netcdf-fortran, which this image lacks,
# and is therefore treated as unbuildable (DESIGN.md "Oracle").
# Flags: -O1 mirrors the reference's plain -O (Macrolib/macro.gfortran:26); -ffp-contract=off
# mirrors an x86-64 baseline build without FMA contraction.
set -e
cd "$(dirname "$0")"
REF=/root/reference/src
FC=${FC:-amdflang}
mkdir -p _ref
if [ ! -f _ref/libeos_ref.so ] || [ eos_ref_driver.f90 -nt _ref/libeos_ref.so ] || [ $REF/eos.f90 -nt _ref/libeos_ref.so ]; then
  $FC -O1 -fPIC -ffp-contract=off -module-dir _ref -c $REF/eos.f90 -o _ref/eos.o
  $FC -O1 -fPIC -ffp-contract=off -module-dir _ref -I_ref -c eos_ref_driver.f90 -o _ref/eos_ref_driver.o
  $FC -shared _ref/eos.o _ref/eos_ref_driver.o -o _ref/libeos_ref.so
  echo "oracle/_ref/libeos_ref.so built from $REF/eos.f90"
else
  echo "oracle/_ref/libeos_ref.so up to date"
fi

# Harnesses around literal statements of the six loop programs (ref_lift.py explains what they are and are not):
# generated into _ref/, compiled with the same flags, linked with the reference's eos module (cdfmoc -decomp calls sigmai).
if [ ! -f _ref/liblift_ref.so ] || [ ref_lift.py -nt _ref/liblift_ref.so ] || [ ref_build.sh -nt _ref/liblift_ref.so ]; then
  # the generated sources hold reference text: they live in a scratch directory for the time of the compilation only
  # (KEEP_LIFT_SRC=1 leaves them in _ref/ for inspection; _ref/ is git-ignored either way)
  if [ -n "$KEEP_LIFT_SRC" ]; then gen=_ref; else gen=$(mktemp -d /tmp/cdft_lift.XXXXXX); fi
  python3 ref_lift.py --out "$gen"
  objs=""
  for t in cdfmoc cdfheatc cdfmhst cdfpsi cdfvt cdfcurl cdfvhst cdfvtrp cdfsum cdfmean cdfvertmean cdfvint cdfzonalsum cdfzonalmean cdfmocsig cdfsig0 cdfsigi cdfmoc_rapid cdfmaxmoc; do
    $FC -O1 -fPIC -ffp-contract=off -module-dir _ref -I_ref -c "$gen"/lift_$t.f90 -o _ref/lift_$t.o
    objs="$objs _ref/lift_$t.o"
  done
  $FC -shared _ref/eos.o $objs -o _ref/liblift_ref.so
  [ -n "$KEEP_LIFT_SRC" ] || rm -rf "$gen"
  echo "oracle/_ref/liblift_ref.so built from literal statements of 18 programs (19 harnesses: cdfmoc also gives its -rapid branch) under $REF"
else
  echo "oracle/_ref/liblift_ref.so up to date"
fi
