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
