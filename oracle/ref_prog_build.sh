#!/bin/sh
# The reference's own PROGRAMS (the eight of SURVEY 8a and the ten of 8f), unmodified (cdfmean: with the OTHER of the two spellings its source offers
# for one OPEN statement, see below), compiled where they lie under /root/reference/src into oracle/_ref/prog/
# (git-ignored, never pushed to the GPU box: .gpurunignore).  Build container only.
#
# What this is, and is not.  The reference programs need netcdf-fortran, which the image lacks; here they are linked against
# the NetCDF layer this repository ships for its own programs (cdftools_amd/fortran/netcdf.f90 over libncclassic.so) -- a
# product component that exists whatever the reference does, not something written to make the reference build.  The
# result is a DIFFERENTIAL of the program level (option parsing, CreateOutput, the cdfio calls, the order of variables and
# attributes, stdout), used by tests/make_golden_prog.py to write tests/golden/prog_*.npz.  It is NOT a parity pin: both
# sides read and write through the same NetCDF backend, whose own check is scipy.io.netcdf_file (tests/test_ncclassic.py).
# The arithmetic pins stay oracle/ref_lift.py (statements) and src/eos.f90 (ref_build.sh).
# Flags as ref_build.sh: -O1 (the reference's plain -O), no FMA contraction; -Dkey_netcdf4 as the reference's Makefile.
set -e
cd "$(dirname "$0")"
REF=/root/reference/src
FC=${FC:-amdflang}
FB=../cdftools_amd/fortran/build
LIBDIR=$(cd ../cdftools_amd/lib && pwd)
[ -f $FB/netcdf.o ] || { echo "oracle/_ref/prog: build cdftools_amd/fortran first (its netcdf module is linked)"; exit 1; }
mkdir -p _ref/prog/obj
O=_ref/prog/obj
FLAGS="-O1 -ffp-contract=off -cpp -Dkey_netcdf4 -w"
new=0
for f in modcdfnames.F90 cdfio.F90 modutils.f90 eos.f90 cdftools.f90; do
  o=$O/${f%.*}.o
  if [ ! -f $o ] || [ $REF/$f -nt $o ] || [ $FB/netcdf.o -nt $o ]; then
    $FC $FLAGS -I$FB -I$O -module-dir $O -c $REF/$f -o $o
    new=1
  fi
done
for p in cdfmoc cdfmhst cdfvT cdfheatc cdfpsi cdfcurl cdfsig0 cdfsigi \
         cdfvhst cdfvtrp cdfmean cdfsum cdfzonalsum cdfzonalmean cdfvint cdfvertmean cdfmocsig cdfmaxmoc; do
  if [ $new = 1 ] || [ ! -f _ref/prog/$p ] || [ $REF/$p.f90 -nt _ref/prog/$p ]; then
    src=$REF/$p.f90
    if [ $p = cdfmean ]; then
      # The one exception to "unmodified".  src/cdfmean.f90:193-195 offers two spellings of one statement,
      #     OPEN(6,FORM='FORMATTED',RECL=256)  ! ifort
      #   ! OPEN(6,FORM='FORMATTED')          ! gfortran
      # and ships with the ifort one, which the flang runtime refuses ("RECL= may not be changed for an open unit": the program
      # stops before it reads an argument).  The build takes the source's OWN other spelling: the comment mark moves from line
      # 195 to line 194, nothing else changes (checked below: exactly these two lines, found where they are expected), in a
      # temporary copy that is deleted after the compilation.  Rounds 1-4 left cdfmean out of the differential instead.
      [ "$(sed -n 194p $src)" = "  OPEN(6,FORM='FORMATTED',RECL=256)  ! ifort" ] && \
      [ "$(sed -n 195p $src)" = "  ! OPEN(6,FORM='FORMATTED')          ! gfortran" ] || { echo "cdfmean.f90:194-195 are not the two OPEN lines"; exit 1; }
      src=$O/cdfmean_with_its_gfortran_line.f90
      sed -e "194s/^  OPEN/  ! OPEN/" -e "195s/^  ! OPEN/  OPEN/" $REF/$p.f90 > $src
      [ "$(diff $REF/$p.f90 $src | grep -c '^[<>]')" = 4 ] || { echo "cdfmean: more than the two lines differ"; exit 1; }
    fi
    $FC $FLAGS -I$FB -I$O -module-dir $O -c $src -o $O/$p.prog.o
    [ $p = cdfmean ] && rm -f $src
    $FC $O/$p.prog.o $O/modcdfnames.o $O/cdfio.o $O/modutils.o $O/eos.o $O/cdftools.o $FB/netcdf.o -o _ref/prog/$p \
        -L$LIBDIR -lncclassic -Wl,-rpath,$LIBDIR
  fi
done
echo "oracle/_ref/prog: the 18 reference programs of SURVEY 8(a) and 8(f), unmodified (cdfmean: its own gfortran spelling of one OPEN), over this repository's NetCDF layer (a differential, not a pin)"
