#!/bin/sh
# Compiles the parts of the reference that build from their own sources alone, where they lie
# under /root/reference, into oracle/_ref/ (git-ignored; travels to the GPU box with gpurun).
# Only src/eos.f90 qualifies: every cdf*.f90 program needs netcdf-fortran, which this image lacks,
# and is therefore treated as unbuildable (DESIGN.md "Oracle").
set -e
cd "$(dirname "$0")"
mkdir -p _ref
echo "oracle/_ref: nothing to build yet"
