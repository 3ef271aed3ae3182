#!/usr/bin/env python3
"""Generates tests/golden/eos.npz from the REAL reference: oracle/_ref/libeos_ref.so, i.e. the
unmodified /root/reference/src/eos.f90 compiled with amdflang by oracle/ref_build.sh.
Run in the build container only (the reference cannot travel); the .npz holds inputs and the
reference's outputs, nothing of its source."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_lib  # noqa: E402


def main():
    ref = oracle_lib.RefEos()
    rng = np.random.default_rng(20240601)
    shape = (18, 24)
    T = (-2.0 + 32.0 * rng.random(shape)).astype(np.float32)
    S = (30.0 + 8.0 * rng.random(shape)).astype(np.float32)
    S[rng.random(shape) < 0.25] = 0.0          # land: sigmai masks on S == 0
    S[0, 0], T[0, 0] = 35.5, 3.0               # the reference's documented check point (src/eos.f90:817,:820)
    S[0, 1], T[0, 1] = -1.5, 10.0              # ABS() branch
    S[0, 2], T[0, 2] = 42.0, 40.0
    S[0, 3], T[0, 3] = 0.5, -2.0
    D = (5500.0 * rng.random(shape)).astype(np.float32)
    D[0, 0] = 3000.0
    out = {"T": T, "S": S, "D": D}
    for teos, tag in ((False, "eos80"), (True, "teos10")):
        out[f"sigma0_{tag}"] = ref.sigma0(T, S, teos)
        out[f"sigmai2000_{tag}"] = ref.sigmai(T, S, 2000.0, teos)
        out[f"sigmai3000_{tag}"] = ref.sigmai(T, S, 3000.0, teos)
        out[f"sigmai2d_{tag}"] = ref.sigmai2d(T, S, D, teos)
    out["sigmantr"] = ref.sigmantr(T, S)
    out["sigmai2000_jmd94"] = ref.sigmai_jmd94(T, S, 2000.0)
    out["sigmai2d_jmd94"] = ref.sigmai2d_jmd94(T, S, D)
    path = os.path.join(HERE, "golden", "eos.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
