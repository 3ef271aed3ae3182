#!/usr/bin/env python3
"""Generates tests/golden/eos_more.npz from the REAL reference (oracle/_ref/libeos_ref.so = the unmodified
/root/reference/src/eos.f90 compiled where it lies by oracle/ref_build.sh): spice (both members), alpha, beta, albet,
beta of McDougall 1987 (2-D and profile members, alpha / beta also at one r8 point) and eosbn2 (both members, both
plane orders).  Build container only; the .npz holds inputs and the reference's outputs, nothing of its source.

    python3 tests/make_golden_eos_more.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_lib  # noqa: E402


def inputs():
    rng = np.random.default_rng(20240602)
    shape = (19, 23)
    T = (-2.0 + 32.0 * rng.random(shape)).astype(np.float32)
    S = (30.0 + 8.0 * rng.random(shape)).astype(np.float32)
    S[rng.random(shape) < 0.2] = 0.0                       # land: alpha and beta mask on S == 0
    S[0, 0], T[0, 0] = 33.0, 15.0                          # the check point of spice (src/eos.f90:700)
    S[0, 1], T[0, 1] = -1.5, 10.0                          # ABS() branch of the polynomial forms
    S[0, 2], T[0, 2] = 35.0, 0.0                           # s - 35 == 0
    S[0, 3], T[0, 3] = 42.0, 40.0
    D = (5500.0 * rng.random(shape)).astype(np.float32)
    T2 = (T + rng.normal(0, 1, shape)).astype(np.float32)  # the level below
    S2 = (S + rng.normal(0, 0.2, shape)).astype(np.float32)
    S2[S == 0] = 0.0
    S[1, 0] = 34.25
    S2[1, 0] = 70.0 - S[1, 0]                              # mean salinity exactly 35: the anomaly handed to alpha_0d is 0
    E = (1.0 + 200.0 * rng.random(shape)).astype(np.float32)
    return T, S, D, T2, S2, E


POINTS = [(3.0, 35.5, 3000.0), (15.0, 33.0, 0.0), (-1.5, 0.0, 120.0), (28.0, -2.0, 5000.0)]


def main():
    ref = oracle_lib.RefEos()
    T, S, D, T2, S2, E = inputs()
    out = {"T": T, "S": S, "D": D, "T2": T2, "S2": S2, "E": E}
    out["spice_pol"], out["spice_f02"] = ref.spice(T, S, False), ref.spice(T, S, True)
    for teos, tag in ((False, "eos80"), (True, "teos10")):
        for which, nm in ((0, "alpha"), (1, "beta"), (2, "albet"), (3, "beta_mcd87")):
            for dep in (0.0, 1234.5):
                out[f"{nm}_{dep:g}_{tag}"] = ref.coef(which, T, S, dep, teos)
            out[f"{nm}_z_{tag}"] = ref.coef(which, T, S, D, teos)
        out[f"points_{tag}"] = np.array([[ref.coef0d(w, *p, teos10=teos) for p in POINTS] for w in (0, 1)])
        for mcd, nm in ((False, "pol"), (True, "mcd87")):
            for up in (1, 2):
                out[f"bn2_{nm}_up{up}_{tag}"] = ref.bn2(T, T2, S, S2, E, 777.0, teos, mcd, up)
    path = os.path.join(HERE, "golden", "eos_more.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
