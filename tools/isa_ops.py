#!/usr/bin/env python3
"""r8 operations per point of the fp64-bound kernels, counted in the gfx950 ISA the compiler emits (not estimated from the
source): eos_kernel (sigma0 / sigmai, exact and FMA forms), eos_bn2_kernel, mocsig_class_kernel.

Each kernel is compiled to device assembly with the flags of cdftools_amd/csrc/Makefile, its main loop is found (the
backward branch whose body holds the most f64 instructions) and the f64 instructions of that body are counted by kind;
divided by the points one iteration handles, that is what tools/bench_tools.py prices against the vector fp64 rate:
    v_mul_f64 / v_add_f64 : one operation each;  v_fma_f64 : one instruction, two flops;
    everything else f64 (conversions, v_rcp / v_rsq / v_div_* steps of a division or square root, compares, ldexp): "other",
    each one issue slot of the same half-rate pipeline.
Writes profiles/<round>/isa_fp64_ops.json.      python3 tools/isa_ops.py [--out profiles/r03/isa_fp64_ops.json]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cdftools_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S"]
# kernel name pattern (demangled) -> points handled by one iteration of its main loop
WANT = [
    ("eos.hip", r"eos_kernel<0, false, 2, 4, 1>", 4, "sigma0 EOS-80, exact, tool form (the surface: r0 alone)"),
    ("eos.hip", r"eos_kernel<0, false, 2, 4, 0>", 4, "sigmai EOS-80, exact, tool form (a reference depth: four polynomials)"),
    ("eos.hip", r"eos_kernel<0, true, 2, 4, 0>", 4, "the same with FMA contraction (CDFT_EOS_FAST_FMA)"),
    ("eos.hip", r"eos_kernel<1, false, 2, 4, 0>", 4, "TEOS-10, exact, tool form"),
    ("eos_more.hip", r"eos_bn2_kernel<0, 0, 4>", 4, "eosbn2, polynomial alpha and beta"),
    ("mocsig.hip", r"mocsig_class_kernel<5, 0, 256, 2, 0>", 20, "cdfmocsig class kernel, EOS-80, 5 vectors per lane (eORCA12 rows)"),
    ("mocsig.hip", r"mocsig_class_kernel<5, 0, 256, 2, 3>", 20, "the same when the three missing values are 0 (NEMO files)"),
    ("mocsig.hip", r"mocsig_class_kernel<2, 0, 256, 2, 0>", 8, "cdfmocsig class kernel, EOS-80, 2 vectors per lane (eORCA025 rows)"),
]
F64 = re.compile(r"^\s+(v_[a-z0-9_]*f64[a-z0-9_]*)\b")


def device_asm(src):
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["hipcc", *FLAGS, os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.remove(out)
    return txt


def functions(asm):
    """-> {mangled name: [lines]} of every function body"""
    fn, cur, name = {}, None, None
    for ln in asm.split("\n"):
        m = re.match(r"^(_Z[\w$.]+):", ln)
        if m:
            name, cur = m.group(1), []
            fn[name] = cur
        elif ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            cur.append(ln)
    return fn


def main_loop(lines):
    """the backward-branch body with the most f64 instructions -> (counts by kind, body length in instructions)"""
    label_at = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = i
    best = None
    for i, ln in enumerate(lines):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            body = lines[label_at[m.group(1)]:i]
            kinds = {}
            for b in body:
                k = F64.match(b)
                if k:
                    kinds[k.group(1)] = kinds.get(k.group(1), 0) + 1
            n = sum(kinds.values())
            if best is None or n > best[0]:
                other = {}
                for b in body:                                    # the vector instructions that are not r8 arithmetic, by mnemonic
                    k = re.match(r"^\s+(v_[a-z0-9_]+)", b)
                    if k and "f64" not in k.group(1):
                        other[k.group(1)] = other.get(k.group(1), 0) + 1
                best = (n, kinds, sum(1 for b in body if re.match(r"^\s+[vs]_", b)), other)
    return best


def main():
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "profiles", "r04", "isa_fp64_ops.json")
    asm, res = {}, {}
    for src, pat, points, what in WANT:
        if src not in asm:
            asm[src] = functions(device_asm(src))
        names = list(asm[src])
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        hit = [n for n, d in zip(names, dem) if pat in d]
        if not hit:
            print("not found:", pat, file=sys.stderr)
            continue
        n, kinds, ninstr, nonr8 = main_loop(asm[src][hit[0]])
        mul = sum(v for k, v in kinds.items() if k.startswith("v_mul_f64"))
        add = sum(v for k, v in kinds.items() if k.startswith("v_add_f64"))
        fma = sum(v for k, v in kinds.items() if k.startswith("v_fma_f64"))
        other = n - mul - add - fma
        if points is None:                                # one point per lane and iteration
            points = 1
        res[pat] = {"what": what, "source": src, "points_per_iteration": points, "loop_instructions": ninstr,
                    "f64_instructions_per_iteration": kinds, "vector_instructions_not_f64_per_iteration": nonr8,
                    "per_point": {"mul": mul / points, "add": add / points, "fma": fma / points, "other_f64": other / points,
                                  "issue_slots": n / points, "flops": (mul + add + 2 * fma) / points,
                                  "vector_not_f64": sum(nonr8.values()) / points, "vector_all": (n + sum(nonr8.values())) / points}}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k, json.dumps(v["per_point"]))


if __name__ == "__main__":
    main()
