"""Program-level differential: this build's programs against what the reference's OWN PROGRAMS left behind on the same files
(tests/golden/prog_*.npz, written in the build container by tests/make_golden_prog.py from the unmodified reference sources
linked over this repository's NetCDF layer -- oracle/ref_prog_build.sh).

What it is worth: both sides went through the same NetCDF backend, so this pins nothing about that backend (scipy does:
tests/test_ncclassic.py); it is a DIAGNOSTIC of everything above it -- option parsers, CreateOutput (dimension and variable
order, every attribute and its type and order), cdfio's getvar / putvar conventions, the text on stdout, the .dat dumps --
and of the arithmetic end to end through the real programs.  Every difference found is either fixed or listed in EXPLAINED
with its reason; anything else fails.

Tolerances on data: cdfcurl, cdfsig0, cdfsigi, cdfvT, cdfpsi: the same bits.  cdfheatc (whose default sums are
the reference's sequential r4 sums): the same bits.  cdfmoc: written floats within 1 ulp(r4); cdfmhst: within 2 ulp(r4)
(r8 sums regrouped along i, then the r4 rounding of the store)."""
import os
import re
import subprocess

import numpy as np
import pytest

import nemo_files
import prog_cases as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cdftools_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")

ULPS = {"cdfmoc": 1, "cdfmhst": 2}                      # float data of these programs: within n ulp(r4) of the reference program's
# cdfmoc -decomp: the shear, barotropic and ageostrophic parts are differences of zonal r8 sums that this build regroups
# (1e-13 of the sum of |terms|, tests/test_moc_decomp.py): a value that cancels to little is held to the r4 resolution of the
# FIELD's scale (the largest |value| of the file's variables), not of itself
DECOMP_SCALE_EPS = 2.0 ** -22

# Differences that are known and explained: (case regex, kind, item regex) -> reason.
#   kind "att": item = "<file>:<variable or GLOBAL>:<attribute>"; "stdout": item = the reference's line; "data": "<file>::<var>"
EXPLAINED = [
    (r"cdfheatc_.*", "att", r".*:(heatc3d|heatc2d|heatc3dpervol):(_FillValue|valid_min|valid_max)$",
     "the reference never sets rmissing_value / valid_min / valid_max of these descriptors (src/cdfheatc.f90:271-292): it writes "
     "whatever the uninitialised TYPE(variable) holds; this build writes the defaults of its descriptor"),
    (r"cdf(psi|vtrp)_.*", "data", r".*::depthu$",
     "putheadervar's search for a depth variable (src/cdfio.F90 putheadervar) leaves its loop at the FIRST name whether the file "
     "holds it or not (idep starts as NF90_NOERR and only getvar1d changes it): a U file has no deptht, so the reference writes "
     "the never-assigned z1d (0.0 in this build of it); this build writes the file's depthu"),
    (r"cdfpsi_ssh", "data", r".*::sobarstf(ssh|total)$",
     "without -V or -mean the reference's SSH term multiplies zv (src/cdfpsi.f90:299), an array it ALLOCATEs and never assigns on "
     "that path: whatever the heap holds (this build of the reference: not zeros); this build and the lifted harness take it as 0. "
     "With -mean both velocities are read and the term is defined: the case cdfpsi_ssh_mean compares it"),
]


def explained(case, kind, item):
    return any(re.fullmatch(c, case) and k == kind and re.fullmatch(i, item) for c, k, i, _ in EXPLAINED)


def ulp4(x):
    return np.spacing(np.abs(x.astype(np.float32))).astype(np.float64)


def numbers(text):
    return [float(t) for t in re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?", text)]


def diff_case(case, ref, ref_arr, got, got_arr):
    """-> list of (kind, item, detail) for everything that differs"""
    name, prog = case[0], case[1]
    out = []
    if ref["rc"] != got["rc"]:
        out.append(("rc", "rc", f"reference {ref['rc']}, here {got['rc']}"))
    rl, gl = ref["stdout"].splitlines(), got["stdout"].splitlines()
    for i in range(max(len(rl), len(gl))):
        a = rl[i] if i < len(rl) else "<nothing>"
        b = gl[i] if i < len(gl) else "<nothing>"
        if a != b:
            out.append(("stdout", a, f"line {i + 1}: reference {a!r} / here {b!r}"))
    for fname, rs in ref["files"].items():
        gs = got["files"].get(fname)
        if gs is None:
            out.append(("file", fname, "not written"))
            continue
        if isinstance(rs, str):                                      # a text dump: the same layout, numbers within 2 ulp(r4)
            ra, ga = numbers(rs), numbers(gs)
            skel = lambda t: re.sub(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?", "#", t)      # noqa: E731
            if skel(rs) != skel(gs) or len(ra) != len(ga):
                out.append(("text", fname, "layout differs"))
            elif ra:
                ra, ga = np.array(ra), np.array(ga)
                if not np.all(np.abs(ra - ga) <= 2 * ulp4(ra) + 1e-5 * np.abs(ra)):   # (printed with ~6 digits)
                    out.append(("text", fname, f"numbers differ by up to {np.abs(ra - ga).max():.3e}"))
            continue
        for key in ("version_byte", "dimensions"):
            if rs[key] != gs[key]:
                out.append(("schema", f"{fname}:{key}", f"reference {rs[key]} / here {gs[key]}"))

        def atts(where, ra, ga):
            rn, gn = [k for k, _ in ra], [k for k, _ in ga]
            if rn != gn:
                out.append(("att", f"{fname}:{where}:<names>", f"reference {rn} / here {gn}"))
            gd = dict((k, v) for k, v in ga)
            for k, v in ra:
                if k in gd and gd[k] != v:
                    out.append(("att", f"{fname}:{where}:{k}", f"reference {v} / here {gd[k]}"))
        atts("GLOBAL", rs["global"], gs["global"])
        rv, gv = [v["name"] for v in rs["variables"]], [v["name"] for v in gs["variables"]]
        if rv != gv:
            out.append(("schema", f"{fname}:<variables>", f"reference {rv} / here {gv}"))
        gby = {v["name"]: v for v in gs["variables"]}
        for v in rs["variables"]:
            g = gby.get(v["name"])
            if g is None:
                continue
            for key in ("type", "dims", "shape"):
                if v[key] != g[key]:
                    out.append(("schema", f"{fname}:{v['name']}:{key}", f"reference {v[key]} / here {g[key]}"))
            atts(v["name"], v["atts"], g["atts"])
            k = f"{fname}::{v['name']}"
            a, b = ref_arr[k], got_arr.get(k)
            if b is None or a.shape != b.shape:
                continue
            if np.array_equal(a, b, equal_nan=True):
                continue
            n = ULPS.get(prog, 0) if a.dtype == np.float32 else 0
            if n and np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= n * ulp4(a)):
                continue
            if prog == "cdfvtrp" and v["name"] in ("soastrp", "socstrp"):
                # -bathy rotates the transports by the bottom slope's angle through the runtime's r4 ATAN2 / COS / SIN, which are not
                # correctly rounded: the tolerance of tests/test_vtrp_bathy.py, 3 * 2^-23 * (|utrp| + |vtrp|) (+ the stored value's ulp)
                du = np.abs(ref_arr[f"{fname}::sozoutrp"].astype(np.float64))        # (t,y,x): the transports at U and V points ...
                dv = np.abs(ref_arr[f"{fname}::somevtrp"].astype(np.float64))
                wu, wv = np.zeros_like(du), np.zeros_like(dv)                        # ... brought to the T point as :215-221 does
                wu[..., 1:] = 0.5 * (du[..., 1:] + du[..., :-1]); wu[..., 0] = wu[..., -2]
                wv[..., 1:, :] = 0.5 * (dv[..., 1:, :] + dv[..., :-1, :])
                lim = 3 * 2.0 ** -23 * (wu + wv)
                if np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= lim + ulp4(a)):
                    continue
            if prog == "cdfmoc" and "-decomp" in case[2] and a.dtype == np.float32:
                scale = max(float(np.abs(ref_arr[f"{fname}::{w['name']}"]).max()) for w in rs["variables"] if w["name"].startswith("zomsf"))
                if np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= n * ulp4(a) + DECOMP_SCALE_EPS * scale):
                    continue
            with np.errstate(invalid="ignore"):
                worst = np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(ulp4(a), 1e-300)) if a.dtype.kind == "f" else -1
            out.append(("data", k, f"{int((a != b).sum())} of {a.size} values differ, worst {worst:.3g} ulp(r4)"))
    return out


@pytest.mark.gpu
def test_programs_against_the_reference_programs(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cdftools_amd", "fortran")], stdout=subprocess.DEVNULL)
    d = str(tmp_path)
    g_, rec_ = nemo_files.make_case(d)
    pc.extra_files(d, g_, rec_)
    report, unexplained = [], []
    out_of_it = pc.left_out()
    for case in pc.CASES:
        if case[0] in out_of_it:                                       # the reference program does not run as it is: no golden
            pc.run_case(BIN, case, d)                                  # (still run here: a later case may read its output)
            continue
        ref, ref_arr = pc.load_golden(os.path.join(GOLD, f"prog_{case[0]}.npz"))
        got, got_arr = pc.run_case(BIN, case, d)
        for kind, item, detail in diff_case(case, ref, ref_arr, got, got_arr):
            ok = explained(case[0], kind, item)
            report.append(f"{case[0]:24s} {'explained' if ok else 'UNEXPLAINED':11s} {kind:7s} {item}: {detail}")
            if not ok:
                unexplained.append(report[-1])
    out = os.path.join(ROOT, "gpurun_out", "r04")
    try:
        os.makedirs(out, exist_ok=True)
        open(os.path.join(out, "prog_diff_report.txt"), "w").write("\n".join(report) + "\n")
    except OSError:
        pass
    assert not unexplained, "\n".join(unexplained[:60])


@pytest.fixture(scope="module")
def programs_built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cdftools_amd", "netcdf")], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "cdftools_amd", "lib", "libcdftools_hip.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cdftools_amd", "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cdftools_amd", "fortran")], stdout=subprocess.DEVNULL)


@pytest.mark.parametrize("case", pc.USAGE_CASES, ids=[c[0] for c in pc.USAGE_CASES])
def test_usage_texts_are_the_reference_programs(programs_built, case):
    """A program called without an argument prints the reference program's usage text (tests/golden/prog_<program>_usage.npz:
    the stdout of the reference's own binary), then this build's "MI355X BUILD" lines; exit status 0.  This build writes every
    line whole (PRINT '(99a)'); the reference's PRINT * under flang cuts records at column 79, so the golden is compared with
    this build's text cut the same way.  CPU: no device is touched before the usage."""
    ref, _ = pc.load_golden(os.path.join(GOLD, f"prog_{case[0]}.npz"))
    p = subprocess.run([case[1]], executable=os.path.join(BIN, case[1]), capture_output=True, text=True, timeout=120)
    assert p.returncode == ref["rc"] == 0
    ours = p.stdout.rstrip("\n").split("\n")
    k = next(i for i, ln in enumerate(ours) if "MI355X BUILD :" in ln)
    assert k >= 2 and len(ours) > k + 1 and not ours[k - 1].strip()                 # a blank line, the heading, at least one line
    assert all(ln.startswith("  ") for ln in ours if ln.strip())                   # nothing folded in this build's output
    assert pc.fold_like_list_directed_output("\n".join(ours[:k - 1])) == ref["stdout"].rstrip("\n")


def test_goldens_are_there_for_every_case():
    out_of_it = pc.left_out()
    assert set(out_of_it) <= {c[0] for c in pc.CASES} and len(out_of_it) <= 4
    for case in pc.CASES:
        if case[0] in out_of_it:
            continue
        meta, arrays = pc.load_golden(os.path.join(GOLD, f"prog_{case[0]}.npz"))
        assert (meta["rc"] == 0 or not case[3]) and meta["program"] == case[1] and meta["args"] == case[2]
        assert all(v is not None for v in meta["files"].values()) and (arrays or not case[3])
