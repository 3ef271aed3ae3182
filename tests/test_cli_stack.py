"""No slab-sized array may live on the stack: a NEMO slab is 7 MB at eORCA025 and 62 MB at eORCA12, the default stack
8 MB.  The programs are run with a 1 MB stack limit on a grid whose r4 slab is 1.5 MB; one slab-sized automatic
array or stack temporary ends them with SIGSEGV (the first cdfmoc of this build did exactly that at eORCA025 size,
found by tools/e2e_cdfmoc.py).  tools/e2e_all.py repeats this at full eORCA025 size outside the test suite."""
import os
import resource
import subprocess

import pytest

import nemo_files
import run_program
from test_cli import BIN, tools  # noqa: F401

T, U, V = "SYN_y0001_gridT.nc", "SYN_y0001_gridU.nc", "SYN_y0001_gridV.nc"
RUNS = [
    ("cdfmoc", ["-v", V]),
    ("cdfmoc", ["-v", V, "-t", T, "-decomp", "-o", "dec.nc"]),
    ("cdfheatc", ["-f", T]),
    ("cdfsig0", ["-t", T]),
    ("cdfsigi", ["-t", T, "-r", "2000"]),
    ("cdfcurl", ["-u", U, "vozocrtx", "-v", V, "vomecrty", "-l", "1"]),
    ("cdfpsi", ["-u", U, "-v", V]),
    ("cdfvT", ["-c", "SYN", "-l", "y0001"]),
    ("cdfmhst", ["-vt", "vt.nc", "-MST"]),
    ("cdfvhst", ["-f", "vt.nc"]),
    ("cdfvtrp", ["-u", U, "-v", V]),
    ("cdfmean", ["-f", T, "-v", "votemper", "-p", "T", "-var", "-S", "-zeromean"]),
    ("cdfsum", ["-f", T, "-v", "votemper", "-p", "T"]),
    ("cdfzonalsum", ["-f", T, "-p", "T", "-b", "new_maskglo.nc"]),
    ("cdfzonalmean", ["-f", T, "-p", "T", "-b", "new_maskglo.nc", "-max"]),
    ("cdfvint", ["-f", T]),
    ("cdfvertmean", ["-f", T, "-l", "votemper,vosaline", "-p", "T", "-zlim", "15", "120"]),
    ("cdfmocsig", ["-v", V, "-t", T, "-r", "2000", "-isodep"]),
]


def small_stack():
    resource.setrlimit(resource.RLIMIT_STACK, (1024 * 1024, 1024 * 1024))


@pytest.mark.gpu
def test_programs_run_with_a_stack_smaller_than_one_slab(tools, tmp_path):  # noqa: F811
    d = str(tmp_path)
    nemo_files.make_case(d, npi=768, npj=512, npk=4, npt=1)               # one r4 slab = 1.5 MB
    bad = []
    for tool, args in RUNS:
        p = run_program.run_watched([os.path.join(BIN, tool)] + args, cwd=d, env=None, timeout=600, preexec_fn=small_stack)
        if p.returncode != 0:
            bad.append((tool, p.returncode, (p.stdout + p.stderr)[-300:]))
    assert not bad, bad
