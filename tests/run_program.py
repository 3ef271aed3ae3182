"""One program run under a time limit, for every test that starts one of cdftools_amd/bin.  A run that does not end is never tried
again and never passes: before it is killed, the state, wait channel and system call of each of its threads are read from /proc,
a backtrace of every thread is taken when rocgdb may attach, and the whole goes to gpurun_out/hung_programs.txt (what travels
back from the GPU box), to stderr and into the test's outcome -- XFAILED with the report as its reason (`pytest -rx` prints it; the
summary line counts it), or FAILED when CDFT_TEST_HANG=fail (the builder's stress runs).  Round 5 saw one run in several
thousand not end and could not make it happen again (some 25 000 program runs since, round 6's 9 000 included, without one:
profiles/r06/README.md, section 5); its tests tried such a run once more, which would have let a deadlock in the ring, the events or
the I/O threads through unseen -- the round-5 advisor's finding, who asked for a failure "or at least a loud xfail" with the
threads' state captured.  The xfail is the default because nothing is known of where the one occurrence so far stood (it may as
well have been the HIP runtime's start or end of a process as this repository's code): a suite run must not go red on it, and
must not hide it either."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def describe(pid):
    """what every thread of a live process is doing"""
    out = []
    for t in sorted(glob.glob(f"/proc/{pid}/task/*")):
        def rd(name):
            try:
                return open(os.path.join(t, name)).read().strip()
            except OSError as e:
                return f"<{e.strerror}>"
        st = [ln for ln in rd("status").splitlines() if ln.startswith(("Name", "State"))]
        out.append(f"thread {os.path.basename(t)}: {' '.join(st)} wchan={rd('wchan')} syscall={rd('syscall')[:60]}")
    try:
        r = subprocess.run(["rocgdb", "-p", str(pid), "-batch", "-ex", "thread apply all bt 16"], capture_output=True, text=True, timeout=120)
        out.append("--- rocgdb\n" + (r.stdout + r.stderr)[-8000:])
    except (OSError, subprocess.TimeoutExpired) as e:
        out.append(f"(rocgdb: {e})")
    return "\n".join(out)


def run_watched(argv, cwd, env, timeout, executable=None, preexec_fn=None):
    """subprocess.run(..., capture_output=True, text=True) that fails the test, with a description of the process, on a hang"""
    p = subprocess.Popen(argv, executable=executable, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         preexec_fn=preexec_fn)
    try:
        so, se = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        what = describe(p.pid)
        p.kill()
        so, se = p.communicate()
        keys = ("CDFT_", "HIP_", "HSA_", "OMP_")
        report = "HUNG after %d s: %s  (cwd %s)\nenv %s\n%s\n--- stdout so far\n%s\n--- stderr so far\n%s\n" % (
            timeout, " ".join(argv), cwd, {k: v for k, v in (env or {}).items() if k.startswith(keys)}, what, (so or "")[-1500:], (se or "")[-1500:])
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "hung_programs.txt"), "a") as f:
                f.write(report)
        except OSError:
            pass
        sys.stderr.write(report)
        if os.environ.get("CDFT_TEST_HANG", "xfail") == "fail":
            raise AssertionError(report)
        import pytest
        pytest.xfail("a program run did not end (described in gpurun_out/hung_programs.txt):\n" + report)
    return subprocess.CompletedProcess(argv, p.returncode, so, se)
