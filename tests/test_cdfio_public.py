"""cdfio's public surface (reference src/cdfio.F90:173-180) -- what the tools of this build call and, since round 5, the rest
of that list too (copyatt, cvaratt, gettimeatt, puttimeatt, atted, getdimvar, getvarid, getvarxz / yz, getvar3dt, getvar4d,
gettimeseries, ncopen: no program here calls them; one call each below) -- and the key_CMIP6 name set of modcdfnames.

tests/fortran/cdfio_public.f90 calls each routine with the reference's argument order and keywords on files written here by
scipy.io.netcdf_file (an independent implementation of the format); what it prints is compared with NumPy slices of the
same arrays, what it writes is read back with scipy.  Covered on the way: variables of every rank readvar3d can meet (a
time-less (x,y,z) mask or metric, (x,y,t), (x,y)), packed variables, windows over the eastern edge, the table of read
handles against open / close per call (CDFT_IO_NOCACHE=1).  CPU only: the driver links the module objects of
cdftools_amd/fortran/build (made by `build()`), no kernel is called."""
import os
import shutil
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "cdftools_amd", "fortran")
BUILD = os.path.join(FDIR, "build")
LIBDIR = os.path.join(ROOT, "cdftools_amd", "lib")
MODS = ["netcdf", "modcdfnames", "cdftools_hip", "cdfio", "cdffeed", "eos", "modutils", "cdftools", "cdfmoc_rapid"]
FFLAGS = ["-O1", "-fPIC", "-cpp", "-ffp-contract=off"]
f32 = np.float32
NPI, NPJ, NPK, NPT = 6, 5, 4, 3

pytestmark = pytest.mark.skipif(shutil.which("amdflang") is None or not os.path.exists(os.path.join(BUILD, "cdfio.o")),
                                reason="needs amdflang and the built Fortran modules (python -c 'import __graft_entry__ as g; g.build()')")


def link_args():
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    return [os.path.join(BUILD, m + ".o") for m in MODS] + [
        f"-L{LIBDIR}", "-lcdftools_hip", "-lncclassic", f"-Wl,-rpath,{LIBDIR}", f"-L{rocm}/lib", "-lamdhip64", f"-Wl,-rpath,{rocm}/lib", "-lstdc++"]


def fields():
    rng = np.random.default_rng(20)
    t = rng.normal(10, 5, (NPT, NPK, NPJ, NPI)).astype(f32)
    t[:, :, 0, 0] = 0
    ssh = rng.normal(0, 1, (NPT, NPJ, NPI)).astype(f32)
    pk = rng.integers(-3000, 3000, (NPT, NPK, NPJ, NPI)).astype(np.int16)
    pk[:, 1, 2, 3] = -32768                                  # the packed variable's missing value
    m4 = rng.integers(0, 2, (NPT, NPK, NPJ, NPI)).astype(np.int8)
    e3 = rng.uniform(1, 200, (NPK, NPJ, NPI)).astype(f32)
    m3 = rng.integers(0, 2, (NPK, NPJ, NPI)).astype(np.int8)
    area = rng.uniform(1e6, 1e9, (NPJ, NPI)).astype(f32)
    return dict(t=t, ssh=ssh, pk=pk, m4=m4, e3=e3, m3=m3, area=area)


def write_files(d, F):
    f = netcdf_file(os.path.join(d, "T.nc"), "w", version=2)
    f.createDimension("time_counter", None)
    f.createDimension("deptht", NPK)
    f.createDimension("y", NPJ)
    f.createDimension("x", NPI)
    lon = f.createVariable("nav_lon", "f4", ("y", "x")); lon[:] = np.arange(NPJ * NPI, dtype=f32).reshape(NPJ, NPI)
    lat = f.createVariable("nav_lat", "f4", ("y", "x")); lat[:] = -np.arange(NPJ * NPI, dtype=f32).reshape(NPJ, NPI)
    dd = f.createVariable("deptht", "f4", ("deptht",)); dd[:] = np.array([5, 15, 27.5, 45], f32)
    tc = f.createVariable("time_counter", "f8", ("time_counter",)); tc[:] = np.array([43200.0, 129600.0, 216000.0])
    tc.calendar, tc.title, tc.long_name, tc.units, tc.time_origin = "noleap", "Time", "Time axis", "seconds since 0001-01-01", "0001-JAN-01"
    v = f.createVariable("votemper", "f4", ("time_counter", "deptht", "y", "x")); v[:] = F["t"]
    v.units, v.valid_max, v._FillValue, v.iweight, v.long_name = "C", f32(40.0), f32(0.0), np.int32(12), "Temperature"
    s = f.createVariable("sossheig", "f4", ("time_counter", "y", "x")); s[:] = F["ssh"]
    s.missing_value = f32(1e20)
    p = f.createVariable("packed", "i2", ("time_counter", "deptht", "y", "x")); p[:] = F["pk"]
    p.scale_factor, p.add_offset, p.missing_value = f32(0.01), f32(20.0), np.int16(-32768)
    m = f.createVariable("tmask4d", "i1", ("time_counter", "deptht", "y", "x")); m[:] = F["m4"]
    f.close()
    f = netcdf_file(os.path.join(d, "mesh.nc"), "w", version=2)              # no record dimension at all
    f.createDimension("z", NPK)
    f.createDimension("y", NPJ)
    f.createDimension("x", NPI)
    e = f.createVariable("e3static", "f4", ("z", "y", "x")); e[:] = F["e3"]
    m = f.createVariable("tmaskstatic", "i1", ("z", "y", "x")); m[:] = F["m3"]
    a = f.createVariable("area", "f4", ("y", "x")); a[:] = F["area"]
    f.close()


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("cdfio_public"))
    exe = os.path.join(d, "cdfio_public")
    r = subprocess.run(["amdflang", *FFLAGS, f"-I{BUILD}", "-module-dir", d, os.path.join(ROOT, "tests", "fortran", "cdfio_public.f90"),
                        "-o", exe, *link_args()], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    F = fields()
    write_files(d, F)
    r = subprocess.run([exe, d], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0 and "done" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    out = {}
    for ln in r.stdout.splitlines():
        k, _, rest = ln.strip().partition(" ")
        out.setdefault(k, rest.strip())
    return d, out, F, r.stdout


def test_switches_change_nothing_on_stdout(run):
    """CDFT_TIMING=1: the read / write wall-time totals on stderr at closeout; CDFT_IO_NOCACHE=1: a file opened and closed at
    every call as in the reference instead of a kept handle -- the same answers either way"""
    d, _, _, stdout = run
    exe = os.path.join(d, "cdfio_public")
    r1 = subprocess.run([exe, d], capture_output=True, text=True, cwd=d, env=dict(os.environ, CDFT_TIMING="1"))
    assert r1.returncode == 0 and r1.stdout == stdout
    assert "CDFT_TIMING cdfio (s): reads" in r1.stderr and "writes" in r1.stderr
    r2 = subprocess.run([exe, d], capture_output=True, text=True, cwd=d, env=dict(os.environ, CDFT_IO_NOCACHE="1"))
    assert r2.returncode == 0 and r2.stdout == stdout and "CDFT_TIMING" not in r2.stderr


def nums(s, shape=None):
    a = np.array(s.split(), dtype=np.float64).astype(f32)
    return a if shape is None else a.reshape(shape)          # Fortran order printed = reversed C shape


def unpack(pk):
    x = pk.astype(f32)
    return np.where(pk == -32768, x, x * f32(0.01) + f32(20.0)).astype(f32)


def test_inquiries(run):
    d, out, F, _ = run
    assert out["getdim"].split() == [str(NPI), str(NPJ), str(NPK), str(NPT)] and out["getdim_true"] == "deptht"
    assert out["getdim_absent"].split() == ["0", "1", "unknown"]
    assert out["getvdim"].split() == ["3", "1"]
    v, name = out["getspval"].split()
    assert f32(v) == f32(1e20) and name == "missing_value"
    a = out["getatt"].split()
    assert f32(a[0]) == f32(40.0) and a[1] == "12" and a[2] == "C"
    v, rest = out["getvaratt"].split(None, 1)
    assert f32(v) == f32(0.0) and rest == "|C|Temperature|keep|"          # an absent attribute leaves its argument alone
    assert out["getnvar"] == "8"
    assert out["getvarname"].split() == ["nav_lon", "nav_lat", "deptht", "time_counter", "votemper", "sossheig", "packed", "tmask4d"]
    assert out["getipk"].split() == ["0", "0", "0", "0", str(NPK), "1", str(NPK), str(NPK)]
    v, rest = out["stypin_votemper"].split(None, 1)
    assert f32(v) == f32(40.0) and rest == "|C|"


def test_slab_and_block_reads(run):
    d, out, F, _ = run
    t, ssh, pk = F["t"], F["ssh"], F["pk"]
    assert np.array_equal(nums(out["getvar_packed"], (NPJ, NPI)), unpack(pk)[1, 1])
    assert np.array_equal(nums(out["getvar_wrap"], (NPJ, 4)), t[0, 2][:, [3, 4, 1, 2]])
    assert np.array_equal(nums(out["getvar_edge"], (NPJ, 4)), t[0, 2][:, 2:6])
    assert np.array_equal(nums(out["getvar3d"], (NPK - 1, NPJ - 2, NPI - 1)), t[1, 1:, 2:, 1:])
    assert np.array_equal(nums(out["readvar3d_packed"], (NPK, NPJ, NPI)), unpack(pk)[2])
    assert np.array_equal(nums(out["readvar3d_xyt"], (NPK, NPJ, NPI)), np.broadcast_to(ssh[1], (NPK, NPJ, NPI)))
    # the advisor's round-2 finding: a (x,y,z) variable without a time axis gave its first level only
    assert np.array_equal(nums(out["readvar3d_xyz"], (NPK, NPJ, NPI)), F["e3"])
    assert np.array_equal(nums(out["readvar3d_xyz_window"], (NPK - 1, NPJ - 2, NPI - 1)), F["e3"][1:, 2:, 1:])
    assert np.array_equal(nums(out["readvar3d_xy"], (NPK, NPJ, NPI)), np.broadcast_to(F["area"], (NPK, NPJ, NPI)))
    assert np.array_equal(nums(out["readvar3d_wrap"], (NPK, NPJ, 4)), t[1][:, :, [3, 4, 1, 2]])
    assert np.array_equal(nums(out["readvar3d_i1_xyz"], (NPK, NPJ, NPI)), F["m3"].astype(f32))
    assert np.array_equal(nums(out["readvar3d_i1_xyzt"], (NPK, NPJ, NPI)), F["m4"][1].astype(f32))
    assert out["getvar1d"].split() == ["4.3200000000000000E+04", "1.2960000000000000E+05"]
    v, flag = out["getvar1d_absent"].split()
    assert float(v) == 99999999999.0 and flag == "1"


def test_the_rest_of_the_public_list(run):
    """the thirteen names of src/cdfio.F90:173-180 no program of this build calls, against NumPy slices of the same arrays and
    scipy's reading of the edited file"""
    d, out, F, stdout = run
    t, ssh, pk = F["t"], F["ssh"], F["pk"]
    assert out["getvarid"].split() == [str(i) for i in range(1, 9)]
    # (no 'axis' in the file: nf90_get_att blanks a text result before it looks the attribute up, so the reference's 'unknown'
    # does not survive the call that fails -- the NF90 layer of this build does the same)
    assert out["gettimeatt"] == "|noleap|Time|Time axis||seconds since 0001-01-01|0001-JAN-01|"
    assert np.array_equal(nums(out["getdimvar"]), np.array([5, 15, 27.5, 45], f32)) and out["getdimvar_name"] == "deptht"
    up = unpack(pk)
    assert np.array_equal(nums(out["getvarxz"], (NPK - 1, NPI - 1)), up[1, 1:, 2, 1:])
    assert np.array_equal(nums(out["getvaryz"], (NPK, NPJ)), t[2, :, :, 3])
    assert np.array_equal(nums(out["getvar3dt"], (2, NPJ, NPI)), t[1:3, 1])
    assert np.array_equal(nums(out["getvar3dt_xyt"], (2, NPJ, NPI)), ssh[0:2])
    assert np.array_equal(nums(out["getvar4d"], (2, 2, 2, 3)), up[1:3, 1:3, 2:4, 1:4])
    lines = stdout.splitlines()
    i0, i1 = lines.index("gettimeseries_begin"), lines.index("gettimeseries_end")
    rows = np.array([[float(x) for x in ln.split()] for ln in lines[i0 + 1:i1]])
    raw = pk[:, 1, 2, 3].astype(f32) * f32(0.01) + f32(20.0)              # scale and offset on every value, missing or not (:3016)
    assert rows.shape == (NPT, 2) and np.array_equal(rows[:, 0].astype(f32), (np.array([43200.0, 129600.0, 216000.0], f32) / f32(86400.0)))
    assert np.array_equal(rows[:, 1].astype(f32), raw)
    assert out["ncopen"] == "1"
    v, units, comment = out["edited"].split(None, 2)
    assert f32(v) == f32(77.0) and units == "new" or out["edited"].split()[1:3] == ["new", "units"]
    rp = nums(out["reputvar"], (NPJ, NPI))                           # a 2 x 2 window of sevens over a field never written before
    assert np.all(rp[1:3, 2:4] == 7) and out["reputvar_att"].split()[0] == "9" and f32(out["reputvar_att"].split()[1]) == f32(5.0)
    f = netcdf_file(os.path.join(d, "edit.nc"), "r", mmap=False)
    fld, tc = f.variables["field"], f.variables["time_counter"]
    assert fld.units == b"new units" and fld.long_name == b"new long" and fld.short_name == b"field" and fld._FillValue == f32(77.0)
    assert fld.comment == b"edited twice" and fld.factor == f32(2.5)
    assert (tc.calendar, tc.title, tc.long_name, tc.axis, tc.units, tc.time_origin) == (b"gregorian", b"Time!", b"the axis of time", b"T",
                                                                                           b"days since then", b"then")
    f.close()


def test_written_files(run):
    d, out, F, _ = run
    f = netcdf_file(os.path.join(d, "out.nc"), "r", mmap=False)
    assert list(f.dimensions) == ["x", "y", "deptht", "time_counter"] and f.dimensions["time_counter"] is None
    assert f.history == b"made by the test" and f.start_date == -1 and f.CONFIG == b"N/A"
    d2 = (np.arange(1, NPI * NPJ + 1) * 0.25).reshape(NPJ, NPI)
    cube = f.variables["cube"]
    assert cube.data.dtype == np.dtype(">f8") and cube.dimensions == ("time_counter", "deptht", "y", "x")
    assert np.array_equal(cube[1], d2[None] * np.arange(1, NPK + 1)[:, None, None]) and cube._FillValue == -1.0
    plane = f.variables["plane"]                              # an r4 variable with a scale factor is stored as short
    assert plane.data.dtype == np.dtype(">i2") and plane.iweight == 5 and plane.scale_factor == f32(0.5)
    assert np.array_equal(plane[0], d2.astype(f32).astype(np.int16))
    assert [a for a in plane._attributes] == ["units", "_FillValue", "valid_min", "valid_max", "long_name", "short_name", "iweight",
                                              "online_operation", "axis", "scale_factor", "add_offset", "savelog10"]
    assert np.array_equal(f.variables["count"][1], (4 * d2).astype(np.int16))
    assert np.array_equal(f.variables["scalar"][:], np.array([5.5, 6.5], f32)) and f.variables["scalar"].dimensions == ("time_counter",)
    assert np.array_equal(f.variables["time_counter"][:], [86400.0, 172800.0]) and f.variables["time_counter"].calendar == b"noleap"
    assert np.array_equal(f.variables["nav_lon"][:], np.arange(NPJ * NPI, dtype=f32).reshape(NPJ, NPI))
    assert np.array_equal(f.variables["deptht"][:], np.array([5, 15, 27.5, 45], f32))
    f.close()
    f = netcdf_file(os.path.join(d, "zonal.nc"), "r", mmap=False)
    assert f.dimensions["x"] == 1 and "depthw" in f.dimensions and f.variables["depthw"].units == b"m"
    line = f.variables["line"]
    assert np.array_equal(line[0, 2, :, 0], np.arange(1, 6, dtype=f32)) and np.array_equal(f.variables["nav_lat"][:, 0], np.full(NPJ, 2, f32))
    assert f.variables["time_counter"].units == b"seconds since 0001-01-01" and f.variables["nav_lon"].units == b"degrees_east"
    f.close()


CMIP6 = dict(cn_output="CMIP6", cn_votemper="thetao", cn_vosaline="so", cn_vozocrtx="uo", cn_vomecrty="vo", cn_vovecrtz="wo", cn_sossheig="zos",
             cn_ve3tvvl="thkcello", cn_ve3wvvl="thkcello", cn_ve3uvvl="thkcello", cn_ve3vvvl="thkcello", cn_sozotaux="tauuo",
             cn_sohefldo="hfds", cn_vozout="uto", cn_vomevt="vto", cn_vozous="uso", cn_vomevs="vso", cn_sozoutrp="umo_vint",
             cn_iicethic3="iicethic", cn_ileadfra3="iiceconc", cn_votemper2="thetao_sqd", cn_vosaline2="so_sqd", cn_vozocrtx2="u2o",
             cn_vomecrty2="v2o", cn_vovecrtz2="w2o", cn_sossheig2="zossq", cn_somxldep="mlotst",
             # untouched by the switch
             cn_x="x", cn_vdeptht="deptht", cn_ve3t0="e3t_0", cn_somxl010="somxl010", cn_vomeeivv="vomeeivv")


def test_key_CMIP6_name_set(tmp_path):
    """-Dkey_CMIP6: the defaults of src/modcdfnames_CMIP6.h90 (values only; the identifiers are the module's)."""
    d = str(tmp_path)
    prog = "PROGRAM names\n  USE modcdfnames\n" + "".join(f"  PRINT '(a,1x,a)', '{k}', TRIM({k})\n" for k in CMIP6) + "END PROGRAM names\n"
    open(os.path.join(d, "names.f90"), "w").write(prog)
    r = subprocess.run(["amdflang", *FFLAGS, "-Dkey_CMIP6", "-module-dir", d, os.path.join(FDIR, "modcdfnames.F90"), os.path.join(d, "names.f90"),
                        "-o", os.path.join(d, "names")], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([os.path.join(d, "names")], capture_output=True, text=True, cwd=d, env=dict(os.environ, HOME=d))
    assert r.returncode == 0, r.stderr
    got = dict(ln.split() for ln in r.stdout.splitlines())
    assert got == CMIP6
    # and without the key the standard names (the build every tool uses)
    r = subprocess.run(["amdflang", *FFLAGS, "-module-dir", d, os.path.join(FDIR, "modcdfnames.F90"), "-c", "-o", os.path.join(d, "std.o")],
                       capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr
    text = open(os.path.join(FDIR, "modcdfnames.F90")).read()
    assert "CMIP6_('votemper', 'thetao')" in text and "CMIP6_('sossheig', 'zos')" in text


REFSRC = "/root/reference/src"
TOOLS = ["cdfmoc", "cdfheatc", "cdfcurl", "cdfsig0", "cdfsigi", "cdfpsi", "cdfmhst", "cdfvT", "cdfvhst", "cdfvtrp", "cdfmean", "cdfzonalsum",
         "cdfzonalmean", "cdfvint", "cdfvertmean", "cdfmocsig", "cdfmaxmoc", "cdfsum"]


@pytest.mark.skipif(not os.path.isdir(REFSRC), reason="the reference tree is only in the build container")
def test_the_reference_versions_of_the_tools_compile_against_this_module_set(tmp_path):
    """The module surface is a contract with the PROGRAM units: the reference's own source of each of the 18 tools of this
    build is checked (amdflang -fsyntax-only: names, generics, argument lists, keywords, types) against the .mod files of
    cdftools_amd/fortran/build, read where it lies; nothing is copied."""
    from concurrent.futures import ThreadPoolExecutor

    def check(p):
        r = subprocess.run(["amdflang", "-fsyntax-only", "-cpp", f"-I{BUILD}", "-module-dir", str(tmp_path), os.path.join(REFSRC, p + ".f90")],
                           capture_output=True, text=True)
        return p, r.returncode, r.stderr

    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(check, TOOLS))
    bad = {p: [ln for ln in err.splitlines() if "error:" in ln][:2] for p, rc, err in res if rc != 0}
    assert not bad, bad
