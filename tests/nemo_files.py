"""Writes a synthetic NEMO file set (classic NetCDF via scipy.io.netcdf_file, an implementation
independent of the bundled backend) in the v3.6 mesh layout the tools expect:
mesh_hgr.nc, mesh_zgr.nc, mask.nc, new_maskglo.nc and <CONF>_<TAG>_grid{T,U,V}.nc data files."""
import os

import numpy as np
from scipy.io import netcdf_file

import synth


def _new(path, dims, version=2):
    f = netcdf_file(path, "w", version=version)
    f.createDimension(dims[0][0], None)                      # scipy wants the record dimension first
    for name, n in dims[1:]:
        f.createDimension(name, n)
    return f


def _var(f, name, dtype, dims, data, **atts):
    v = f.createVariable(name, dtype, dims)
    v[:] = data
    for k, val in atts.items():
        setattr(v, k, val)
    return v


def write_mesh(d, g):
    npk, npj, npi = g["tmask"].shape
    dims = [("t", None), ("z", npk), ("y", npj), ("x", npi)]
    with _new(os.path.join(d, "mesh_hgr.nc"), dims) as f:
        for nm in ("e1t", "e2t", "e1u", "e2u", "e1v", "e2v", "e1f", "e2f", "glamf", "gphif", "gphiv"):
            _var(f, nm, "f4", ("t", "y", "x"), g[nm][None])
        for nm in ("gphit", "gphiu"):                                    # same latitudes on this synthetic grid
            _var(f, nm, "f4", ("t", "y", "x"), g["gphif"][None])
    with _new(os.path.join(d, "mesh_zgr.nc"), dims) as f:
        for nm in ("e3t_0", "e3u_0", "e3v_0"):
            _var(f, nm, "f4", ("t", "z", "y", "x"), g[nm][None])
        _var(f, "gdepw_1d", "f4", ("t", "z"), g["gdepw_1d"][None])
        _var(f, "gdept_1d", "f4", ("t", "z"), g["gdept_1d"][None])
        _var(f, "e3t_1d", "f4", ("t", "z"), g["e3t_1d"][None])
        hdepw = (g["e3t_0"] * g["tmask"]).sum(axis=0).astype(np.float32)          # depth of the sea floor
        _var(f, "hdepw", "f4", ("t", "y", "x"), hdepw[None])
    with _new(os.path.join(d, "mask.nc"), dims) as f:
        for nm in ("tmask", "umask", "vmask", "fmask"):
            _var(f, nm, "i1", ("t", "z", "y", "x"), g[nm][None].astype(np.int8))


def write_basins(d, g):
    npj, npi = g["e1t"].shape
    f = netcdf_file(os.path.join(d, "new_maskglo.nc"), "w", version=2)
    f.createDimension("y", npj)
    f.createDimension("x", npi)
    for nm in ("tmaskatl", "tmaskpac", "tmaskind"):
        _var(f, nm, "i1", ("y", "x"), g[nm].astype(np.int8))
    f.close()


def write_data(path, g, depname, fields, times=None, fill=None):
    """fields: {name: (npt,npk,npj,npi) or (npt,npj,npi) array}."""
    npk, npj, npi = g["tmask"].shape
    first = next(iter(fields.values()))
    npt = first.shape[0]
    times = np.arange(1, npt + 1) * 86400.0 * 30 if times is None else times
    with _new(path, [("time_counter", None), (depname, npk), ("y", npj), ("x", npi)]) as f:
        _var(f, "nav_lon", "f4", ("y", "x"), g["glamf"])
        _var(f, "nav_lat", "f4", ("y", "x"), g["gphif"])
        dep = g["gdepw_1d"] if depname.endswith("w") else g["gdept_1d"]
        _var(f, depname, "f4", (depname,), dep, units="m")
        _var(f, "time_counter", "f8", ("time_counter",), times, units="seconds since 1950-01-01 00:00:00",
             calendar="noleap", time_origin="1950-JAN-01 00:00:00")
        for name, a in fields.items():
            dims = ("time_counter", depname, "y", "x") if a.ndim == 4 else ("time_counter", "y", "x")
            atts = {} if fill is None else {"_FillValue": np.float32(fill)}
            _var(f, name, "f4", dims, a.astype(np.float32), **atts)


def make_case(d, npi=24, npj=18, npk=7, npt=2, seed=20240601, basins=True):
    """-> (grid dict, {'v','u','t','s'} records) with every file written under directory d.
    CDFT_TEST_SHAPE="npi,npj,npk" in the environment replaces the grid size every end-to-end test asks for: the program
    tests then run on other shapes (plane sizes that are not multiples of 4, rows longer than a tile, ...) unchanged
    (at least 20 x 15 x 4: some of them pass zoom windows).  Run on the GPU box, e.g.
    CDFT_TEST_SHAPE=1443,37,4 python -m pytest tests/test_cli.py tests/test_cli_tools.py tests/test_cli_next.py tests/test_cli_sharded.py -m gpu"""
    if os.environ.get("CDFT_TEST_SHAPE") and (npi, npj, npk) in ((24, 18, 7), (26, 19, 6)):     # (the end-to-end tests' two grids)
        npi, npj, npk = (int(x) for x in os.environ["CDFT_TEST_SHAPE"].split(","))
    g = synth.make_grid(npi, npj, npk, seed=seed)
    write_mesh(d, g)
    if basins:
        write_basins(d, g)
    rng = np.random.default_rng(seed + 7)
    v = synth.velocity(g, npt, seed + 1)
    u = synth.velocity(g, npt, seed + 2, which="u")
    t = ((-2 + 32 * rng.random((npt, npk, npj, npi), dtype=np.float32)) * g["tmask"][None]).astype(np.float32)
    s = ((30 + 8 * rng.random((npt, npk, npj, npi), dtype=np.float32)) * g["tmask"][None]).astype(np.float32)
    write_data(os.path.join(d, "SYN_y0001_gridV.nc"), g, "depthv", {"vomecrty": v})
    write_data(os.path.join(d, "SYN_y0001_gridU.nc"), g, "depthu", {"vozocrtx": u})
    write_data(os.path.join(d, "SYN_y0001_gridT.nc"), g, "deptht", {"votemper": t, "vosaline": s}, fill=0.0)
    return g, {"v": v, "u": u, "t": t, "s": s}
