"""ctypes access to oracle/liboracle.so -- the CPU restatement (checker only; tests, smoke, bench
cpu_baseline).  Same reversed-dimension array convention as cdftools_amd/hip.py."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "cdf_oracle.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(LIB)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def moc(e1v, e3v, ibmask, zv):
    """e1v (npj,npi), e3v (npk,npj,npi), ibmask (npj,npi,nb) i2, zv (npt,npk,npj,npi)
    -> dmoc (npt,npk,npj,nb) r8, one oracle_moc call per record."""
    lib = load()
    e1v = np.ascontiguousarray(e1v, np.float32)
    e3v = np.ascontiguousarray(e3v, np.float32)
    ibmask = np.ascontiguousarray(ibmask, np.int16)
    zv = np.ascontiguousarray(zv, np.float32)
    npk, npj, npi = e3v.shape
    nb = ibmask.shape[-1]
    zv = zv.reshape(-1, npk, npj, npi)
    out = np.empty((zv.shape[0], npk, npj, nb), np.float64)
    for t in range(zv.shape[0]):
        lib.oracle_moc(C.c_int(npi), C.c_int(npj), C.c_int(npk), C.c_int(nb), _p(e1v), _p(e3v),
                       _p(ibmask), _p(zv[t]), _p(out[t]))
    return out


def num_threads():
    return load().oracle_num_threads()
