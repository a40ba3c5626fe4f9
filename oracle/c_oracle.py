"""Loader for oracle/dipole_oracle.c (fp64 arbiter).  TEST INFRASTRUCTURE ONLY - see the header
of dipole_oracle.c.  build() compiles it with gcc -fopenmp into oracle/libdipole_oracle.so."""
import ctypes
import os
import subprocess

import numpy as np

from .dipole_oracle import source_leaves

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "dipole_oracle.c")
LIB = os.path.join(_HERE, "libdipole_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"], check=True)
    return LIB


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        i64, p, d = ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
        _lib.oracle_field_grad_f64.argtypes = [p, i64, i64, p, i64, i64, d, p, i64, p]
        _lib.oracle_potential_f64.argtypes = [p, i64, i64, p, i64, i64, p, i64, p]
        _lib.oracle_field_grad_f64.restype = None
        _lib.oracle_potential_f64.restype = None
    return _lib


def _prep(sources, means, recursive, max_pts):
    src = np.ascontiguousarray(np.asarray(sources, dtype=np.float64))
    tgt = np.ascontiguousarray(np.asarray(means, dtype=np.float64))
    leaves = source_leaves(src.shape[0], max_pts if recursive else 0)
    off = np.array([leaves[0][0]] + [hi for _, hi in leaves], dtype=np.int64) if leaves else np.zeros(1, np.int64)
    return src, tgt, off


def field_grad_f64(sources, means, eps=1e-5, recursive=True, max_pts=15000):
    """fp64 field of float/double clouds given as numpy arrays [S,>=6], [T,>=3] -> [T,3] float64."""
    src, tgt, off = _prep(sources, means, recursive, max_pts)
    out = np.zeros((tgt.shape[0], 3), dtype=np.float64)
    if tgt.shape[0] and src.shape[0]:
        _load().oracle_field_grad_f64(src.ctypes.data, src.shape[0], src.shape[1], tgt.ctypes.data, tgt.shape[0],
                                      tgt.shape[1], float(eps), off.ctypes.data, len(off) - 1, out.ctypes.data)
    return out


def potential_f64(sources, means, recursive=True, max_pts=15000):
    src, tgt, off = _prep(sources, means, recursive, max_pts)
    out = np.zeros(tgt.shape[0], dtype=np.float64)
    if tgt.shape[0] and src.shape[0]:
        _load().oracle_potential_f64(src.ctypes.data, src.shape[0], src.shape[1], tgt.ctypes.data, tgt.shape[0],
                                     tgt.shape[1], off.ctypes.data, len(off) - 1, out.ctypes.data)
    return out
