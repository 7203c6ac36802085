"""ctypes loader for oracle/lsap_oracle.c.  TEST INFRASTRUCTURE ONLY (see the C file's header)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "liblsap_oracle.so"
_lib = None


def build():
    if not _SO.exists() or _SO.stat().st_mtime < (_HERE / "lsap_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "liblsap_oracle.so"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_SO))
        _lib.lsap_oracle_f32.restype = C.c_int
        _lib.lsap_oracle_f32.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.lsap_oracle_mask_f32.restype = C.c_int
        _lib.lsap_oracle_mask_f32.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def linear_sum_assignment_f32(cost: np.ndarray):
    cost = np.ascontiguousarray(cost, np.float32)
    nr, nc = cost.shape
    k = min(nr, nc)
    rows, cols = np.zeros(k, np.int64), np.zeros(k, np.int64)
    rc = lib().lsap_oracle_f32(nr, nc, cost.ctypes.data, rows.ctypes.data, cols.ctypes.data)
    if rc != 0:
        raise ValueError("cost matrix is infeasible" if rc == -1 else "matrix contains invalid numeric entries")
    return rows, cols


def assignment_mask(cost: np.ndarray, num_objects: np.ndarray) -> np.ndarray:
    cost = np.ascontiguousarray(cost, np.float32)
    nobj = np.ascontiguousarray(num_objects, np.int32)
    B, M, N = cost.shape
    mask = np.zeros_like(cost)
    rc = lib().lsap_oracle_mask_f32(B, M, N, cost.ctypes.data, nobj.ctypes.data, mask.ctypes.data)
    if rc != 0:
        raise ValueError("lsap oracle failed")
    return mask
