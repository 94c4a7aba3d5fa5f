"""Oracle (test infrastructure): ctypes loader for oracle/csrc/oracle.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-f", os.path.join(_HERE, "Makefile")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_project.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        _lib.orc_project.restype = None
        _lib.orc_nn_exact.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                      ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p]
        _lib.orc_nn_exact.restype = None
        _lib.orc_round_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        _lib.orc_round_bf16.restype = None
    return _lib


def project_fma(pt_array, cam):
    """get_render_location with the k-sequential fma chain (bit-exact twin of the kernel)."""
    pts = np.ascontiguousarray(pt_array, dtype=np.float64)
    cam = np.ascontiguousarray(cam, dtype=np.float64)
    out = np.empty((3, pts.shape[0]), dtype=np.float64)
    lib().orc_project(pts.ctypes.data, pts.shape[0], cam.ctypes.data, out.ctypes.data)
    return out


def nn_exact(q, r, k=1, index_base=1):
    """q (C,Nq) f32, r (C,Nr) f32 -> idx (k,Nq) int32, d2 (k,Nq) f64; exact float64 ordering."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    r = np.ascontiguousarray(r, dtype=np.float32)
    c, nq = q.shape
    c2, nr = r.shape
    assert c == c2
    idx = np.empty((k, nq), dtype=np.int32)
    d2 = np.empty((k, nq), dtype=np.float64)
    lib().orc_nn_exact(q.ctypes.data, r.ctypes.data, c, nq, nr, k, index_base,
                       idx.ctypes.data, d2.ctypes.data)
    return idx, d2


def round_bf16(x):
    """Round-to-nearest-even to bfloat16 precision, returned as float32."""
    y = np.array(x, dtype=np.float32, order="C", copy=True)
    lib().orc_round_bf16(y.ctypes.data, y.size)
    return y
