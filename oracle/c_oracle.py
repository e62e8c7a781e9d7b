"""ctypes loader for the plain-C oracle (oracle/c/tet_energy_oracle.c).  TEST INFRASTRUCTURE."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtet_energy_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c", "tet_energy_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        lib.tso_face_adjacency.restype = ctypes.c_int
        lib.tso_face_adjacency.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        lib.tso_energy_grad.restype = ctypes.c_int
        lib.tso_energy_grad.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        lib.tso_rounding_model.restype = ctypes.c_int
        lib.tso_rounding_model.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


def face_adjacency(tets: np.ndarray) -> np.ndarray:
    t = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
    nbr = np.empty_like(t)
    rc = _load().tso_face_adjacency(t.shape[0], t.ctypes.data, nbr.ctypes.data)
    if rc:
        raise ValueError(f"tso_face_adjacency failed with code {rc}")
    return nbr


def energy_and_grad(rest, tets, x, c1, c2, order, grad_output=1.0, want_grad=True, nbr=None):
    rest = np.ascontiguousarray(rest, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3)
    E = np.zeros(3)
    g = np.empty((rest.shape[0], 3)) if want_grad else None
    nb = None if nbr is None else np.ascontiguousarray(nbr, dtype=np.int32)
    rc = _load().tso_energy_grad(rest.shape[0], t.shape[0], rest.ctypes.data, t.ctypes.data,
                                 None if nb is None else nb.ctypes.data, x.ctypes.data,
                                 float(c1), float(c2), int(order), float(grad_output),
                                 E.ctypes.data, None if g is None else g.ctypes.data)
    if rc:
        raise ValueError(f"tso_energy_grad failed with code {rc}")
    return float(E[0]), float(E[1]), float(E[2]), g


def rounding_error_model(rest, tets, x, c1, c2, order, nbr=None):
    """``(std_E, std_g[n])`` -- the plain-C twin of ``tet_energy_oracle.rounding_error_model`` (uniform element
    Laplacian), for scenes the numpy model cannot hold in memory."""
    rest = np.ascontiguousarray(rest, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3)
    std_e = np.zeros(1)
    std_g = np.empty(rest.shape[0])
    nb = None if nbr is None else np.ascontiguousarray(nbr, dtype=np.int32)
    rc = _load().tso_rounding_model(rest.shape[0], t.shape[0], rest.ctypes.data, t.ctypes.data,
                                    None if nb is None else nb.ctypes.data, x.ctypes.data,
                                    float(c1), float(c2), int(order), std_e.ctypes.data, std_g.ctypes.data)
    if rc:
        raise ValueError(f"tso_rounding_model failed with code {rc}")
    return float(std_e[0]), std_g
