"""ctypes loader for oracle/_ref/libref_det.so -- the reference's OWN det / ddetA_dA / penalty kernels
(/root/reference/tssplat_ext/tet_spheres/tet_spheres_cuda.cu:9-102) compiled for the host by
oracle/ref_recipe/Makefile.  TEST INFRASTRUCTURE: used to pin the oracle's restatement of those lines.

``available()`` is False where neither the built library nor /root/reference exists (the reference does not
travel to the GPU box; the built library does)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libref_det.so")
_REF = "/root/reference/tssplat_ext/tet_spheres/tet_spheres_cuda.cu"
_lib = None


def build(force: bool = False) -> str | None:
    """Compile oracle/_ref from the reference sources where they lie; None if the reference is absent."""
    if not os.path.exists(_REF):
        return _SO if os.path.exists(_SO) else None
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "ref_recipe")], stderr=subprocess.DEVNULL)
    return _SO


def available() -> bool:
    return os.path.exists(_SO) or os.path.exists(_REF)


def _load():
    global _lib
    if _lib is None:
        so = build()
        if so is None:
            raise FileNotFoundError("oracle/_ref/libref_det.so is missing and /root/reference is not here to build it")
        lib = ctypes.CDLL(so)
        lib.ref_det_f32.restype = ctypes.c_float
        lib.ref_det_f64.restype = ctypes.c_double
        for f in (lib.ref_det_f32, lib.ref_det_f64):
            f.argtypes = [ctypes.c_void_p]
        for f in (lib.ref_cof_f32, lib.ref_cof_f64):
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            f.restype = None
        for f in (lib.ref_forward_det, lib.ref_backward_det):
            f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
            f.restype = None
        _lib = lib
    return _lib


def det(M: np.ndarray) -> np.ndarray:
    """`det` (.cu:21-30) of every 9-vector of M[k, 9], at M's dtype (float32 or float64)."""
    lib = _load()
    M = np.ascontiguousarray(M)
    fn = lib.ref_det_f32 if M.dtype == np.float32 else lib.ref_det_f64
    return np.array([fn(M[k].ctypes.data) for k in range(M.shape[0])], dtype=M.dtype)


def ddetA_dA(M: np.ndarray) -> np.ndarray:
    """`ddetA_dA` (.cu:32-46) of every 9-vector of M[k, 9], at M's dtype."""
    lib = _load()
    M = np.ascontiguousarray(M)
    out = np.empty_like(M)
    fn = lib.ref_cof_f32 if M.dtype == np.float32 else lib.ref_cof_f64
    for k in range(M.shape[0]):
        fn(M[k].ctypes.data, out[k].ctypes.data)
    return out


def forward_det(F: np.ndarray, order: int) -> np.ndarray:
    """`cuda_forward_det` (.cu:48-66): per-tet penalty value, fp32."""
    F = np.ascontiguousarray(F, dtype=np.float32).reshape(-1, 9)
    out = np.empty(F.shape[0], dtype=np.float32)
    _load().ref_forward_det(F.shape[0], F.ctypes.data, out.ctypes.data, int(order))
    return out


def backward_det(F: np.ndarray, order: int) -> np.ndarray:
    """`cuda_backward_det` (.cu:68-102): per-tet d(penalty)/dF, fp32 [m, 9]."""
    F = np.ascontiguousarray(F, dtype=np.float32).reshape(-1, 9)
    out = np.empty_like(F)
    _load().ref_backward_det(F.shape[0], F.ctypes.data, out.ctypes.data, int(order))
    return out
