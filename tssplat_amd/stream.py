"""Streaming-tile evaluator (EXPERIMENTAL, round 3): the same energy and gradient as ``tet_spheres_ext.forward`` /
``backward`` (reference: /root/reference/tssplat_ext/tet_spheres/tet_spheres_cuda.cu:118-263) from another plan and
another tile kernel -- tubes swept band by band with a rolling LDS window and four stage-specialised wave groups
(csrc/stream_plan.h, csrc/stream_kernels.hip).  ``TetSpheres`` / ``tsamd_create`` stay the product path until this one has
measured faster; there is no CPU fallback here either.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from .tet_spheres_ext import _stream_ptr

__all__ = ["StreamTetSpheres"]

_lib = _capi.load()


class StreamTetSpheres:
    """``StreamTetSpheres(vertices f32[3n], elements i32[4m])`` -- built-in uniform operator only."""

    def __init__(self, vertices, elements, *, device=None, host_only: bool = False, num_threads: int = 0):
        self._h = C.c_void_p()
        v = np.ascontiguousarray(np.asarray(vertices).reshape(-1), dtype=np.float32)
        f = np.ascontiguousarray(np.asarray(elements).reshape(-1), dtype=np.int32)
        self.device = None
        if not host_only:
            if not torch.cuda.is_available():
                raise RuntimeError("tssplat_amd: no HIP device visible; there is no CPU fallback (host_only=True builds the plan only)")
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        _capi.check(_lib.tsamd_stream_create(v.ctypes.data, v.size // 3, f.ctypes.data, f.size // 4,
                                             -1 if self.device is None else self.device.index, int(host_only), int(num_threads),
                                             C.byref(self._h)))
        self.n, self.nele = v.size // 3, f.size // 4

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.tsamd_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def plan_info(self) -> dict:
        info = _capi.StreamPlanInfo()
        _capi.check(_lib.tsamd_stream_info(self._h, C.byref(info)))
        return info.as_dict()

    def tube(self, t: int):
        """Host view of tube ``t``: ``(n_bands, n_vslots, n_owned, n_slots, blob bytes as uint8 array, slot_tet [n_bands, band])``."""
        view = _capi.StreamTubeView()
        _capi.check(_lib.tsamd_stream_get_tube(self._h, int(t), C.byref(view)))
        blob = np.ctypeslib.as_array(view.blob, shape=(int(view.blob_bytes),))
        band = self.plan_info()["band_slots"]
        slot_tet = np.ctypeslib.as_array(view.slot_tet, shape=(int(view.n_bands), band))
        return int(view.n_bands), int(view.n_vslots), int(view.n_owned), int(view.n_slots), blob, slot_tet

    def finish_lists(self):
        nf, ns = C.c_int64(), C.c_int64()
        vid, off = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        _capi.check(_lib.tsamd_stream_get_finish_lists(self._h, C.byref(nf), C.byref(ns), C.byref(vid), C.byref(off)))
        k = int(nf.value)
        return (np.ctypeslib.as_array(vid, shape=(k,)).copy() if k else np.zeros(0, np.int32),
                np.ctypeslib.as_array(off, shape=(k + 1,)).copy(), int(ns.value))

    def forward_backward(self, x: torch.Tensor, c1: float, c2: float, order: int, grad_output: torch.Tensor | None = None):
        """``(energy 0-dim, grad like x)`` -- one fused evaluation, ``grad_output`` (device scalar) applied on the device."""
        if not x.is_cuda or x.dtype != torch.float32 or x.numel() != 3 * self.n or x.device != self.device:
            raise RuntimeError("StreamTetSpheres: x must be a float32 GPU tensor of 3 * n_vertices elements on the handle's device")
        x = x.contiguous()
        e = torch.empty((), dtype=torch.float32, device=x.device)
        g = torch.empty_like(x)
        go = None if grad_output is None else grad_output.to(device=x.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
        _capi.check(_lib.tsamd_stream_forward_backward(self._h, x.data_ptr(), None if go is None else go.data_ptr(), c1, c2, int(order),
                                                       _stream_ptr(x.device), e.data_ptr(), g.data_ptr()))
        return e, g

    def set_timing(self, enable: bool) -> None:
        _capi.check(_lib.tsamd_stream_set_timing(self._h, int(enable)))

    def get_timing(self):
        a, b, n = C.c_double(), C.c_double(), C.c_int64()
        _capi.check(_lib.tsamd_stream_get_timing(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def energy_terms(self):
        out = (C.c_double * 2)()
        _capi.check(_lib.tsamd_stream_read_energy_terms(self._h, _stream_ptr(self.device), out))
        return float(out[0]), float(out[1])
