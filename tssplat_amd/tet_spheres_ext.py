"""Drop-in for the reference's native module ``tet_spheres.tet_spheres_ext``.

Same names, argument order and meaning as the pybind11 module defined at
/root/reference/tssplat_ext/tet_spheres/tet_spheres.cpp:225-266:

    TetSpheres(filename) | TetSpheres(vertices_f32_1d, elements_i32_1d)
    forward(input, tet_sph, c1, c2, order)            -> 0-dim float32 tensor
    backward(gradH, input, tet_sph, c1, c2, order)    -> tensor like input
    random_x(tet_sph)                                  -> CPU tensor [n, 3]
    grad_limit(grad, s_threshold, s)                   -> None (in place)

so that /root/reference/energies/smooth_barrier.py works unmodified on top of
it (``from tet_spheres import tet_spheres_ext``; the top-level ``tet_spheres``
package in this repo re-exports this module).

The arithmetic runs in hand-written gfx950 kernels behind the C ABI of
``include/tssplat_amd.h``; PyTorch only supplies device memory and the stream.
There is no CPU path: a missing library or a CPU tensor raises.

Deliberate differences from the reference (all listed in DESIGN.md):

* ``forward`` returns the energy on ``input``'s device and never blocks the host
  (the reference returns a CPU 0-dim tensor after two blocking reads,
  tet_spheres_cuda.cu:154,185,194).  Set ``TSSPLAT_AMD_CPU_ENERGY=1`` before import
  (or ``tet_spheres_ext.CPU_ENERGY = True``) for the reference's CPU return.  ``backward`` accepts ``gradH`` on either device and
  applies it on the GPU without ``.item()`` (contrast .cu:257).
* when ``input.requires_grad`` the forward call
  evaluates energy *and* gradient in one fused pass and keeps the unscaled
  gradient on the ``TetSpheres`` object; the matching ``backward`` call only
  multiplies it by ``gradH``.  The cache key is ``(data_ptr, _version, shape, c1,
  c2, order)``; any miss recomputes, and every ``forward`` replaces or drops the
  entry (except an explicit energy-only ``forward(..., fuse=False)``, which leaves it alone).  The key cannot see writes that bypass torch's version counter (``.data``
  ops, the raw-pointer ``AdamUniform`` step): the cache assumes that ``backward``
  follows ITS ``forward`` with no such write in between, which is what
  ``loss.backward()`` does.  The reference recomputes ``G x`` in backward (.cu:221).
* a 2-D vertex/element array makes the reference print to stderr and hand back
  an empty, unusable object (tet_spheres.cpp:238-250); we print the same line
  and return an object whose use raises ``RuntimeError`` instead of crashing.
* ``grad_limit`` implements the intended clamp (utils/optimizer.py:84-86), not
  the shipped ``grad[0]`` bug (.cu:278), and does not print the shape.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import sys

import numpy as np
import torch

from . import _capi

__all__ = ["TetSpheres", "forward", "backward", "random_x", "grad_limit"]

_lib = _capi.load()          # fail loudly at import if the HIP library is absent
CPU_ENERGY = os.environ.get("TSSPLAT_AMD_CPU_ENERGY", "0") == "1"   # read once: os.environ lookups are slow
# TetSpheres(rebuild_dminv=None) = the library's choice (tsamd_options.rebuild_dminv 0: stream the exactly rounded Dm^-1 planes, except
# for mid-size batches); TSSPLAT_AMD_REBUILD_DMINV=1 / 2 makes "always rebuild" / "always stream" the default instead
REBUILD_DMINV = {"1": 1, "2": 2}.get(os.environ.get("TSSPLAT_AMD_REBUILD_DMINV", "0"), 0)
print("initializing")         # tet_spheres.cpp:19 prints this at module import


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device` (the raw query is ~10x cheaper than building a
    torch.cuda.Stream object on every call)."""
    if _raw_stream is not None and device.index is not None:
        return int(_raw_stream(device.index))
    return int(torch.cuda.current_stream(device).cuda_stream)


_NULL_CTX = contextlib.nullcontext()


def _device_ctx(device: torch.device):
    """Handle-less entry points (scale, grad_limit) launch on the current device: switch only if needed."""
    return _NULL_CTX if torch.cuda.current_device() == device.index else torch.cuda.device(device)


class TetSpheres:
    """Device state for one batch of tet-spheres (reference: tet_spheres.h:9-42).

    ``TetSpheres(filename)`` loads a Vega ``.veg`` tet mesh (tet_spheres.cpp:108-117);
    ``TetSpheres(vertices, elements)`` takes flat float32 ``[3n]`` / int32 ``[4m]``
    arrays, 0-based (tet_spheres.cpp:234-258, caller energies/smooth_barrier.py:38-40).
    Keyword-only extras select the HIP device and tune the tiler.  ``operator`` (array constructor only)
    replaces the assumed uniform face-adjacency umbrella by an explicit element operator ``L`` -- a scipy
    sparse matrix or a ``(rowptr, col, val)`` CSR triple over tets, e.g. the matrix libpgo really builds
    (tet_spheres.cpp:148; see tools/pin_L_with_pypgo.py) -- through ``tsamd_create_with_operator``.
    """

    def __init__(self, vertices=None, elements=None, *, device=None, host_only: bool = False,
                 lds_budget_bytes: int = 0, max_threads: int = 0, target_owned: int = 0,
                 num_threads: int = 0, debug_flags: int = 0, lane_search_sweeps: int = 0,
                 slots_per_thread: int = 0, operator=None, rebuild_dminv: bool | None = None):
        self._h = C.c_void_p()
        self.n = self.nele = self.n3 = 0
        self._cache = None
        self.fuse_forward_backward = True
        self.device = None
        if vertices is None and elements is None:
            return                                  # TetSpheres() {} -- tet_spheres.h:14
        if not host_only:
            if not torch.cuda.is_available():
                raise RuntimeError("tssplat_amd: no HIP device visible; there is no CPU fallback "
                                   "(pass host_only=True to build the tiling plan only)")
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            if dev.type != "cuda":
                raise RuntimeError("tssplat_amd: device must be a HIP ('cuda') device")
            self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        opts = _capi.make_options(device=-1 if self.device is None else self.device.index,
                                  host_only=int(host_only), lds_budget_bytes=lds_budget_bytes,
                                  max_threads=max_threads, target_owned=target_owned,
                                  num_threads=num_threads,
                                  debug_flags=int(debug_flags), lane_search_sweeps=int(lane_search_sweeps), slots_per_thread=slots_per_thread,
                                  # (the environment default only applies where it can; an explicit True that cannot be
                                  # honoured -- together with operator= -- is rejected by the library, not ignored)
                                  rebuild_dminv=(REBUILD_DMINV if operator is None or REBUILD_DMINV != 1 else 0) if rebuild_dminv is None
                                  else (1 if rebuild_dminv else 2))
        if isinstance(vertices, (str, os.PathLike)) and elements is None:
            if operator is not None:
                raise TypeError("operator= needs the (vertices, elements) constructor")
            rc = _lib.tsamd_create_from_veg(os.fspath(vertices).encode(), C.byref(opts), C.byref(self._h))
            _capi.check(rc)
        else:
            v = np.asarray(vertices)
            f = np.asarray(elements)
            if v.ndim != 1:
                print(f"Wrong vertex type:{v.ndim},{v.dtype.char}", file=sys.stderr)
                return
            if f.ndim != 1:
                print(f"Wrong tet type:{f.ndim},{f.dtype.char},q", file=sys.stderr)
                return
            v = np.ascontiguousarray(v, dtype=np.float32)      # py::array::forcecast
            f = np.ascontiguousarray(f, dtype=np.int32)
            if operator is None:
                rc = _lib.tsamd_create(v.ctypes.data, v.size // 3, f.ctypes.data, f.size // 4,
                                       C.byref(opts), C.byref(self._h))
            else:
                if hasattr(operator, "tocsr"):
                    csr = operator.tocsr()
                    if csr.shape != (f.size // 4, f.size // 4):
                        raise ValueError(f"operator must be {f.size // 4} x {f.size // 4} (tets x tets), got {csr.shape}")
                    operator = (csr.indptr, csr.indices, csr.data)
                rp = np.ascontiguousarray(operator[0], dtype=np.int64)
                ci = np.ascontiguousarray(operator[1], dtype=np.int32)
                va = np.ascontiguousarray(operator[2], dtype=np.float64)
                if rp.size != f.size // 4 + 1 or ci.size != va.size or (rp.size and rp[-1] != ci.size):
                    raise ValueError("operator: inconsistent CSR arrays")
                rc = _lib.tsamd_create_with_operator(v.ctypes.data, v.size // 3, f.ctypes.data, f.size // 4,
                                                     rp.ctypes.data, ci.ctypes.data, va.ctypes.data,
                                                     C.byref(opts), C.byref(self._h))
            _capi.check(rc)
        self.n = int(_lib.tsamd_num_vertices(self._h))
        self.nele = int(_lib.tsamd_num_tets(self._h))
        self.n3 = 3 * self.n

    # -- lifetime ---------------------------------------------------------- #
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.tsamd_destroy(self._h)
            self._h = C.c_void_p()
        self._cache = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if not self._h.value:
            raise RuntimeError("TetSpheres object is empty (construction failed or it was closed)")
        return self._h

    # -- introspection ----------------------------------------------------- #
    def plan_info(self) -> dict:
        info = _capi.PlanInfo()
        _capi.check(_lib.tsamd_get_plan_info(self._handle(), C.byref(info)))
        return info.as_dict()

    def set_timing(self, enable: bool) -> None:
        """Record HIP events around the kernels of every evaluation (bench.py roofline leg)."""
        _capi.check(_lib.tsamd_set_timing(self._handle(), int(enable)))

    def get_timing(self) -> tuple[float, float, int]:
        """(tile kernel ms, finish kernel ms, evaluations) since the last call; blocks the host."""
        a, b, n = C.c_double(), C.c_double(), C.c_int64()
        _capi.check(_lib.tsamd_get_timing(self._handle(), C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def energy_terms(self) -> tuple[float, float]:
        """(E_s, E_b) of the last evaluation, in double.  Blocks the host."""
        out = (C.c_double * 2)()
        _capi.check(_lib.tsamd_read_energy_terms(self._handle(), _stream_ptr(self.device), out))
        return float(out[0]), float(out[1])


def _check_input(x: torch.Tensor, ts: TetSpheres) -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        raise TypeError("input must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError("tssplat_amd: input must live on the GPU; there is no CPU fallback")
    if ts.device is None or x.device != ts.device:
        raise RuntimeError(f"input is on {x.device} but the TetSpheres object lives on {ts.device}")
    if x.dtype != torch.float32:
        raise RuntimeError("input must be float32")
    if x.numel() != ts.n3:
        raise RuntimeError(f"input has {x.numel()} elements, expected {ts.n3} (= 3 * n_vertices)")
    return x if x.is_contiguous() else x.detach().contiguous()   # tet_spheres_cuda.cu:124 (only data_ptr is used)


def _cache_key(x: torch.Tensor, c1: float, c2: float, order: int):
    # (a miss only costs a recomputation, so the coefficients are compared as given, not after fp32 narrowing)
    return (x.data_ptr(), x._version, x.shape, float(c1), float(c2), int(order))


def cpu_energy_mode() -> bool:
    """True while ``forward`` returns the energy as a CPU tensor, the reference's convention (see the module docstring)."""
    return CPU_ENERGY


def forward(input: torch.Tensor, tet_sph: TetSpheres, c1: float, c2: float, order: int, *,
            fuse: bool | None = None) -> torch.Tensor:
    """Energy ``c1 * 1/2 |L G x|^2 + c2 * sum_e max(-det F_e, 0)^order`` (tet_spheres_cuda.cu:118-195).

    ``fuse`` (keyword-only, not in the reference): ``None`` = ``tet_sph.fuse_forward_backward``; ``False`` = energy only,
    whatever ``input.requires_grad`` says (logging / validation calls); such a call leaves a fused gradient kept for a
    pending ``backward`` alone."""
    h = tet_sph._handle()
    x = _check_input(input, tet_sph)
    energy = torch.empty((), dtype=torch.float32, device=x.device)
    stream = _stream_ptr(x.device)
    # (no torch.cuda.device() guard: the library switches to the handle's device itself)
    # (grad mode cannot be consulted here: inside autograd.Function.forward it is always off.  Callers that evaluate
    # under torch.no_grad() switch the fusion off themselves -- SmoothnessBarrierEnergy.forward does.)
    if input.requires_grad and (tet_sph.fuse_forward_backward if fuse is None else fuse):
        g = torch.empty_like(x)
        _capi.check(_lib.tsamd_forward_backward(h, x.data_ptr(), None, c1, c2, int(order), stream,
                                                energy.data_ptr(), g.data_ptr()))
        tet_sph._cache = (_cache_key(input, c1, c2, order), g)
    else:
        # An explicit energy-only call (fuse=False: logging / validation) leaves a kept gradient alone -- placed between
        # `loss = energy(x)` and `loss.backward()` it must not cost that backward a second full pass.  Every other
        # non-fused evaluation drops the entry, as before.
        if fuse is None:
            tet_sph._cache = None
        _capi.check(_lib.tsamd_forward(h, x.data_ptr(), c1, c2, int(order), stream, energy.data_ptr()))
    if CPU_ENERGY:
        return energy.cpu()                             # the reference's convention, .cu:194
    return energy


def backward(gradH: torch.Tensor, input: torch.Tensor, tet_sph: TetSpheres, c1: float, c2: float,
             order: int) -> torch.Tensor:
    """``gradH * dE/dx`` with the shape/dtype/device of ``input`` (tet_spheres_cuda.cu:197-263)."""
    cached = tet_sph._cache
    tet_sph._cache = None
    hit = cached is not None and cached[0] == _cache_key(input, c1, c2, order)
    # a hit means `input` is the very tensor the fused forward validated a moment ago
    x = input if hit and input.is_contiguous() else _check_input(input, tet_sph)
    dev = x.device
    if isinstance(gradH, torch.Tensor) and gradH.device == dev and gradH.dtype == torch.float32 \
            and gradH.numel() == 1:
        go = gradH                                      # the usual case: autograd hands over a device scalar
    else:
        if not isinstance(gradH, torch.Tensor):
            gradH = torch.tensor(float(gradH), dtype=torch.float32)
        go = gradH.detach().to(device=dev, dtype=torch.float32, non_blocking=True).reshape(-1)[:1].contiguous()
    stream = _stream_ptr(dev)
    if hit:
        # the cached gradient is ours: scale it in place (the kernel returns at once when gradH == 1,
        # the usual case for a loss term, so no second pass over the gradient) and hand it over
        out = cached[1]
        with _device_ctx(dev):
            _capi.check(_lib.tsamd_scale(out.data_ptr(), go.data_ptr(), out.data_ptr(), out.numel(), stream))
    else:
        out = torch.empty_like(x)
        _capi.check(_lib.tsamd_backward(tet_sph._handle(), x.data_ptr(), go.data_ptr(), c1, c2, int(order), stream,
                                        out.data_ptr()))
    # (not a view when the shapes already agree: autograd can then take the buffer instead of cloning it)
    return out if out.shape == input.shape else out.view(input.shape)


def random_x(tet_sph: TetSpheres) -> torch.Tensor:
    """``torch.rand(n, 3)`` on the CPU (tet_spheres.cpp:218-221)."""
    return torch.rand(tet_sph.n, 3)


def grad_limit(grad: torch.Tensor, s_threshold: float, s: float) -> None:
    """In place: ``if max|grad| > s_threshold: grad *= s / max|grad|`` -- no host sync."""
    if not grad.is_cuda or grad.dtype != torch.float32 or not grad.is_contiguous():
        raise RuntimeError("grad_limit expects a contiguous float32 GPU tensor")
    ws = torch.empty(int(_lib.tsamd_grad_limit_workspace_bytes()), dtype=torch.uint8, device=grad.device)
    with _device_ctx(grad.device):
        _capi.check(_lib.tsamd_grad_limit(grad.data_ptr(), grad.numel(), float(s_threshold), float(s),
                                          ws.data_ptr(), _stream_ptr(grad.device)))
