"""The two nvdiffrast operators the reference's renderer uses, on the MI355X kernels (first slice).

/root/reference/renderers/mesh_rasterizer.py does ``import nvdiffrast.torch as dr`` (:2) and calls

    self.glctx = dr.RasterizeCudaContext()                                                          :34
    rast_out, _ = dr.rasterize(self.glctx, pos_clip, t_pos_idx, resolution=res, grad_db=False)      :103
    positions_all, _ = dr.interpolate(v_pos[None, ...], rast_out, t_pos_idx)                        :117  (:145, :153)
    alpha = dr.antialias(...)                                                                        :107, :128   NOT in this slice

This module offers the first three under the same names and argument order (``import tssplat_amd.dr as dr``).  nvdiffrast
is a separate library, not vendored by the reference and not installed here: the semantics are a restatement of its
published algorithm, pinned down in oracle/raster_oracle.py -- PARITY UNPINNED against the library itself.

What this slice does not do, loudly:
* ``antialias`` raises ``NotImplementedError`` -- and with it goes the silhouette gradient the reference's alpha loss
  lives on; this is a stand-alone operator pair with its own oracle and bench, not yet a renderer for trainer.py;
* ``rasterize`` is not differentiable: nvdiffrast propagates d(u, v) / d(pos) through it, here ``rast`` is returned
  detached (``interpolate`` still produces the gradient w.r.t. ``rast``'s (u, v), so nothing is lost silently downstream
  of this module -- it simply stops at ``rast``).  ``grad_db=True`` and ``ranges`` are rejected;
* no polygon clipping: a triangle with a vertex at ``w <= 0`` is dropped.
"""
from __future__ import annotations

import torch

from . import _capi
from .tet_spheres_ext import _device_ctx, _stream_ptr

__all__ = ["RasterizeCudaContext", "rasterize", "interpolate", "antialias"]

_lib = _capi.load()


class RasterizeCudaContext:
    """``dr.RasterizeCudaContext()`` (mesh_rasterizer.py:34): holds the workspace (depth keys, snapped vertices) between calls."""

    def __init__(self, device=None):
        self.device = None if device is None else torch.device(device)
        self._ws = None

    def workspace(self, batch: int, n_vertices: int, height: int, width: int, device: torch.device) -> torch.Tensor:
        need = int(_lib.tsamd_rasterize_workspace_bytes(batch, n_vertices, height, width))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(max(need, 8), dtype=torch.uint8, device=device)
        return self._ws


def _check_cuda_f32(name: str, t: torch.Tensor) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"tssplat_amd.dr: {name} must be a GPU tensor (there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"tssplat_amd.dr: {name} must be float32")
    return t.contiguous()


def _check_tri(tri: torch.Tensor, device) -> torch.Tensor:
    if not isinstance(tri, torch.Tensor) or tri.dim() != 2 or tri.shape[1] != 3:
        raise RuntimeError("tssplat_amd.dr: tri must be an [T, 3] tensor")
    if tri.dtype != torch.int32:
        raise RuntimeError("tssplat_amd.dr: tri must be int32 (as nvdiffrast requires)")
    if tri.device != device:
        raise RuntimeError("tssplat_amd.dr: tri must live on the same device as pos / rast")
    return tri.contiguous()


def rasterize(glctx: RasterizeCudaContext, pos: torch.Tensor, tri: torch.Tensor, resolution, ranges=None, grad_db: bool = True):
    """``(rast, rast_db)``: ``rast[B, H, W, 4] = (u, v, z/w, triangle_id + 1)``, zeros on background; ``rast_db`` is an empty
    tensor (image-space derivatives are only produced with ``grad_db=True``, which this slice rejects)."""
    if ranges is not None:
        raise NotImplementedError("tssplat_amd.dr.rasterize: range mode is not part of this slice")
    if grad_db:
        raise NotImplementedError("tssplat_amd.dr.rasterize: grad_db=True is not part of this slice (the reference passes grad_db=False)")
    pos = _check_cuda_f32("pos", pos.detach())
    if pos.dim() != 3 or pos.shape[2] != 4:
        raise RuntimeError("tssplat_amd.dr.rasterize: pos must be [B, V, 4] clip-space positions (instanced mode)")
    tri = _check_tri(tri, pos.device)
    height, width = int(resolution[0]), int(resolution[1])
    B, V = int(pos.shape[0]), int(pos.shape[1])
    rast = torch.empty((B, height, width, 4), dtype=torch.float32, device=pos.device)
    ws = glctx.workspace(B, V, height, width, pos.device)
    with _device_ctx(pos.device):
        _capi.check(_lib.tsamd_rasterize(pos.data_ptr(), B, V, tri.data_ptr(), int(tri.shape[0]), height, width, ws.data_ptr(),
                                         rast.data_ptr(), _stream_ptr(pos.device)))
    return rast, torch.empty((B, height, width, 0), dtype=torch.float32, device=pos.device)


class _InterpolateFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri):
        B, H, W = int(rast.shape[0]), int(rast.shape[1]), int(rast.shape[2])
        A, V, Cn = int(attr.shape[0]), int(attr.shape[1]), int(attr.shape[2])
        out = torch.empty((B, H, W, Cn), dtype=torch.float32, device=rast.device)
        with _device_ctx(rast.device):
            _capi.check(_lib.tsamd_interpolate(attr.data_ptr(), A, V, Cn, rast.data_ptr(), tri.data_ptr(), B, H, W, out.data_ptr(),
                                               _stream_ptr(rast.device)))
        ctx.save_for_backward(attr, rast, tri)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        attr, rast, tri = ctx.saved_tensors
        B, H, W = int(rast.shape[0]), int(rast.shape[1]), int(rast.shape[2])
        A, V, Cn = int(attr.shape[0]), int(attr.shape[1]), int(attr.shape[2])
        g = grad_out.contiguous()
        grad_attr = torch.empty_like(attr)
        grad_rast = torch.empty_like(rast) if ctx.needs_input_grad[1] else None
        with _device_ctx(rast.device):
            _capi.check(_lib.tsamd_interpolate_backward(attr.data_ptr(), A, V, Cn, rast.data_ptr(), tri.data_ptr(), B, H, W, g.data_ptr(),
                                                        grad_attr.data_ptr(), None if grad_rast is None else grad_rast.data_ptr(),
                                                        _stream_ptr(rast.device)))
        return grad_attr, grad_rast, None


def interpolate(attr: torch.Tensor, rast: torch.Tensor, tri: torch.Tensor, rast_db=None, diff_attrs=None):
    """``(out, out_da)``: ``out[B, H, W, C] = u a0 + v a1 + (1 - u - v) a2``, 0 on background; ``out_da`` is an empty tensor
    (attribute derivatives need ``rast_db``, not part of this slice)."""
    if rast_db is not None or diff_attrs is not None:
        raise NotImplementedError("tssplat_amd.dr.interpolate: rast_db / diff_attrs are not part of this slice")
    rast = _check_cuda_f32("rast", rast)
    attr = _check_cuda_f32("attr", attr)
    if rast.dim() != 4 or rast.shape[3] != 4:
        raise RuntimeError("tssplat_amd.dr.interpolate: rast must be [B, H, W, 4]")
    if attr.dim() != 3 or attr.shape[0] not in (1, rast.shape[0]):
        raise RuntimeError("tssplat_amd.dr.interpolate: attr must be [1 or B, V, C]")
    if attr.device != rast.device:
        raise RuntimeError("tssplat_amd.dr.interpolate: attr and rast must live on the same device")
    tri = _check_tri(tri, rast.device)
    out = _InterpolateFunc.apply(attr, rast, tri)
    return out, torch.empty(tuple(rast.shape[:3]) + (0,), dtype=torch.float32, device=rast.device)


def antialias(*args, **kwargs):
    raise NotImplementedError("tssplat_amd.dr.antialias is not part of this slice (SURVEY 8(f) row 4: rasterize + interpolate only); "
                              "the reference calls it at renderers/mesh_rasterizer.py:107,128")
