"""The nvdiffrast operators the reference's renderer uses, on the MI355X kernels.

/root/reference/renderers/mesh_rasterizer.py does ``import nvdiffrast.torch as dr`` (:2) and calls

    self.glctx = dr.RasterizeCudaContext()                                                          :34
    rast_out, _ = dr.rasterize(self.glctx, pos_clip, t_pos_idx, resolution=res, grad_db=False)      :103
    positions_all, _ = dr.interpolate(v_pos[None, ...], rast_out, t_pos_idx)                        :117  (:145, :153)
    alpha = dr.antialias(alpha, rast_out, pos_clip, t_pos_idx, topology_hash=None, pos_gradient_boost=1.0)   :107, :128

This module offers all four under the same names and argument order (``import tssplat_amd.dr as dr``), differentiable
where nvdiffrast is: ``rasterize`` w.r.t. ``pos`` through ``(u, v)``, ``interpolate`` w.r.t. ``attr`` and ``rast``,
``antialias`` w.r.t. ``color`` and ``pos``.  nvdiffrast is a separate library, not vendored by the reference and not
installed here: the semantics are a restatement of its published algorithm, pinned down in oracle/raster_oracle.py --
PARITY UNPINNED against the library itself.

What it does not do, loudly: ``grad_db=True``, ``ranges`` (range mode) and ``rast_db`` / ``diff_attrs`` are
rejected (``RasterizeGLContext`` is an alias of the HIP context); clipping is against the NEAR plane only (a triangle with vertices at ``w <= 0`` is clipped there; one with a vertex
beyond the +-16384-pixel guard band is dropped; the silhouette of a clipped triangle is not antialiased); no depth peeling,
no texture sampling.
"""
from __future__ import annotations

import weakref

import os

import torch

from . import _capi
from .tet_spheres_ext import _device_ctx, _stream_ptr

__all__ = ["RasterizeCudaContext", "rasterize", "interpolate", "antialias", "antialias_construct_topology_hash"]

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


class RasterizeGLContext(RasterizeCudaContext):
    """``dr.RasterizeGLContext()`` (mesh_rasterizer.py:35-36, ``context_type == "gl"``).  nvdiffrast's second back end runs the same
    operators through OpenGL; there is no OpenGL on this platform and nothing in the operators' contract depends on it, so the
    name is served by the same HIP kernels (``output_db`` / ``mode`` are accepted and ignored: image-space derivatives are not
    offered by either context here)."""

    def __init__(self, output_db: bool = True, mode: str = "automatic", device=None):
        super().__init__(device)


def _check_cuda_f32(name: str, t: torch.Tensor) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"tssplat_amd.dr: {name} must be a GPU tensor (there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"tssplat_amd.dr: {name} must be float32")
    return t.contiguous()


# TSSPLAT_AMD_DR_CHECK=1: count, on every rasterize call, the triangles this slice DROPS -- a vertex that is not finite or
# outside the +-16384-pixel guard band (a finite vertex at w <= 0 is clipped against the near plane, not dropped) -- and warn.
# A device read per call: a debugging aid, off by default (objects inside the frustum, the reference's case, lose nothing).
_CHECK_DROPPED = os.environ.get("TSSPLAT_AMD_DR_CHECK", "0") == "1"


def count_dropped_triangles(pos: torch.Tensor, tri: torch.Tensor, height: int, width: int) -> int:
    """(view, triangle) pairs that ``rasterize`` drops whole (see the module docstring)."""
    with torch.no_grad():
        w = pos[..., 3]
        finite = torch.isfinite(pos).all(dim=-1)
        front = finite & (w > 0)
        ws = torch.where(front, w, torch.ones_like(w))
        sx = (pos[..., 0] / ws * 0.5 + 0.5) * width
        sy = (pos[..., 1] / ws * 0.5 + 0.5) * height
        ok = (front & (sx.abs() <= 16384.0) & (sy.abs() <= 16384.0)) | (finite & (w <= 0))    # (w <= 0: clipped at the near plane)
        t = tri.long()
        bad = ~(ok[:, t[:, 0]] & ok[:, t[:, 1]] & ok[:, t[:, 2]])
        return int(bad.sum())


def _warn_dropped(pos, tri, height, width) -> None:
    n = count_dropped_triangles(pos, tri, height, width)
    if n:
        import warnings
        warnings.warn(f"tssplat_amd.dr.rasterize: {n} (view, triangle) pairs have a vertex that is not finite or beyond the +-16384-pixel "
                      f"guard band and are DROPPED whole", RuntimeWarning, stacklevel=3)


def _check_tri(tri: torch.Tensor, device) -> torch.Tensor:
    if not isinstance(tri, torch.Tensor) or tri.dim() != 2 or tri.shape[1] != 3:
        raise RuntimeError("tssplat_amd.dr: tri must be an [T, 3] tensor")
    if tri.dtype != torch.int32:
        raise RuntimeError("tssplat_amd.dr: tri must be int32 (as nvdiffrast requires)")
    if tri.device != device:
        raise RuntimeError("tssplat_amd.dr: tri must live on the same device as pos / rast")
    return tri.contiguous()


class _RasterizeFunc(torch.autograd.Function):
    """``rast`` with the gradient of its ``(u, v)`` channels w.r.t. ``pos`` (tsamd_rasterize_backward)."""

    @staticmethod
    def forward(ctx, pos, tri, glctx, height, width, pair_masks):
        B, V = int(pos.shape[0]), int(pos.shape[1])
        rast = torch.empty((B, height, width, 4), dtype=torch.float32, device=pos.device)
        ws = glctx.workspace(B, V, height, width, pos.device)
        with _device_ctx(pos.device):
            _capi.check(_lib.tsamd_rasterize(pos.data_ptr(), B, V, tri.data_ptr(), int(tri.shape[0]), height, width, ws.data_ptr(),
                                             rast.data_ptr(), None if pair_masks is None else pair_masks.data_ptr(), _stream_ptr(pos.device)))
        ctx.save_for_backward(pos, tri, rast)
        return rast

    @staticmethod
    def backward(ctx, grad_rast):
        pos, tri, rast = ctx.saved_tensors
        B, V, H, W = int(pos.shape[0]), int(pos.shape[1]), int(rast.shape[1]), int(rast.shape[2])
        g = grad_rast.contiguous()
        grad_pos = torch.empty_like(pos)
        with _device_ctx(pos.device):
            _capi.check(_lib.tsamd_rasterize_backward(pos.data_ptr(), B, V, tri.data_ptr(), int(tri.shape[0]), H, W, rast.data_ptr(),
                                                      g.data_ptr(), grad_pos.data_ptr(), _stream_ptr(pos.device)))
        return grad_pos, None, None, None, None, None


# The pair masks of the last rasterised image per device (a by-product of the resolve pass, 2 bits per pixel: which pixels show a
# different triangle than their right / upper neighbour), for the antialias call that receives the SAME rast tensor unmodified --
# identity through a weak reference plus the version counter, as for the topology table below.  mesh_rasterizer.py:103-107 is
# exactly that sequence; any other use (a copy, a slice, an edited image) finds no masks and antialias scans the image itself.
# The by-product costs the resolve pass a second stream of depth keys (+0.06 ms at 120 views x 512^2), so it is produced only
# while it is being used: a rasterize call that finds the previous masks unclaimed stops producing them, an antialias call that
# finds none asks for them again.
PAIR_MASKS_FROM_RASTERIZE = os.environ.get("TSSPLAT_AMD_DR_PAIR_MASKS", "1") != "0"
_last_pair_masks: dict = {}     # device -> (masks, weakref(rast), version)
_pair_masks_wanted: dict = {}   # device -> bool (absent: yes)


def _pair_masks_for(rast: torch.Tensor):
    hit = _last_pair_masks.get(rast.device)
    if hit is not None:
        masks, ref, version = hit
        if ref() is rast and version == rast._version:
            del _last_pair_masks[rast.device]          # claimed
            return masks
    _pair_masks_wanted[rast.device] = True             # an antialias call without masks: the next rasterize makes them
    return None


def rasterize(glctx: RasterizeCudaContext, pos: torch.Tensor, tri: torch.Tensor, resolution, ranges=None, grad_db: bool = True):
    """``(rast, rast_db)``: ``rast[B, H, W, 4] = (u, v, z/w, triangle_id + 1)``, zeros on background; ``rast_db`` is an empty
    tensor (image-space derivatives are only produced with ``grad_db=True``, which is rejected: the reference passes False)."""
    if ranges is not None:
        raise NotImplementedError("tssplat_amd.dr.rasterize: range mode is not supported")
    if grad_db:
        raise NotImplementedError("tssplat_amd.dr.rasterize: grad_db=True is not supported (the reference passes grad_db=False)")
    pos = _check_cuda_f32("pos", pos)
    if pos.dim() != 3 or pos.shape[2] != 4:
        raise RuntimeError("tssplat_amd.dr.rasterize: pos must be [B, V, 4] clip-space positions (instanced mode)")
    tri = _check_tri(tri, pos.device)
    height, width = int(resolution[0]), int(resolution[1])
    # limits of this slice (include/tssplat_amd.h): the triangle id + 1 travels as a float32 (exact up to 2^24), and window
    # coordinates are snapped within +-16384 pixels -- a triangle with a vertex beyond that guard band is dropped whole (one with
    # vertices at w <= 0 is clipped against the near plane): harmless for objects inside the frustum, which the reference renders
    if int(tri.shape[0]) > (1 << 24) - 1:
        raise RuntimeError("tssplat_amd.dr.rasterize: more than 2^24 - 1 triangles")
    if not (0 <= height <= 8192 and 0 <= width <= 8192):
        raise RuntimeError("tssplat_amd.dr.rasterize: resolution out of range (0 .. 8192 pixels per side)")
    if _CHECK_DROPPED and tri.shape[0] > 0:
        _warn_dropped(pos, tri, height, width)
    masks = None
    if _last_pair_masks.pop(pos.device, None) is not None:
        _pair_masks_wanted[pos.device] = False         # the last ones were never claimed by an antialias call
    if PAIR_MASKS_FROM_RASTERIZE and _pair_masks_wanted.get(pos.device, True) and height * width > 0:
        masks = torch.empty((int(_lib.tsamd_pair_masks_bytes(int(pos.shape[0]), height, width)),), dtype=torch.uint8, device=pos.device)
    rast = _RasterizeFunc.apply(pos, tri, glctx, height, width, masks)
    if masks is not None:
        _last_pair_masks[pos.device] = (masks, weakref.ref(rast), rast._version)
    return rast, torch.empty((int(pos.shape[0]), height, width, 0), dtype=torch.float32, device=pos.device)


class _InterpolateFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri):
        B, H, W = int(rast.shape[0]), int(rast.shape[1]), int(rast.shape[2])
        A, V, Cn = int(attr.shape[0]), int(attr.shape[1]), int(attr.shape[2])
        out = torch.empty((B, H, W, Cn), dtype=torch.float32, device=rast.device)
        with _device_ctx(rast.device):
            _capi.check(_lib.tsamd_interpolate(attr.data_ptr(), A, V, Cn, rast.data_ptr(), tri.data_ptr(), int(tri.shape[0]), B, H, W, out.data_ptr(),
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
            _capi.check(_lib.tsamd_interpolate_backward(attr.data_ptr(), A, V, Cn, rast.data_ptr(), tri.data_ptr(), int(tri.shape[0]), B, H, W, g.data_ptr(),
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


class TopologyHash:
    """``dr.antialias_construct_topology_hash(tri)``: the edge partner table of one triangle list (``opp[3 t + e]`` = the
    vertex across edge ``e`` of triangle ``t``, -1 on a boundary)."""

    def __init__(self, tri: torch.Tensor):
        tri = _check_tri(tri, tri.device)
        if not tri.is_cuda:
            raise RuntimeError("tssplat_amd.dr: tri must be a GPU tensor (there is no CPU fallback)")
        T = int(tri.shape[0])
        self.n_triangles = T
        self.opp = torch.empty(3 * T, dtype=torch.int32, device=tri.device)
        need = int(_lib.tsamd_antialias_topology_workspace_bytes(T))
        if need < 0:
            raise RuntimeError("tssplat_amd.dr: too many triangles for the topology table")
        ws = torch.empty(max(need, 8), dtype=torch.uint8, device=tri.device)
        with _device_ctx(tri.device):
            _capi.check(_lib.tsamd_antialias_topology(tri.data_ptr(), T, ws.data_ptr(), self.opp.data_ptr(), _stream_ptr(tri.device)))


def antialias_construct_topology_hash(tri: torch.Tensor) -> TopologyHash:
    return TopologyHash(tri)


_last_topology: dict = {}


def _topology_for(tri: torch.Tensor) -> TopologyHash:
    """``topology_hash=None``: nvdiffrast rebuilds the table on every call; the last one per device is kept here for as long as
    the SAME (alive, unmodified) index tensor comes back -- identity through a weak reference plus the version counter, not
    the address: a freed tensor whose storage a new one reuses can never be served the old table."""
    hit = _last_topology.get(tri.device)
    if hit is not None:
        topo, ref, version = hit
        if ref() is tri and version == tri._version:
            return topo
    topo = TopologyHash(tri)
    _last_topology[tri.device] = (topo, weakref.ref(tri), tri._version)
    return topo


# None: decide per call (see _AntialiasFunc.forward); True / False force the prepared / the per-pair form of the antialias analysis
PREPARE_ANTIALIAS = None


class _AntialiasFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, opp, boost, pair_masks):
        B, H, W, Cn = (int(k) for k in color.shape)
        V, T = int(pos.shape[1]), int(tri.shape[0])
        out = torch.empty_like(color)
        # window coordinates per (view, vertex), mask of the pixel pairs on two triangles, edge flags per (view, triangle): once
        # for the forward and the backward analysis -- when the mesh is small against the image (the reference's use: one object,
        # 120 views x 512^2).  With more triangles than pixels the per-vertex / per-triangle tables cost more than they save
        # (512 spheres in 8 views: 0.32 against 0.16 ms) and the kernels work everything out per pixel pair.
        prepare = PREPARE_ANTIALIAS if PREPARE_ANTIALIAS is not None else 4 * (V + T) <= H * W
        win = torch.empty((int(_lib.tsamd_antialias_prepared_bytes(B, V, T, H, W)) if prepare else 0,), dtype=torch.uint8, device=color.device)
        with _device_ctx(color.device):
            stream = _stream_ptr(color.device)
            if prepare:
                _capi.check(_lib.tsamd_antialias_prepare(rast.data_ptr(), pos.data_ptr(), tri.data_ptr(), opp.data_ptr(),
                                                         None if pair_masks is None else pair_masks.data_ptr(), B, V, T, H, W, win.data_ptr(), stream))
            _capi.check(_lib.tsamd_antialias(color.data_ptr(), rast.data_ptr(), pos.data_ptr(), win.data_ptr() if prepare else None, tri.data_ptr(),
                                             opp.data_ptr(), B, V, T, H, W, Cn, out.data_ptr(), stream))
        ctx.save_for_backward(color, rast, pos, tri, opp, win)
        ctx.boost = float(boost)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        color, rast, pos, tri, opp, win = ctx.saved_tensors
        B, H, W, Cn = (int(k) for k in color.shape)
        V, T = int(pos.shape[1]), int(tri.shape[0])
        g = grad_out.contiguous()
        grad_color = torch.empty_like(color) if ctx.needs_input_grad[0] else None
        grad_pos = torch.empty_like(pos) if ctx.needs_input_grad[2] else None
        if grad_color is None and grad_pos is None:
            return None, None, None, None, None, None, None
        with _device_ctx(color.device):
            _capi.check(_lib.tsamd_antialias_backward(color.data_ptr(), rast.data_ptr(), pos.data_ptr(), win.data_ptr() if win.numel() else None,
                                                      tri.data_ptr(), opp.data_ptr(),
                                                      B, V, T, H, W, Cn, g.data_ptr(), ctx.boost, None if grad_color is None else grad_color.data_ptr(),
                                                      None if grad_pos is None else grad_pos.data_ptr(), _stream_ptr(color.device)))
        return grad_color, None, grad_pos, None, None, None, None


def antialias(color: torch.Tensor, rast: torch.Tensor, pos: torch.Tensor, tri: torch.Tensor, topology_hash=None, pos_gradient_boost: float = 1.0):
    """``dr.antialias`` (mesh_rasterizer.py:107,128): ``color[B, H, W, C]`` with the silhouette pixels blended by the position
    of the silhouette edge between the pixel centres; differentiable w.r.t. ``color`` and ``pos`` (``rast`` carries none)."""
    color = _check_cuda_f32("color", color)
    pair_masks = _pair_masks_for(rast)              # (by identity of the tensor dr.rasterize handed out)
    rast = _check_cuda_f32("rast", rast.detach())
    pos = _check_cuda_f32("pos", pos)
    if color.dim() != 4 or rast.dim() != 4 or rast.shape[3] != 4 or tuple(color.shape[:3]) != tuple(rast.shape[:3]):
        raise RuntimeError("tssplat_amd.dr.antialias: color must be [B, H, W, C] and rast [B, H, W, 4] of the same image size")
    if pos.dim() != 3 or pos.shape[2] != 4 or pos.shape[0] != rast.shape[0]:
        raise RuntimeError("tssplat_amd.dr.antialias: pos must be [B, V, 4] clip-space positions (instanced mode)")
    if color.device != rast.device or pos.device != rast.device:
        raise RuntimeError("tssplat_amd.dr.antialias: color, rast and pos must live on the same device")
    tri = _check_tri(tri, rast.device)
    topo = _topology_for(tri) if topology_hash is None else topology_hash
    if not isinstance(topo, TopologyHash) or topo.n_triangles != int(tri.shape[0]) or topo.opp.device != rast.device:
        raise RuntimeError("tssplat_amd.dr.antialias: topology_hash does not belong to this triangle list")
    return _AntialiasFunc.apply(color, rast, pos, tri, topo.opp, float(pos_gradient_boost), pair_masks)
