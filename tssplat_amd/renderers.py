"""Mirror of ``MeshRasterizer`` (/root/reference/renderers/mesh_rasterizer.py:20-166) over this package's operators -- the
caller of SURVEY 8(f) rows 2 and 4: geometry forward (energy + surface gather), ``transform_pos``, ``dr.rasterize``,
``dr.antialias`` of the alpha image, and the optional colour / normal / depth branches through ``dr.interpolate``.

Same ``forward`` arguments and output keys (``"shaded"``, ``"geo_regularization"``, ``"n"``, ``"d"``) as the reference, so that
trainer.py:81-130 reads the same against either.  Host logic only: every tensor operation that is not one of this package's
kernels is the torch call the reference itself makes (clamp, lerp, masked assignment, norm).  Not mirrored: ``export``
(xatlas / pymeshlab / cv2, an offline tool) and the reference's structured-config plumbing (``context_type="gl"`` is served
by the same HIP kernels as ``"cuda"``; the two
``Config`` fields are keyword arguments).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import dr

__all__ = ["MeshRasterizer"]


class MeshRasterizer(torch.nn.Module):
    def __init__(self, geometry: torch.nn.Module, materials: Optional[torch.nn.Module] = None, context_type: str = "cuda", is_orhto: bool = False):
        super().__init__()
        if context_type not in ("cuda", "gl"):                       # mesh_rasterizer.py:33-38
            raise ValueError("Invalid context type")
        self.is_orhto = bool(is_orhto)                               # (the reference's spelling, mesh_rasterizer.py:24)
        self.glctx = dr.RasterizeCudaContext() if context_type == "cuda" else dr.RasterizeGLContext()   # (both are the HIP kernels)
        self.geometry = geometry
        self.materials = materials
        self.device = geometry.device
        n_surface = int(self.geometry.surface_vid.shape[0])
        self.ones_surface_v = torch.ones([n_surface, 1], device=self.device)      # mesh_rasterizer.py:49-53
        self.zeros_surface_v = torch.zeros([n_surface, 1], device=self.device)
        self.tri_hash = None                                         # mesh_rasterizer.py:55: the topology is rebuilt (here: cached) per call

    def transform_pos(self, mtx, pos, is_vec: bool = False):
        """``[pos, 1 | 0] @ mtx^T`` per view (mesh_rasterizer.py:57-78)."""
        t_mtx = torch.from_numpy(mtx).to(self.device) if isinstance(mtx, np.ndarray) else mtx.to(self.device)
        posw = torch.cat([pos, self.zeros_surface_v if is_vec else self.ones_surface_v], dim=1)
        res = torch.matmul(posw, t_mtx.transpose(1, 2))
        if not is_vec and self.is_orhto:
            res = res.clone()
            res[..., 2] /= 6
        return res

    def forward(self, mvp: torch.Tensor, only_alpha: bool, iter_num: int, resolution: int, permute_surface_scheduler=None,
                fit_normal: bool = False, fit_depth: bool = False, background: Optional[torch.Tensor] = None,
                campos: Optional[torch.Tensor] = None):
        geo_input = {"iter_num": iter_num}
        if permute_surface_scheduler is not None:                    # mesh_rasterizer.py:90-94
            permute_dev = permute_surface_scheduler(iter_num)
            if permute_dev is not None:
                geo_input["permute_surface_v"] = True
                geo_input["permute_surface_v_dev"] = permute_dev
        data = self.geometry(**geo_input)
        res = [resolution, resolution]
        tri = data.t_pos_idx

        pos_clip = self.transform_pos(mvp, data.v_pos).contiguous()
        rast_out, _ = dr.rasterize(self.glctx, pos_clip, tri, resolution=res, grad_db=False)
        # mesh_rasterizer.py:106-108.  The id channel carries no gradient (a triangle id + 1 is >= 1: the clamp is flat there, and
        # rasterize's backward ignores that channel anyway), so it is detached here: autograd would otherwise run the clamp's and
        # the slice's backward and a rasterize backward over all B x H x W pixels to deliver zeros -- 0.3 ms of 2 ms at 120 views.
        alpha = torch.clamp(rast_out[..., -1:].detach(), 0, 1)
        alpha = dr.antialias(alpha.contiguous(), rast_out, pos_clip, tri, topology_hash=self.tri_hash, pos_gradient_boost=1.0)

        shaded = alpha
        if not only_alpha:                                           # mesh_rasterizer.py:111-133
            assert self.materials is not None
            assert background is not None
            mask = rast_out[..., -1:] > 0
            selector = mask[..., 0]
            positions_all, _ = dr.interpolate(data.v_pos[None, ...], rast_out, tri)
            color = self.materials(positions=positions_all[selector])["color"]
            gb_fg = torch.zeros(rast_out.shape[0], res[0], res[0], 3, device=self.device)
            gb_fg[selector] = color
            gb_mat = torch.lerp(background, gb_fg, mask.float())
            shaded = dr.antialias(gb_mat.contiguous(), rast_out, pos_clip, tri, topology_hash=self.tri_hash, pos_gradient_boost=1.0)

        out = {"shaded": shaded, "geo_regularization": data.smooth_barrier_energy}

        if fit_normal:                                               # mesh_rasterizer.py:138-148
            v_s = data._compute_vertex_normal()[None, ...]
            scale = torch.tensor([1, 1, -1], dtype=torch.float32, device=self.device)[None, None, :]
            v_n, _ = dr.interpolate((v_s * scale).contiguous(), rast_out, tri)
            out["n"] = v_n

        if fit_depth:                                                # mesh_rasterizer.py:150-161
            assert campos is not None
            world_pos, _ = dr.interpolate(data.v_pos[None, ...], rast_out, tri)
            out["d"] = torch.norm(world_pos - campos[:, None, None, :], dim=-1, keepdim=True)
        return out
