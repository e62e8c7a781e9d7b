"""Surface glue of the geometry module on MI355X.

Mirror of what /root/reference/geometry/tetmesh_geometry.py does with the boundary triangles around the
energy in every iteration:

* ``TetMeshGeometryForwardData`` (:27-66) -- ``v_pos = tet_v[surface_vid]`` (:33) and
  ``_compute_vertex_normal()`` (:39-66), same constructor and attribute names;
* ``permute_surface_v`` (``TetMeshGeometry.forward``, :176-182) -- uniform noise on the surface vertices;
* ``get_surface_vf`` (geometry/mesh_utils.py:5-35) -- the one-off boundary extraction (host).

Both differentiable steps run as per-vertex gather kernels behind the C ABI (``tsamd_surface_positions``,
``tsamd_vertex_normals`` and their ``_backward`` twins in include/tssplat_amd.h) instead of the reference's
index / cross / three ``scatter_add_`` / where / normalize chain with atomics: deterministic, one launch
forward, two backward.  There is no CPU fallback: without the HIP library these calls raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import weakref

import torch

from . import _capi


def get_surface_vf(elem: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Boundary of a tet mesh as (surface vertex ids, triangles over those vertices), same order and
    orientation as geometry/mesh_utils.py:5-35.  Host code (runs once per mesh)."""
    lib = _capi.load()
    t = np.ascontiguousarray(np.asarray(elem).reshape(-1, 4), dtype=np.int32)
    n = int(t.max()) + 1 if t.size else 0
    ns, nf = C.c_int64(0), C.c_int64(0)
    _capi.check(lib.tsamd_extract_surface(t.ctypes.data, t.shape[0], n, None, C.byref(ns), None, C.byref(nf)))
    vid = np.empty(ns.value, dtype=np.int32)
    faces = np.empty((nf.value, 3), dtype=np.int32)
    _capi.check(lib.tsamd_extract_surface(t.ctypes.data, t.shape[0], n, vid.ctypes.data, C.byref(ns),
                                          faces.ctypes.data, C.byref(nf)))
    return vid, faces


class SurfaceOps:
    """Device copy of the surface topology (``surface_vid``, ``surface_fid`` of tetmesh_geometry.py:146-149)."""

    def __init__(self, surface_vid, surface_f, n_tet_vertices: int, device: Optional[torch.device] = None):
        lib = _capi.load()
        vid = np.ascontiguousarray(_to_numpy(surface_vid).reshape(-1), dtype=np.int32)
        faces = np.ascontiguousarray(_to_numpy(surface_f).reshape(-1, 3), dtype=np.int32)
        if device is None:
            device = surface_vid.device if isinstance(surface_vid, torch.Tensor) and surface_vid.is_cuda else torch.device("cuda")
        self.device = torch.device(device)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self.n_surface_vertices, self.n_faces, self.n_tet_vertices = int(vid.size), int(faces.shape[0]), int(n_tet_vertices)
        h = C.c_void_p()
        _capi.check(lib.tsamd_surface_create(vid.ctypes.data, vid.size, faces.ctypes.data, faces.shape[0],
                                             int(n_tet_vertices), index, C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _capi.load().tsamd_surface_destroy(h)
            except Exception:
                pass

    def _check(self, t: torch.Tensor, rows: int, what: str) -> torch.Tensor:
        if not t.is_cuda or t.device != self.device:
            raise RuntimeError(f"{what} must live on {self.device}, got {t.device}")
        if t.dtype != torch.float32 or t.dim() != 2 or tuple(t.shape) != (rows, 3):
            raise RuntimeError(f"{what} must be float32 [{rows}, 3], got {t.dtype} {tuple(t.shape)}")
        return t.contiguous()

    def positions(self, tet_v: torch.Tensor) -> torch.Tensor:
        return _SurfacePositions.apply(tet_v, self)

    def vertex_normals(self, v_pos: torch.Tensor) -> torch.Tensor:
        return _VertexNormals.apply(v_pos, self)


def _to_numpy(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _SurfacePositions(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tet_v, ops: SurfaceOps):
        tv = ops._check(tet_v, ops.n_tet_vertices, "tet_v")
        out = torch.empty((ops.n_surface_vertices, 3), dtype=torch.float32, device=ops.device)
        with torch.cuda.device(ops.device):
            _capi.check(_capi.load().tsamd_surface_positions(ops._h, tv.data_ptr(), _stream(ops.device), out.data_ptr()))
        ctx.ops = ops
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ops = ctx.ops
        g = ops._check(grad_out, ops.n_surface_vertices, "grad of v_pos")
        out = torch.empty((ops.n_tet_vertices, 3), dtype=torch.float32, device=ops.device)
        with torch.cuda.device(ops.device):
            _capi.check(_capi.load().tsamd_surface_positions_backward(ops._h, g.data_ptr(), _stream(ops.device), out.data_ptr()))
        return out, None


class _VertexNormals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v_pos, ops: SurfaceOps):
        vp = ops._check(v_pos, ops.n_surface_vertices, "v_pos")
        nrm = torch.empty_like(vp)
        raw = torch.empty_like(vp)
        with torch.cuda.device(ops.device):
            _capi.check(_capi.load().tsamd_vertex_normals(ops._h, vp.data_ptr(), _stream(ops.device), nrm.data_ptr(), raw.data_ptr()))
        ctx.ops = ops
        ctx.save_for_backward(vp, raw)
        return nrm

    @staticmethod
    def backward(ctx, grad_out):
        ops = ctx.ops
        vp, raw = ctx.saved_tensors
        g = ops._check(grad_out, ops.n_surface_vertices, "grad of v_nrm")
        out = torch.empty_like(vp)
        ws = torch.empty_like(vp)
        with torch.cuda.device(ops.device):
            _capi.check(_capi.load().tsamd_vertex_normals_backward(ops._h, vp.data_ptr(), raw.data_ptr(), g.data_ptr(),
                                                                   ws.data_ptr(), _stream(ops.device), out.data_ptr()))
        return out, None


_OPS_CACHE: dict = {}


def _ops_for(surface_vid: torch.Tensor, surface_f: torch.Tensor, n_tet_vertices: int) -> SurfaceOps:
    """The reference rebuilds TetMeshGeometryForwardData every iteration from the same index tensors
    (tetmesh_geometry.py:190-192); the topology handle is cached per pair of index tensors so that only the first
    iteration pays for the host-side incidence build.  An entry is valid only while BOTH tensors it was built from
    are alive and unmodified: it holds weak references to them and checks identity, version and device, so a freed
    tensor whose address a new one happens to reuse (TetMeshGeometry.reset()) can never be served the old topology."""
    key = (surface_vid.data_ptr(), surface_f.data_ptr(), str(surface_vid.device), tuple(surface_vid.shape),
           tuple(surface_f.shape), int(n_tet_vertices))
    hit = _OPS_CACHE.get(key)
    if hit is not None:
        ops, ref_v, ref_f, ver = hit
        if ref_v() is surface_vid and ref_f() is surface_f and ver == (surface_vid._version, surface_f._version):
            return ops
        del _OPS_CACHE[key]
    for k in [k for k, (_, rv, rf, _) in _OPS_CACHE.items() if rv() is None or rf() is None]:
        del _OPS_CACHE[k]                                  # entries whose tensors are gone release their device memory
    if len(_OPS_CACHE) > 8:
        _OPS_CACHE.clear()
    ops = SurfaceOps(surface_vid, surface_f, n_tet_vertices)
    _OPS_CACHE[key] = (ops, weakref.ref(surface_vid), weakref.ref(surface_f), (surface_vid._version, surface_f._version))
    return ops


class TetMeshGeometryForwardData:
    """Drop-in for geometry/tetmesh_geometry.py:27-66 (constructor signature, ``tet_v``, ``tet_elem``,
    ``v_pos``, ``t_pos_idx``, ``smooth_barrier_energy``, ``_compute_vertex_normal()``)."""

    def __init__(self, tet_v: torch.Tensor, tet_elem: torch.Tensor, surface_vid: torch.Tensor, surface_f: torch.Tensor,
                 smooth_barrier_energy=None, surface_ops: Optional[SurfaceOps] = None):
        self.tet_v = tet_v
        self.tet_elem = tet_elem
        self._ops = surface_ops if surface_ops is not None else _ops_for(surface_vid, surface_f, tet_v.shape[0])
        # surface
        self.v_pos = self._ops.positions(tet_v)                    # tetmesh_geometry.py:33
        self.t_pos_idx = surface_f
        # geometry regularization
        self.smooth_barrier_energy = smooth_barrier_energy

    def _compute_vertex_normal(self) -> torch.Tensor:
        return self._ops.vertex_normals(self.v_pos)                # tetmesh_geometry.py:39-66


def permute_surface_v(tet_v: torch.Tensor, surface_vid: torch.Tensor, dev: float,
                      generator: Optional[torch.Generator] = None) -> None:
    """``tet_v[surface_vid] += U(0,1) * dev - dev / 2`` in place, no gradient (tetmesh_geometry.py:176-182).
    Host logic over torch's device RNG, like the reference: not a hot path (runs on permutation events only)."""
    with torch.no_grad():
        idx = surface_vid.long()
        noise = torch.rand((idx.shape[0], tet_v.shape[1]), device=tet_v.device, dtype=tet_v.dtype, generator=generator)
        tet_v[idx] += noise * dev - dev * 0.5


class TetMeshGeometry(torch.nn.Module):
    """Mirror of ``TetMeshGeometry`` (geometry/tetmesh_geometry.py:118-193) for the per-iteration path: the parameter ``tet_v``,
    the index buffers, ``mesh_smooth_barrier`` and ``forward(iter_num, **kwargs) -> TetMeshGeometryForwardData`` with the
    reference's surface permutation (:176-182) and energy schedule (:184-189).

    The reference reads its configuration with omegaconf and its mesh with pypgo (``TetrahedronMesh(veg_file_path=...)``);
    neither is part of the hot path, so the constructor takes the arrays (``from_veg`` reads a ``.veg`` file with this
    package's own reader) and the reference's ``Config`` fields as keyword arguments.  Remeshing / export / uv are not
    mirrored (offline tools)."""

    def __init__(self, vtx_init, elem, use_smooth_barrier: bool = True, smooth_barrier_param=None, optimize_geo: bool = True,
                 device=None, surface_vid=None, surface_fid=None, **energy_kwargs):
        super().__init__()
        self.device = torch.device("cuda" if device is None else device)
        if self.device.type != "cuda":
            raise RuntimeError("tssplat_amd.geometry.TetMeshGeometry needs a GPU device (there is no CPU fallback)")
        vtx = np.ascontiguousarray(np.asarray(vtx_init, dtype=np.float32).reshape(-1, 3))
        elem = np.ascontiguousarray(np.asarray(elem, dtype=np.int32).reshape(-1, 4))
        if surface_vid is None or surface_fid is None:
            surface_vid, surface_fid = get_surface_vf(elem)            # geometry/tetrahedron_mesh.py: the one-off boundary extraction
        tet_v = torch.from_numpy(vtx).to(self.device)
        if optimize_geo:                                                # tetmesh_geometry.py:138-141
            self.tet_v = torch.nn.Parameter(tet_v, requires_grad=True)
        else:
            self.register_buffer("tet_v", tet_v)
        self.tet_elem = torch.from_numpy(elem).to(self.device)
        self.surface_vid = torch.from_numpy(np.ascontiguousarray(np.asarray(surface_vid, dtype=np.int32))).to(self.device)
        self.surface_fid = torch.from_numpy(np.ascontiguousarray(np.asarray(surface_fid, dtype=np.int32).reshape(-1, 3))).to(self.device)
        self.use_smooth_barrier = bool(use_smooth_barrier)
        self.mesh_smooth_barrier = None
        if self.use_smooth_barrier:                                     # tetmesh_geometry.py:156-162
            from .energies import SmoothnessBarrierEnergy
            for key in ("smooth_eng_coeff", "barrier_coeff", "increase_order_iter"):
                if not hasattr(smooth_barrier_param, key) and not (isinstance(smooth_barrier_param, dict) and key in smooth_barrier_param):
                    raise ValueError(f"smooth_barrier_param needs {key!r} (config/gso.yaml:8-11)")
            flags = smooth_barrier_param
            if isinstance(flags, dict):
                import types
                flags = types.SimpleNamespace(**flags)
            with torch.cuda.device(self.device):
                self.mesh_smooth_barrier = SmoothnessBarrierEnergy(vtx, elem, flags, **energy_kwargs)
        self._surface_ops = _ops_for(self.surface_vid, self.surface_fid, int(tet_v.shape[0]))

    @classmethod
    def from_veg(cls, path, **kwargs):
        from . import scenes
        v, t = scenes.read_veg(path)
        return cls(v, t, **kwargs)

    def forward(self, iter_num, **kwargs):
        if "permute_surface_v" in kwargs:                               # tetmesh_geometry.py:176-182
            assert "permute_surface_v_dev" in kwargs
            permute_surface_v(self.tet_v.data if isinstance(self.tet_v, torch.nn.Parameter) else self.tet_v, self.surface_vid,
                              float(kwargs["permute_surface_v_dev"]))
        smooth_barrier_energy = None
        if self.use_smooth_barrier:                                     # tetmesh_geometry.py:184-189
            smooth_coeff, barrier_coeff = self.mesh_smooth_barrier.coeff_scheduler(iter_num)
            smooth_barrier_energy = self.mesh_smooth_barrier(self.tet_v, iter_num, smooth_coeff, barrier_coeff)
        return TetMeshGeometryForwardData(self.tet_v, self.tet_elem, self.surface_vid, self.surface_fid, smooth_barrier_energy,
                                          surface_ops=self._surface_ops)
