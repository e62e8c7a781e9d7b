"""BASELINE config 5 / north_star "so trainer.py drops in unchanged": the reference's OWN geometry and renderer modules,
loaded UNMODIFIED from /root/reference (authoring container only -- the files are never copied), on top of this repo's
shims: `pypgo` -> the top-level stand-in, `tet_spheres` -> tssplat_amd.tet_spheres_ext, `nvdiffrast.torch` -> a RECORDING
stand-in with tssplat_amd.dr's names.  Third-party modules the two files import but never touch on this path (cv2, trimesh,
pymeshlab, xatlas, omegaconf; `materials`, which pulls tinycudann) are stubbed in sys.modules.

What is checked (CPU: there is no GPU here and /root/reference is not on the GPU box):
  * reference `TetrahedronMesh` + `TetMeshGeometry` build from a .veg file through the pypgo shim; the boundary the
    reference's `get_surface_vf` extracts equals what tssplat_amd.geometry.get_surface_vf (C++) returns, bit for bit;
  * reference `TetMeshGeometryForwardData` (surface gather + vertex normals) equals this repo's surface oracle;
  * reference `MeshRasterizer.forward` runs end to end against the recording `dr`: every call it makes binds to the
    signature of the same-named function of tssplat_amd.dr (names, positional order, keywords such as grad_db=False,
    topology_hash=, pos_gradient_boost=), and this repo's mirror `tssplat_amd.renderers.MeshRasterizer`, driven by the SAME
    reference geometry object, makes the same sequence of calls with the same shapes and returns the same output keys;
  * `transform_pos` of reference and mirror agree exactly;
  * VALUES: with a `dr` stand-in that answers every call with oracle/raster_oracle.py's images, the reference's forward and the
    mirror's return identical tensors (alpha-only and shaded paths, normals, depth) for the same geometry, cameras and material.
The device-side comparison of the mirrors with the kernels' oracles is tests/test_renderer_pipeline.py / test_raster.py (GPU).
"""
import importlib
import inspect
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "renderers", "mesh_rasterizer.py")),
                                reason="the reference checkout only exists in the authoring container")


class _Recorder(types.ModuleType):
    """`nvdiffrast.torch` stand-in: records every call and returns tensors of the shapes nvdiffrast documents."""

    def __init__(self):
        super().__init__("nvdiffrast.torch")
        self.calls = []

    class RasterizeCudaContext:
        def __init__(self, device=None):
            pass

    class RasterizeGLContext(RasterizeCudaContext):
        def __init__(self, output_db=True, mode="automatic", device=None):
            pass

    @staticmethod
    def _sig(args, kwargs):
        def one(a):
            if isinstance(a, torch.Tensor):
                return ("tensor", tuple(a.shape), str(a.dtype))
            if isinstance(a, (list, tuple)):
                return tuple(a)
            return a if isinstance(a, (int, float, bool, type(None))) else type(a).__name__
        return tuple(one(a) for a in args), tuple(sorted((k, one(v)) for k, v in kwargs.items()))

    def rasterize(self, glctx, pos, tri, resolution, *args, **kwargs):
        self.calls.append(("rasterize", (glctx, pos, tri, resolution) + args, kwargs))
        b, (h, w) = pos.shape[0], resolution
        out = torch.zeros(b, h, w, 4)
        out[:, h // 4:3 * h // 4, w // 4:3 * w // 4, 3] = 1.0          # a block of "triangle 0"
        return out, torch.zeros(b, h, w, 4)

    def interpolate(self, attr, rast, tri, *args, **kwargs):
        self.calls.append(("interpolate", (attr, rast, tri) + args, kwargs))
        return torch.zeros(*rast.shape[:3], attr.shape[-1]), None

    def antialias(self, color, rast, pos, tri, *args, **kwargs):
        self.calls.append(("antialias", (color, rast, pos, tri) + args, kwargs))
        return color

    def summary(self):
        return [(name,) + self._sig(a[1:] if name == "rasterize" else a, k) for name, a, k in self.calls]   # (the context object aside)


class _OracleDr(_Recorder):
    """The same stand-in computing REAL images: every call is answered by oracle/raster_oracle.py (numpy, forward only), so that the
    reference's forward and the mirror's can be compared on values, not just on the calls they make."""

    def rasterize(self, glctx, pos, tri, resolution, *args, **kwargs):
        from oracle import raster_oracle as RO
        self.calls.append(("rasterize", (glctx, pos, tri, resolution) + args, kwargs))
        rast = RO.rasterize(pos.detach().numpy(), tri.detach().numpy(), resolution)
        return torch.from_numpy(rast.astype(np.float32)), torch.zeros(rast.shape, dtype=torch.float32)

    def interpolate(self, attr, rast, tri, *args, **kwargs):
        from oracle import raster_oracle as RO
        self.calls.append(("interpolate", (attr, rast, tri) + args, kwargs))
        return torch.from_numpy(RO.interpolate(attr.detach().numpy(), rast.detach().numpy(), tri.detach().numpy()).astype(np.float32)), None

    def antialias(self, color, rast, pos, tri, *args, **kwargs):
        from oracle import raster_oracle as RO
        self.calls.append(("antialias", (color, rast, pos, tri) + args, kwargs))
        return torch.from_numpy(RO.antialias(color.detach().numpy(), rast.detach().numpy(), pos.detach().numpy(), tri.detach().numpy()).astype(np.float32))


@pytest.fixture()
def reference_modules(monkeypatch, tmp_path, request):
    """The reference's packages importable by their own names, third parties stubbed; everything is undone afterwards."""
    rec = getattr(request, "param", _Recorder)()
    stubs = {"nvdiffrast": types.ModuleType("nvdiffrast"), "nvdiffrast.torch": rec}
    stubs["nvdiffrast"].torch = rec
    for name in ("cv2", "trimesh", "pymeshlab"):
        stubs[name] = types.ModuleType(name)
    xatlas = types.ModuleType("xatlas")
    xatlas.parametrize = lambda v, f: (np.arange(len(v)), np.asarray(f), np.zeros((len(v), 2), np.float32))   # (export-only data)
    stubs["xatlas"] = xatlas
    omega = types.ModuleType("omegaconf")
    omega.DictConfig = dict

    class OmegaConf:
        @staticmethod
        def structured(obj):
            return obj
    omega.OmegaConf, omega.open_dict = OmegaConf, (lambda cfg: cfg)
    stubs["omegaconf"] = omega
    materials = types.ModuleType("materials")
    materials.ExplicitMaterial = type("ExplicitMaterial", (torch.nn.Module,), {})
    stubs["materials"] = materials
    for name, mod in stubs.items():
        monkeypatch.setitem(sys.modules, name, mod)
    for name in [n for n in sys.modules if n.split(".")[0] in ("geometry", "renderers", "utils", "energies")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(REF)
    cfg = importlib.import_module("utils.config")
    monkeypatch.setattr(cfg, "get_device", lambda: torch.device("cpu"))          # (the reference asks for cuda:<rank>)
    geo = importlib.import_module("geometry.tetmesh_geometry")
    ren = importlib.import_module("renderers.mesh_rasterizer")
    for m in (geo, ren):                                                         # they did `from utils.config import get_device`
        monkeypatch.setattr(m, "get_device", lambda: torch.device("cpu"))
    yield geo, ren, rec
    for name in [n for n in sys.modules if n.split(".")[0] in ("geometry", "renderers", "utils", "energies")]:
        del sys.modules[name]


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _veg(tmp_path):
    from tssplat_amd import scenes
    v, t = scenes.kuhn_ball(3)
    path = tmp_path / "ball.veg"
    scenes.write_veg(path, v, t)
    return str(path), v, t


def test_reference_geometry_runs_unmodified_on_the_shims(reference_modules, tmp_path):
    geo, _, _ = reference_modules
    assert geo.__file__.startswith(REF) and sys.modules["pypgo"].__file__.startswith("/root/repo")
    path, v, t = _veg(tmp_path)
    g = geo.TetMeshGeometry(_Cfg(use_smooth_barrier=False, initial_mesh_path=path, smooth_barrier_param=None, optimize_geo=True))
    assert np.array_equal(g.tetmesh.elem, t) and np.allclose(g.tetmesh.vtx_init, v)
    # the boundary: reference get_surface_vf (python) == this repo's C++ extraction, bit for bit
    from tssplat_amd.geometry import get_surface_vf
    vid, fid = get_surface_vf(t)
    assert np.array_equal(g.tetmesh.surface_vid, vid) and np.array_equal(g.tetmesh.surface_fid, fid)
    data = g(iter_num=0)
    assert isinstance(data, geo.TetMeshGeometryForwardData) and data.smooth_barrier_energy is None
    from oracle import surface_oracle as SO
    x = g.tet_v.detach().numpy().astype(np.float64)
    assert np.array_equal(data.v_pos.detach().numpy(), g.tet_v.detach().numpy()[vid])
    n_ref = data._compute_vertex_normal().detach().numpy()
    n_orc = SO.vertex_normals(x[vid], fid) if hasattr(SO, "vertex_normals") else None
    if n_orc is not None:
        assert np.abs(n_ref - n_orc).max() <= 2e-6


def test_reference_rasterizer_calls_bind_to_tssplat_amd_dr_and_match_the_mirror(reference_modules, tmp_path):
    geo, ren, rec = reference_modules
    path, v, t = _veg(tmp_path)
    g = geo.TetMeshGeometry(_Cfg(use_smooth_barrier=False, initial_mesh_path=path, smooth_barrier_param=None, optimize_geo=True))
    ref = ren.MeshRasterizer(g, None, _Cfg(context_type="cuda", is_orhto=False))
    from tssplat_amd import scenes
    mvp = torch.from_numpy(scenes.orbit_mvps(3))
    campos = torch.zeros(3, 3)
    out_ref = ref(mvp, only_alpha=True, iter_num=5, resolution=16, fit_normal=True, fit_depth=True, campos=campos)
    calls_ref = rec.summary()
    assert [c[0] for c in calls_ref] == ["rasterize", "antialias", "interpolate", "interpolate"]      # mesh_rasterizer.py:103,107,145,153
    # every call of the reference binds to the same-named function of tssplat_amd.dr
    # (importing tssplat_amd.dr itself loads the HIP library: fine on CPU, it is only inspected here)
    import tssplat_amd.dr as our_dr
    for name, args, kwargs in rec.calls:
        bound = inspect.signature(getattr(our_dr, name)).bind(*args, **kwargs)
        if name == "rasterize":
            assert bound.arguments["grad_db"] is False and bound.arguments.get("ranges") is None
        if name == "antialias":
            assert bound.arguments["topology_hash"] is None and bound.arguments["pos_gradient_boost"] == 1.0
    for ctx in ("RasterizeCudaContext", "RasterizeGLContext"):
        assert hasattr(our_dr, ctx)
    # the mirror, driven by the SAME reference geometry object and the same recording dr
    import tssplat_amd.renderers as mirror_mod
    rec.calls.clear()
    real_dr = mirror_mod.dr
    mirror_mod.dr = rec
    try:
        mir = mirror_mod.MeshRasterizer(g, None, context_type="cuda", is_orhto=False)
        out_mir = mir(mvp, only_alpha=True, iter_num=5, resolution=16, fit_normal=True, fit_depth=True, campos=campos)
    finally:
        mirror_mod.dr = real_dr
    assert rec.summary() == calls_ref, (rec.summary(), calls_ref)
    assert set(out_mir) == set(out_ref) == {"shaded", "geo_regularization", "n", "d"}
    for k in ("shaded", "n", "d"):
        assert out_mir[k].shape == out_ref[k].shape and torch.equal(out_mir[k], out_ref[k])
    # transform_pos: identical arithmetic
    pos = g.tet_v.detach()[g.surface_vid.long()]
    assert torch.equal(ref.transform_pos(mvp, pos), mir.transform_pos(mvp, pos))
    assert torch.equal(ref.transform_pos(mvp, pos, is_vec=True), mir.transform_pos(mvp, pos, is_vec=True))


@pytest.mark.parametrize("reference_modules", [_OracleDr], indirect=True)
def test_reference_forward_and_mirror_render_the_same_images(reference_modules, tmp_path):
    """Values, not just calls: the reference's `MeshRasterizer.forward` (unmodified) and this repo's mirror, both over a `dr` that
    answers with the raster oracle's images, return the SAME tensors -- alpha-only and shaded, with normals and depth -- for the same
    geometry object, cameras, background and material.  (The HIP `dr` against that oracle is tests/test_raster.py on the GPU.)"""
    geo, ren, rec = reference_modules
    path, v, t = _veg(tmp_path)
    g = geo.TetMeshGeometry(_Cfg(use_smooth_barrier=False, initial_mesh_path=path, smooth_barrier_param=None, optimize_geo=True))
    with torch.no_grad():                                   # a shape with some relief, deterministic
        g.tet_v.mul_(torch.tensor([1.0, 0.8, 1.15])).add_(0.03 * torch.sin(7.0 * g.tet_v.flip(1)))

    class Flat(torch.nn.Module):                            # stands in for materials.ExplicitMaterial: colour = f(position)
        def forward(self, positions):
            return {"color": torch.sigmoid(3.0 * positions)}
    from tssplat_amd import scenes
    import tssplat_amd.renderers as mirror_mod
    mvp = torch.from_numpy(scenes.orbit_mvps(2))
    campos = torch.tensor([[0.0, 1.0, 3.0], [2.0, 1.0, -2.0]])
    res = 24
    bg = torch.rand(2, res, res, 3, generator=torch.Generator().manual_seed(3))
    ref = ren.MeshRasterizer(g, Flat(), _Cfg(context_type="cuda", is_orhto=False))
    real_dr = mirror_mod.dr
    mirror_mod.dr = rec
    try:
        mir = mirror_mod.MeshRasterizer(g, Flat(), context_type="cuda", is_orhto=False)
        for only_alpha in (True, False):
            kw = dict(only_alpha=only_alpha, iter_num=7, resolution=res, fit_normal=True, fit_depth=True, campos=campos, background=bg)
            rec.calls.clear()
            a = ref(mvp, **kw)
            calls_ref = rec.summary()
            rec.calls.clear()
            b = mir(mvp, **kw)
            assert rec.summary() == calls_ref
            assert set(a) == set(b)
            for k in ("shaded", "n", "d"):
                assert a[k].shape == b[k].shape and torch.equal(a[k].detach(), b[k].detach()), k
            cover = float((a["shaded"][..., :1] > 0).float().mean()) if only_alpha else None
            if only_alpha:                                  # the images are not trivial: partly covered, antialiased edge pixels present
                assert 0.05 < cover < 0.9
                frac = a["shaded"].detach()
                assert ((frac > 0.02) & (frac < 0.98)).any()
            else:
                assert a["shaded"].shape == (2, res, res, 3) and float(a["n"].abs().max()) > 0.5 and float(a["d"].max()) > 1.0
    finally:
        mirror_mod.dr = real_dr
