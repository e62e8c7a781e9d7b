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
    mirror's return identical tensors (alpha-only and shaded paths, normals, depth) for the same geometry, cameras and material;
  * the reference's `trainer.train(cfg)` itself runs, unmodified, for a few iterations over this repo's module surface
    (`pypgo` shim real; `tet_spheres.tet_spheres_ext` and `nvdiffrast.torch` answered by the oracles under this repo's names and
    signatures): the loss falls, the exports re-load.
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


class _OracleAutogradDr(_OracleDr):
    """The oracle-backed stand-in with backward passes (oracle/raster_oracle.py's analytic gradients behind autograd nodes): what the
    reference's training loop needs to run end to end on the CPU."""

    def rasterize(self, glctx, pos, tri, resolution, *args, **kwargs):
        from oracle import raster_oracle as RO
        self.calls.append(("rasterize", (glctx, pos, tri, resolution) + args, kwargs))

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, pos):
                rast = torch.from_numpy(RO.rasterize(pos.detach().numpy(), tri.numpy(), resolution).astype(np.float32))
                ctx.save_for_backward(pos, rast)
                return rast

            @staticmethod
            def backward(ctx, g):
                pos, rast = ctx.saved_tensors
                return torch.from_numpy(RO.rasterize_backward(pos.numpy(), tri.numpy(), rast.numpy(), g.numpy()).astype(np.float32)).reshape(pos.shape)
        rast = Fn.apply(pos)
        return rast, torch.zeros_like(rast)

    def interpolate(self, attr, rast, tri, *args, **kwargs):
        from oracle import raster_oracle as RO
        self.calls.append(("interpolate", (attr, rast, tri) + args, kwargs))

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, attr, rast):
                ctx.save_for_backward(attr, rast)
                return torch.from_numpy(RO.interpolate(attr.detach().numpy(), rast.detach().numpy(), tri.numpy()).astype(np.float32))

            @staticmethod
            def backward(ctx, g):
                attr, rast = ctx.saved_tensors
                ga, gr = RO.interpolate_backward(attr.numpy(), rast.numpy(), tri.numpy(), g.numpy())
                return torch.from_numpy(ga.astype(np.float32)), torch.from_numpy(gr.astype(np.float32))
        return Fn.apply(attr, rast), None

    def antialias(self, color, rast, pos, tri, *args, **kwargs):
        from oracle import raster_oracle as RO
        self.calls.append(("antialias", (color, rast, pos, tri) + args, kwargs))
        boost = kwargs.get("pos_gradient_boost", 1.0)

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, color, pos):
                ctx.save_for_backward(color, pos)
                return torch.from_numpy(RO.antialias(color.detach().numpy(), rast.detach().numpy(), pos.detach().numpy(), tri.numpy()).astype(np.float32))

            @staticmethod
            def backward(ctx, g):
                color, pos = ctx.saved_tensors
                gc, gp = RO.antialias_backward(color.numpy(), rast.detach().numpy(), pos.numpy(), tri.numpy(), g.numpy(), pos_gradient_boost=boost)
                return torch.from_numpy(gc.astype(np.float32)), torch.from_numpy(gp.astype(np.float32)).reshape(pos.shape)
        return Fn.apply(color, pos)


@pytest.mark.parametrize("reference_modules", [_OracleAutogradDr], indirect=True)
def test_reference_trainer_runs_unmodified_over_this_repos_module_surface(reference_modules, tmp_path, monkeypatch):
    """`trainer.train(cfg)` of the reference, UNMODIFIED (trainer.py:35-185), for a few iterations on the CPU: its own
    `TetMeshMultiSphereGeometry` (pre-computed mesh branch), `SmoothnessBarrierEnergy` + autograd function, `MeshRasterizer`,
    `AdamUniform`, cosine schedule, exports -- on top of this repo's `pypgo` shim and of stand-ins that answer under THIS repo's
    module names and signatures: `tet_spheres.tet_spheres_ext` (TetSpheres / forward / backward computed by the oracle, every call
    also bound to the signature of the same name in tssplat_amd.tet_spheres_ext) and `nvdiffrast.torch` (the raster oracle with its
    analytic backward, every call bound to tssplat_amd.dr).  The data loader (image files, cv2) and TetWild are replaced by a
    synthetic batch source / a mesh written here: they are outside the path.  Asserts: the loop runs, the loss it optimises
    falls, the geometry moves, the exports the trainer writes exist and re-load through the shim."""
    geo_mod, ren_mod, rec = reference_modules
    import json
    from oracle import torch_energies as TE
    import tssplat_amd.tet_spheres_ext as our_ext
    import tssplat_amd.dr as our_dr
    from tssplat_amd import scenes

    # ---- tet_spheres.tet_spheres_ext: the names the reference calls (energies/smooth_barrier.py:6-45), answered by the oracle ----
    calls = []

    class StandInTetSpheres:
        def __init__(self, v_flat, f_flat):
            inspect.signature(our_ext.TetSpheres.__init__).bind(None, v_flat, f_flat)
            self.ts = TE.TorchTetSpheres(np.asarray(v_flat, np.float32).reshape(-1, 3), np.asarray(f_flat, np.int32).reshape(-1, 4))

    def ext_forward(x, tet_sp, c1, c2, order):
        inspect.signature(our_ext.forward).bind(x, tet_sp, c1, c2, order)
        calls.append("forward")
        return TE.compute_energy(x.detach(), tet_sp.ts, c1, c2, order)

    def ext_backward(grad_output, x, tet_sp, c1, c2, order):
        inspect.signature(our_ext.backward).bind(grad_output, x, tet_sp, c1, c2, order)
        calls.append("backward")
        return TE.compute_energy_backward(grad_output, x.detach(), tet_sp.ts, c1, c2, order)
    ext = types.ModuleType("tet_spheres.tet_spheres_ext")
    ext.TetSpheres, ext.forward, ext.backward = StandInTetSpheres, ext_forward, ext_backward
    pkg = types.ModuleType("tet_spheres")
    pkg.tet_spheres_ext = ext
    sb = sys.modules["energies.smooth_barrier"]                  # (already imported by the fixture, against the real shim)
    monkeypatch.setattr(sb, "tet_spheres_ext", ext)

    # ---- the pre-computed multi-sphere mesh the reference's geometry class loads (tetmesh_geometry.py:221-231) ----
    sc = scenes.make_scene("kuhn3", 2)
    mesh_dir = tmp_path / "init"
    mesh_dir.mkdir()
    scenes.write_veg(mesh_dir / "final.veg", sc.rest, sc.tets)
    nv, nt = sc.rest.shape[0] // 2, sc.tets.shape[0] // 2
    json.dump([list(range(i * nv, (i + 1) * nv)) for i in range(2)], open(mesh_dir / "spheres_vtx_idx.json", "w"))
    json.dump([sc.tets[i * nt:(i + 1) * nt].tolist() for i in range(2)], open(mesh_dir / "spheres_elem_idx.json", "w"))

    # ---- stubs for what lies outside the path: trimesh's OBJ writer, the data package, materials' loader ----
    class Trimesh:
        def __init__(self, vertices=None, faces=None, **kw):
            self.vertices, self.faces = np.asarray(vertices), np.asarray(faces)

        def export(self, path):
            with open(path, "w") as f:
                f.writelines(f"v {a} {b} {c}\n" for a, b, c in self.vertices)
                f.writelines(f"f {a + 1} {b + 1} {c + 1}\n" for a, b, c in self.faces)
    monkeypatch.setattr(sys.modules["trimesh"], "Trimesh", Trimesh, raising=False)
    res, views = 20, 3
    mvp = torch.from_numpy(scenes.orbit_mvps(views))

    class Loader:
        num_forward_per_iter = 1

        def __init__(self, cfg):
            from oracle import raster_oracle as RO
            # targets: silhouettes of the same two balls, 12 % larger (what the fit must grow into)
            v, f = sys.modules["geometry.tetrahedron_mesh"].get_surface_vf(sc.tets)
            pos = np.concatenate([sc.rest[v] * 1.12, np.ones((len(v), 1), np.float32)], 1)
            clip = np.einsum("vj,bij->bvi", pos, mvp.numpy()).astype(np.float32)
            rast = RO.rasterize(clip, f, [res, res])
            alpha = RO.antialias(np.clip(rast[..., 3:], 0, 1), rast, clip, f)
            self.img = torch.from_numpy(np.concatenate([np.ones((views, res, res, 3)), alpha], -1).astype(np.float32))

        def __call__(self, it, forw_id):
            return {"img": self.img, "mvp": mvp, "resolution": res, "background": torch.ones(views, res, res, 3),
                    "campos": torch.zeros(views, 3), "d": torch.zeros(views, res, res, 1), "n": torch.zeros(views, res, res, 3)}
    data = types.ModuleType("data")
    data.load_dataloader = lambda kind: Loader
    monkeypatch.setitem(sys.modules, "data", data)
    sys.modules["materials"].load_material = lambda kind: None
    monkeypatch.setattr(sys.modules["utils.config"], "load_config", lambda *a, **k: None, raising=False)

    class Cfg(dict):                                               # the slice of omegaconf.DictConfig the trainer uses
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    out_dir = tmp_path / "results"
    cfg = Cfg(fitting_stage="geometry", geometry_type="TetMeshMultiSphereGeometry", dataloader_type="X", data=Cfg(), material_type=None,
              geometry=Cfg(initial_mesh_path=str(mesh_dir), use_smooth_barrier=True,
                           smooth_barrier_param=Cfg(smooth_eng_coeff=2e-4, barrier_coeff=2e-4, increase_order_iter=1000),   # config/gso.yaml:8-11
                           template_surface_sphere_path="", key_points_file_path="", tetwild_exec="", tetwild_cache_folder="",
                           load_precomputed_tetwild_mesh=False, debug_mode=False, optimize_geo=True, output_path=str(out_dir)),
              renderer=Cfg(context_type="cuda", is_orhto=False),
              optimizer=Cfg(lr=0.02, grad_limit=True, grad_limit_values=[0.01, 0.01], grad_limit_iters=[1500]),
              output_path=str(out_dir), total_num_iter=6, use_permute_surface_v=False, verbose=False)
    monkeypatch.syspath_prepend(REF)
    for name in ("geometry", "renderers"):                          # `from geometry import load_geometry` etc.: the packages themselves
        importlib.import_module(name)
    trainer = importlib.import_module("trainer")
    try:
        losses = []
        monkeypatch.setattr(trainer.tqdm, "write", lambda msg, *a, **k: losses.append(float(msg.split("img_loss=")[1].split(",")[0])))
        rec.calls.clear()
        trainer.train(cfg)
    finally:
        sys.modules.pop("trainer", None)
    assert trainer.__file__.startswith(REF)
    assert len(losses) == 6 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert calls.count("forward") == 6 and calls.count("backward") == 6            # the energy, every iteration, both directions
    names = [c[0] for c in rec.calls]
    assert names.count("rasterize") == 6 and names.count("antialias") == 6
    for name, args, kwargs in rec.calls:                          # every renderer call binds to this repo's dr
        inspect.signature(getattr(our_dr, name)).bind(*args, **kwargs)
    # what the trainer wrote (trainer.py:141-143, 178): first-iteration and final exports, re-loaded through the pypgo shim
    import pypgo
    for sub, stem in (("mesh00000", "00000"), ("final", "final")):
        m = pypgo.create_tetmesh_from_file(str(out_dir / sub / f"{stem}.veg"))
        assert np.array_equal(pypgo.get_tetmesh_element_indices(m).reshape(-1, 4), sc.tets)
    moved = pypgo.get_tetmesh_vertex_positions(m).reshape(-1, 3) - sc.rest
    assert 1e-4 < np.abs(moved).max() < 0.5
    assert (out_dir / "final" / "final_vtx.npy").exists() and (out_dir / "final" / "final_sp1_elem.npy").exists()
