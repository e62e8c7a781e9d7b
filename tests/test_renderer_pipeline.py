"""End-to-end mirror of the reference's inner loop (trainer.py:81-134) on this package's kernels only: TetMeshGeometry
(energy + surface gather) -> MeshRasterizer (transform, rasterize, antialias [, interpolate]) -> image loss + geometry
regulariser -> backward -> AdamUniform.  GPU tests; the CPU test checks the modules refuse to run without a GPU."""
import types

import os

import numpy as np
import pytest

from oracle import raster_oracle as R
from tssplat_amd import scenes

FLAGS = types.SimpleNamespace(smooth_eng_coeff=2e-4, barrier_coeff=2e-4, increase_order_iter=1000)      # config/gso.yaml:8-11


def test_modules_refuse_cpu():
    torch = pytest.importorskip("torch")
    from tssplat_amd import geometry
    sc = scenes.make_scene("kuhn4", 1)
    with pytest.raises(RuntimeError):
        geometry.TetMeshGeometry(sc.rest, sc.tets, smooth_barrier_param=FLAGS, device="cpu")


def _target_alpha(geo, renderer, mvp, res, scale, shift):
    """Antialiased silhouettes of the SAME mesh scaled and shifted: the thing the fit must reach."""
    import torch
    with torch.no_grad():
        keep = geo.tet_v.data.clone()
        c = keep.mean(0, keepdim=True)
        geo.tet_v.data.copy_((keep - c) * scale + c + torch.tensor(shift, device=keep.device))
        out = renderer(mvp, only_alpha=True, iter_num=0, resolution=res)["shaded"].clone()
        geo.tet_v.data.copy_(keep)
    return out


@pytest.mark.gpu
def test_forward_keys_shapes_and_gradient_flow():
    import torch
    from tssplat_amd import geometry, renderers
    sc = scenes.make_scene("kuhn8", 2)
    geo = geometry.TetMeshGeometry(sc.rest, sc.tets, smooth_barrier_param=FLAGS)
    ren = renderers.MeshRasterizer(geo)
    mvp = torch.from_numpy(R.orbit_mvps(3)).cuda()
    campos = torch.tensor([[0.0, 1.0, 3.0]] * 3, device="cuda")
    bg = torch.ones(3, 64, 64, 3, device="cuda")

    class Flat(torch.nn.Module):                                  # stands in for materials.ExplicitMaterial: colour = f(position)
        def forward(self, positions):
            return {"color": torch.sigmoid(positions)}
    ren.materials = Flat()
    out = ren(mvp, only_alpha=False, iter_num=5, resolution=64, fit_normal=True, fit_depth=True, background=bg, campos=campos)
    assert out["shaded"].shape == (3, 64, 64, 3) and out["n"].shape == (3, 64, 64, 3) and out["d"].shape == (3, 64, 64, 1)
    assert out["geo_regularization"].dim() == 0
    loss = out["shaded"].square().mean() + out["n"].sum() * 1e-3 + out["d"].mean() * 1e-3 + out["geo_regularization"]
    loss.backward()
    g = geo.tet_v.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
    # alpha only: the gradient reaches the geometry through antialias alone (plus the regulariser)
    geo.tet_v.grad = None
    out = ren(mvp, only_alpha=True, iter_num=5, resolution=64)
    assert out["shaded"].shape == (3, 64, 64, 1) and float(out["shaded"].detach().min()) >= 0 and float(out["shaded"].detach().max()) <= 1
    out["shaded"].sum().backward()
    inside = torch.zeros(geo.tet_v.shape[0], dtype=torch.bool, device="cuda")
    inside[geo.surface_vid.long()] = True
    assert geo.tet_v.grad[inside].abs().max() > 0 and geo.tet_v.grad[~inside].abs().max() == 0


@pytest.mark.gpu
def test_silhouette_fit_converges():
    import torch
    from tssplat_amd import geometry, renderers
    from tssplat_amd.utils.optimizer import AdamUniform
    torch.manual_seed(0)
    sc = scenes.make_scene("kuhn8", 1)
    geo = geometry.TetMeshGeometry(sc.rest, sc.tets, smooth_barrier_param=FLAGS)
    ren = renderers.MeshRasterizer(geo)
    res, views = 96, 6
    mvp = torch.from_numpy(R.orbit_mvps(views)).cuda()
    target = _target_alpha(geo, ren, mvp, res, scale=1.3, shift=[0.08, -0.05, 0.0])
    opt = AdamUniform(ren.parameters(), lr=0.2, grad_limit=True, grad_limit_values=[0.01, 0.01], grad_limit_iters=[1500])   # config/gso.yaml:37-41
    shade_loss = torch.nn.MSELoss()                                # trainer.py:42
    losses, energies = [], []
    for it in range(150):
        out = ren(mvp, only_alpha=True, iter_num=it, resolution=res)
        img_loss = shade_loss(out["shaded"][..., -1], target[..., -1]) * 20          # trainer.py:99-104
        loss = img_loss * 100 + out["geo_regularization"]                             # trainer.py:115
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(img_loss.detach()))
        energies.append(float(out["geo_regularization"].detach()))
    assert all(np.isfinite(losses)) and all(np.isfinite(energies))
    assert losses[-1] < 0.35 * losses[0], (losses[0], losses[-1])   # the silhouettes moved most of the way
    # and the regulariser held the volume mesh together: (almost) no tetrahedron inverted on the way (it is a penalty, not a wall)
    x = geo.tet_v.detach().cpu().numpy().astype(np.float64)
    t = sc.tets
    d = np.linalg.det(np.stack([x[t[:, 1]] - x[t[:, 0]], x[t[:, 2]] - x[t[:, 0]], x[t[:, 3]] - x[t[:, 0]]], axis=1))
    d0 = np.linalg.det(np.stack([sc.rest[t[:, 1]] - sc.rest[t[:, 0]], sc.rest[t[:, 2]] - sc.rest[t[:, 0]], sc.rest[t[:, 3]] - sc.rest[t[:, 0]]], axis=1))
    assert (np.sign(d) == np.sign(d0)).mean() > 0.99


def test_dataset_cameras_follow_the_reference_script():
    """scenes.dataset_mvps = data/render_dataset.py:15-57,96-148: golden-ratio spiral at radius 4, its look_at (eye on -z of the
    view frame, right = lookat x up) and its perspective (flipped y row, fov 39.3077, near 0.001, far 10)."""
    m = scenes.dataset_mvps(120).astype(np.float64)
    assert m.shape == (120, 4, 4)
    o = m @ np.array([0.0, 0.0, 0.0, 1.0])                        # the origin sits in the image centre, 4 in front of every eye
    assert np.abs(o[:, :2]).max() < 1e-6 and np.allclose(o[:, 3], 4.0, atol=1e-5)
    t = np.tan(np.radians(39.3077) / 2)
    # a point one unit "up" in the first view's frame lands at y_ndc = -1 / (t * w) (the script flips y)
    golden = (1 + 5 ** 0.5) / 2
    i = 7
    theta, phi = 2 * np.pi * i / golden, np.arccos(1 - 2 * i / 120)
    eye = 4 * np.array([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)])
    look = -eye / 4
    right = np.cross(look, [0, 0, 1.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, look)
    c = m[i] @ np.append(0.5 * up, 1.0)
    assert abs(c[0] / c[3]) < 1e-6 and np.isclose(c[1] / c[3], -0.5 / (t * 4.0), atol=1e-6)
    c = m[i] @ np.append(0.5 * right, 1.0)
    assert np.isclose(c[0] / c[3], 0.5 / (t * 4.0), atol=1e-6) and abs(c[1] / c[3]) < 1e-6


def test_mesh_goldens_from_the_reference():
    """tests/golden/make_mesh_goldens.py: the reference's template sphere verbatim (closed, 1 500 / 2 996) and its one object."""
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    s1 = np.load(os.path.join(g, "s1_sphere.npz"))
    v, f = s1["vertices"], s1["faces"]
    assert v.shape == (1500, 3) and f.shape == (2996, 3) and v.dtype == np.float32 and f.dtype == np.int32
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all() and len(cnt) == 3 * 2996 // 2           # closed 2-manifold: V - E + F = 2
    assert 1500 - len(cnt) + 2996 == 2
    assert abs(np.linalg.norm(v, axis=1).mean() - 1.0) < 0.05       # a unit sphere
    m = np.load(os.path.join(g, "mario_mesh.npz"))
    assert np.linalg.norm(m["vertices"], axis=1).max() <= 1.0 + 1e-6 and m["faces"].max() < m["vertices"].shape[0]
    cv, ct = scenes.cone_sphere(v, f)                               # BASELINE config 1's literal fixture: one hub of valence 2 996
    assert ct.shape == (2996, 4) and np.bincount(ct.reshape(-1)).max() == 2996


@pytest.mark.gpu
def test_config5_fit_to_the_reference_object():
    """BASELINE config 5 with the reference's own object: tet-spheres placed in the visual hull of silhouettes
    of mesh_data/mario_example/model.obj (golden: mario_mesh.npz) and fitted with the reference's loop and schedule
    (tools/train_object.py): the full 120 views x 512^2 x 1 500 iterations of config/gso.yaml -- about five seconds on an MI355X
    (profiles/r04_train_mario.json: IoU 0.971, 2.7 ms per iteration)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_object", os.path.join(root, "tools", "train_object.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rec = mod.run(n_spheres=20, k=8, views=120, res=512, iters=1500, stages=False)
    assert rec["spheres"]["placed"] >= 12
    assert rec["silhouette_iou_at_start"] < 0.7 < 0.9 <= rec["silhouette_iou"], rec["silhouette_iou"]
    assert rec["inverted_tets"] <= 0.01 * rec["tets"]
    assert rec["img_loss_first_last"][1] < 0.15 * rec["img_loss_first_last"][0]
