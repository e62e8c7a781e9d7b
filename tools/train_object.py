#!/usr/bin/env python3
"""BASELINE config 5 on the reference's own object: the geometry-fitting loop of /root/reference/trainer.py:56-134 with the
schedule of config/gso.yaml, fitted to silhouettes of mesh_data/mario_example/model.obj (tests/golden/mario_mesh.npz: the GSO
mesh centred, scaled into the unit ball and decimated -- tests/golden/make_mesh_goldens.py).

What stands in for what the image does not have (img_data/ is empty in the reference, Mitsuba / TetWild / libpgo / omegaconf are
absent):
* the 120 target alpha images are rendered HERE, with tssplat_amd.dr, from the cameras of data/render_dataset.py:15-148
  (`scenes.dataset_mvps`: golden-ratio spiral at radius 4, fov 39.3, the script's look_at and perspective);
* the tet-spheres are kuhn balls (no TetWild) placed greedily in the visual hull of those images -- largest inscribed ball of
  the still-uncovered part first -- instead of the reference's offline key-point file (geometry.key_points_file_path, a MILP
  coverage solution; tetmesh_geometry.py:268-290 only scales and moves the template sphere to each key point);
* everything else is the reference's loop: `img_loss = MSE(alpha) * 20`, `loss = img_loss * 100 + reg`, `AdamUniform(lr 0.2,
  grad_limit 0.01)`, cosine schedule, order 2 -> 4 at iteration 1 000, smooth_eng_coeff / n_spheres (tetmesh_geometry.py:242).

    python tools/train_object.py [--spheres 20 --k 8 --views 120 --res 512 --iters 1500]

One JSON line: silhouette IoU over all views, inverted tets, seconds per iteration and the stand-alone stage times."""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def render_alpha(glctx, verts, tri, mvp, res):
    """Antialiased alpha image of a triangle mesh: the alpha branch of MeshRasterizer.forward (mesh_rasterizer.py:101-108)."""
    import torch
    import tssplat_amd.dr as dr
    posw = torch.cat([verts, torch.ones_like(verts[:, :1])], dim=1)
    pos_clip = torch.matmul(posw, mvp.transpose(1, 2)).contiguous()
    rast, _ = dr.rasterize(glctx, pos_clip, tri, resolution=[res, res], grad_db=False)
    alpha = torch.clamp(rast[..., -1:], 0, 1).contiguous()
    return dr.antialias(alpha, rast, pos_clip, tri)


def place_spheres(target_alpha, mvp, n_spheres, grid=72, min_radius=0.05, shrink=0.92):
    """Greedy ball placement in the visual hull of the target silhouettes.  A voxel of [-1, 1]^3 is inside when every view sees
    it on the object; the Euclidean distance transform of the hull gives each voxel's largest inscribed ball; pick the largest
    ball whose CENTRE is not yet covered, mark what it covers, repeat."""
    import numpy as np
    import torch
    from scipy import ndimage
    B, H, W = target_alpha.shape[:3]
    lin = (np.arange(grid) + 0.5) / grid * 2 - 1
    P = np.stack(np.meshgrid(lin, lin, lin, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float32)
    pts = torch.from_numpy(np.concatenate([P, np.ones((P.shape[0], 1), np.float32)], axis=1)).to(mvp.device)
    inside = torch.ones(P.shape[0], dtype=torch.bool, device=mvp.device)
    occ = target_alpha[..., 0] > 0.5
    for b in range(B):
        clip = pts @ mvp[b].T
        ndc = clip[:, :2] / clip[:, 3:4]
        px = ((ndc[:, 0] + 1) * 0.5 * W).floor().long().clamp(0, W - 1)
        py = ((ndc[:, 1] + 1) * 0.5 * H).floor().long().clamp(0, H - 1)       # row 0 = bottom, like the rasteriser's output
        inside &= occ[b, py, px] & (ndc.abs() < 1).all(dim=1)
    hull = inside.reshape(grid, grid, grid).cpu().numpy()
    h = 2.0 / grid
    dist = ndimage.distance_transform_edt(hull) * h
    uncovered = hull.copy()
    centres, radii = [], []
    for _ in range(n_spheres):
        score = np.where(uncovered, dist, 0.0)
        k = int(np.argmax(score))
        r = float(score.flat[k]) * shrink
        if r < min_radius:
            break
        c = P[k]
        centres.append(c)
        radii.append(r)
        uncovered &= (np.linalg.norm(P - c, axis=1) > r).reshape(hull.shape)
    return np.asarray(centres, np.float32), np.asarray(radii, np.float32), float(hull.mean()), float(uncovered.sum()) / max(1.0, float(hull.sum()))


def build_spheres(centres, radii, k):
    """`kuhn_ball(k)` scaled and moved to every (centre, radius): the concatenation of tetmesh_geometry.py:310-331."""
    import numpy as np
    from tssplat_amd import scenes
    bv, bt = scenes.kuhn_ball(k)
    bv = bv / np.linalg.norm(bv, axis=1).max()
    rest, tets = [], []
    for s, (c, r) in enumerate(zip(centres, radii)):
        rest.append((bv * r + c).astype(np.float32))
        tets.append(bt + s * bv.shape[0])
    return np.concatenate(rest), np.concatenate(tets).astype(np.int32)


def run(target_npz=None, n_spheres=20, k=8, views=120, res=512, iters=1500, log_every=100, stages=True, verbose=False):
    import numpy as np
    import torch
    import tssplat_amd.dr as dr
    from tssplat_amd import geometry, renderers, scenes
    from tssplat_amd.utils.optimizer import AdamUniform

    target_npz = target_npz or os.path.join(ROOT, "tests", "golden", "mario_mesh.npz")
    tgt = np.load(target_npz)
    tv = torch.from_numpy(tgt["vertices"].astype(np.float32)).cuda()
    tf = torch.from_numpy(tgt["faces"].astype(np.int32)).cuda()
    mvp = torch.from_numpy(scenes.dataset_mvps(views)).cuda()
    glctx = dr.RasterizeCudaContext()
    with torch.no_grad():
        target = render_alpha(glctx, tv, tf, mvp, res).clone()
    centres, radii, hull_frac, left = place_spheres(target, mvp, n_spheres)
    S = len(radii)
    rest, tets = build_spheres(centres, radii, k)
    flags = types.SimpleNamespace(smooth_eng_coeff=2e-4 / S, barrier_coeff=2e-4, increase_order_iter=1000)   # gso.yaml:8-11, tetmesh_geometry.py:242-243
    geo = geometry.TetMeshGeometry(rest, tets, smooth_barrier_param=flags)
    ren = renderers.MeshRasterizer(geo)
    opt = AdamUniform(ren.parameters(), lr=0.2, grad_limit=True, grad_limit_values=[0.01, 0.01], grad_limit_iters=[1500])   # gso.yaml:37-41
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, iters, eta_min=1e-4)                                           # trainer.py:57-58
    shade_loss = torch.nn.MSELoss()
    log = []

    def iou_now(it):
        with torch.no_grad():
            final = ren(mvp, only_alpha=True, iter_num=it, resolution=res)["shaded"]
            a, b = final[..., -1] > 0.5, target[..., -1] > 0.5
            return float((a & b).sum()) / max(1.0, float((a | b).sum()))

    iou0 = iou_now(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        out = ren(mvp, only_alpha=True, iter_num=it, resolution=res)
        img_loss = shade_loss(out["shaded"][..., -1], target[..., -1]) * 20            # trainer.py:99-101
        loss = img_loss * 100 + out["geo_regularization"]                              # trainer.py:115
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()
        if it % log_every == 0 or it == iters - 1:
            log.append((it, float(img_loss.detach()), float(out["geo_regularization"].detach())))
            if verbose:
                print(log[-1], file=sys.stderr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    iou = iou_now(iters)
    x = geo.tet_v.detach().cpu().numpy().astype(np.float64)

    def dets(p):
        return np.linalg.det(np.stack([p[tets[:, 1]] - p[tets[:, 0]], p[tets[:, 2]] - p[tets[:, 0]], p[tets[:, 3]] - p[tets[:, 0]]], axis=1))
    inverted = int((np.sign(dets(x)) != np.sign(dets(rest.astype(np.float64)))).sum())
    rec = {
        "metric": "s per iteration, geometry-fitting loop on the reference's object (BASELINE config 5)", "value": dt / iters, "unit": "s/iteration",
        "higher_is_better": False, "iterations": iters, "wall_s": dt, "ms_per_iteration": 1e3 * dt / iters,
        "silhouette_iou": iou, "silhouette_iou_at_start": iou0, "inverted_tets": inverted, "tets": int(tets.shape[0]),
        "img_loss_first_last": [log[0][1], log[-1][1]], "reg_first_last": [log[0][2], log[-1][2]],
        "spheres": {"placed": S, "asked": n_spheres, "radii_min_max": [float(radii.min()), float(radii.max())], "visual_hull_fraction_of_cube": hull_frac,
                    "hull_voxel_centres_left_uncovered": left},
        "config": {"workload": f"{S} tet-spheres x kuhn{k}: {int(tets.shape[0])} tets, {int(geo.surface_fid.shape[0])} surface triangles; {views} views x {res}^2 per iteration; "
                               f"target {os.path.basename(target_npz)} ({int(tv.shape[0])} vertices, {int(tf.shape[0])} triangles)",
                   "data": "silhouettes of the reference's mesh_data/mario_example/model.obj rendered with tssplat_amd.dr from data/render_dataset.py's cameras",
                   "schedule": "config/gso.yaml: lr 0.2 cosine, grad_limit 0.01, order 2 -> 4 at 1000, coeff_scheduler, smooth_eng_coeff / n_spheres"},
        "log": log,
    }
    if stages:      # the stages on their own (HIP events; they overlap nothing, so their sum is close to one iteration)
        def timed(fn, reps=20):
            for j in range(3):
                fn(j)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for j in range(reps):
                fn(3 + j)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        data = geo(iter_num=5)
        pos = ren.transform_pos(mvp, data.v_pos).contiguous().detach()
        tri = data.t_pos_idx
        rast, _ = dr.rasterize(ren.glctx, pos, tri, resolution=[res, res], grad_db=False)
        alpha = torch.clamp(rast[..., -1:], 0, 1).contiguous()
        pos_g = pos.clone().requires_grad_(True)
        ga = torch.randn_like(alpha)

        def geo_fb(j):
            geo.tet_v.grad = None
            d = geo(iter_num=j)
            (d.smooth_barrier_energy + d.v_pos.sum() * 0).backward()

        def aa_fb(j):
            pos_g.grad = None
            dr.antialias(alpha, rast, pos_g, tri).backward(ga)
        rec["stages_ms"] = {
            "energy_and_surface_forward_backward": timed(geo_fb),
            "rasterize": timed(lambda j: dr.rasterize(ren.glctx, pos, tri, resolution=[res, res], grad_db=False)),
            "antialias_forward_backward": timed(aa_fb),
            "optimizer_step": timed(lambda j: opt.step()),
        }
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--target", default=None)
    ap.add_argument("--spheres", type=int, default=20)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--views", type=int, default=120)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--iters", type=int, default=1500)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    print(json.dumps(run(args.target, args.spheres, args.k, args.views, args.res, args.iters, verbose=args.verbose)))


if __name__ == "__main__":
    main()
