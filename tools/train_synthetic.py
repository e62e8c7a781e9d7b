#!/usr/bin/env python3
"""A synthetic stand-in for BASELINE config 5 (`trainer.py` with `config/gso.yaml`): the reference's geometry-fitting loop
(trainer.py:56-134) on this package's modules -- TetMeshGeometry + MeshRasterizer + AdamUniform + torch's cosine schedule --
against silhouettes rendered from a deformed copy of the starting mesh (the reference's data, `img_data/`, TetWild and its
config loader are not in this image; everything else about the loop is the reference's: 120 views x 512^2 per iteration,
`img_loss = MSE(alpha) * 20`, `loss = img_loss * 100 + reg`, `AdamUniform(lr 0.2, grad_limit 0.01)`, 1 500 iterations with
the order switch at 1 000, a loss print-out every iteration replaced by one every `--log-every`).

    python tools/train_synthetic.py [--scene kuhn19 --spheres 1 --views 120 --res 512 --iters 1500]

One JSON line: seconds per iteration (wall clock, whole loop), start / end image loss, intersection-over-union of the
final silhouettes with the targets, fraction of tetrahedra that kept their orientation."""
import argparse
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kuhn19")
    ap.add_argument("--spheres", type=int, default=1)
    ap.add_argument("--views", type=int, default=120)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--iters", type=int, default=1500)
    ap.add_argument("--log-every", type=int, default=100)
    args = ap.parse_args()
    import numpy as np
    import torch
    from tssplat_amd import geometry, renderers, scenes
    from tssplat_amd.utils.optimizer import AdamUniform

    flags = types.SimpleNamespace(smooth_eng_coeff=2e-4 / args.spheres, barrier_coeff=2e-4, increase_order_iter=1000)   # gso.yaml:8-11, tetmesh_geometry.py:24x
    sc = scenes.make_scene(args.scene, args.spheres)
    geo = geometry.TetMeshGeometry(sc.rest, sc.tets, smooth_barrier_param=flags)
    ren = renderers.MeshRasterizer(geo)
    # cameras: two elevations x views / 2 azimuths around the scene's centre
    centre = sc.rest.mean(0)
    half = max(1, args.views // 2)
    mvps = np.concatenate([scenes.orbit_mvps(half, elevation_deg=15.0), scenes.orbit_mvps(args.views - half, elevation_deg=-35.0)])
    shift = np.eye(4, dtype=np.float32)
    shift[:3, 3] = -centre
    mvp = torch.from_numpy(mvps @ shift).cuda()
    # target: the same mesh stretched to an ellipsoid and moved
    with torch.no_grad():
        keep = geo.tet_v.data.clone()
        c = torch.from_numpy(centre).cuda()
        s = torch.tensor([1.45, 0.8, 1.15], device="cuda")
        geo.tet_v.data.copy_((keep - c) * s + c + torch.tensor([0.06, -0.04, 0.03], device="cuda"))
        target = ren(mvp, only_alpha=True, iter_num=0, resolution=args.res)["shaded"].clone()
        geo.tet_v.data.copy_(keep)
    opt = AdamUniform(ren.parameters(), lr=0.2, grad_limit=True, grad_limit_values=[0.01, 0.01], grad_limit_iters=[1500])   # gso.yaml:37-41
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, args.iters, eta_min=1e-4)                                      # trainer.py:57-58
    shade_loss = torch.nn.MSELoss()
    log = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.iters):
        out = ren(mvp, only_alpha=True, iter_num=it, resolution=args.res)
        img_loss = shade_loss(out["shaded"][..., -1], target[..., -1]) * 20
        loss = img_loss * 100 + out["geo_regularization"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()
        if it % args.log_every == 0 or it == args.iters - 1:
            log.append((it, float(img_loss.detach()), float(out["geo_regularization"].detach())))      # (a device read, like the reference's print)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with torch.no_grad():
        final = ren(mvp, only_alpha=True, iter_num=args.iters, resolution=args.res)["shaded"]
        a, b = final[..., -1] > 0.5, target[..., -1] > 0.5
        iou = float((a & b).sum()) / max(1.0, float((a | b).sum()))
    x = geo.tet_v.detach().cpu().numpy().astype(np.float64)
    t = sc.tets

    def dets(p):
        return np.linalg.det(np.stack([p[t[:, 1]] - p[t[:, 0]], p[t[:, 2]] - p[t[:, 0]], p[t[:, 3]] - p[t[:, 0]]], axis=1))
    kept = float((np.sign(dets(x)) == np.sign(dets(sc.rest.astype(np.float64)))).mean())
    print(json.dumps({
        "metric": "s per iteration, geometry-fitting loop (synthetic stand-in for BASELINE config 5)", "value": dt / args.iters, "unit": "s/iteration",
        "higher_is_better": False, "iterations": args.iters, "wall_s": dt,
        "img_loss_first_last": [log[0][1], log[-1][1]], "reg_first_last": [log[0][2], log[-1][2]], "silhouette_iou": iou, "tets_orientation_kept": kept,
        "config": {"workload": f"{args.spheres} x {args.scene}: {sc.n_tets} tets; {args.views} views x {args.res}^2 per iteration", "data": "synthetic",
                   "schedule": "config/gso.yaml: lr 0.2 cosine, grad_limit 0.01, order 2 -> 4 at 1000, coeff_scheduler"},
        "log": log,
    }))


if __name__ == "__main__":
    main()
