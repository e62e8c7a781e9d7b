#!/usr/bin/env python3
"""Bench of the reference's inner loop (trainer.py:81-134, alpha fitting) on this package's kernels: TetMeshGeometry (energy +
surface gather) -> MeshRasterizer (transform, rasterize, antialias) -> MSE + regulariser -> backward -> AdamUniform.step.

    python tools/bench_pipeline.py [--scene kuhn19 --spheres 512 --views 8 --res 512 --iters 20]

One JSON line: ms per iteration and iterations/s, plus the stand-alone times of the stages (each timed on its own with
HIP events; they overlap nothing, so their sum is close to the iteration)."""
import argparse
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kuhn19")
    ap.add_argument("--spheres", type=int, default=512)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import torch
    from tssplat_amd import geometry, renderers, scenes
    from tssplat_amd.utils.optimizer import AdamUniform

    flags = types.SimpleNamespace(smooth_eng_coeff=2e-4, barrier_coeff=2e-4, increase_order_iter=1000)   # config/gso.yaml:8-11
    sc = scenes.make_scene(args.scene, args.spheres)
    geo = geometry.TetMeshGeometry(sc.rest, sc.tets, smooth_barrier_param=flags)
    ren = renderers.MeshRasterizer(geo)
    mvp = torch.from_numpy(scenes.orbit_mvps(args.views)).cuda()
    opt = AdamUniform(ren.parameters(), lr=0.2, grad_limit=True, grad_limit_values=[0.01, 0.01], grad_limit_iters=[1500])
    with torch.no_grad():
        target = (ren(mvp, only_alpha=True, iter_num=0, resolution=args.res)["shaded"] * 0.9).clone()
    mse = torch.nn.MSELoss()

    def iteration(it):
        out = ren(mvp, only_alpha=True, iter_num=it, resolution=args.res)
        loss = mse(out["shaded"][..., -1], target[..., -1]) * 2000 + out["geo_regularization"]
        opt.zero_grad()
        loss.backward()
        opt.step()

    def timed(fn, reps):
        for k in range(3):
            fn(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps):
            fn(3 + k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_iter = timed(iteration, args.iters)
    # stages on their own
    import tssplat_amd.dr as dr
    data = geo(iter_num=5)
    pos = ren.transform_pos(mvp, data.v_pos).contiguous().detach()
    tri = data.t_pos_idx
    res = [args.res, args.res]
    rast, _ = dr.rasterize(ren.glctx, pos, tri, resolution=res, grad_db=False)
    alpha = torch.clamp(rast[..., -1:], 0, 1).contiguous()
    pos_g = pos.clone().requires_grad_(True)
    ga = torch.randn_like(alpha)

    def geo_fb(it):
        geo.tet_v.grad = None
        d = geo(iter_num=it)
        (d.smooth_barrier_energy + d.v_pos.sum() * 0).backward()

    def aa_fb(it):
        pos_g.grad = None
        dr.antialias(alpha, rast, pos_g, tri).backward(ga)

    stages = {
        "geometry_forward_backward_ms": timed(geo_fb, args.iters),
        "rasterize_ms": timed(lambda it: dr.rasterize(ren.glctx, pos, tri, resolution=res, grad_db=False), args.iters),
        "antialias_forward_backward_ms": timed(aa_fb, args.iters),
        "optimizer_step_ms": timed(lambda it: opt.step(), args.iters),
    }
    print(json.dumps({
        "metric": "iterations/s of the alpha-fitting inner loop (trainer.py:81-134)", "value": 1e3 / t_iter, "unit": "it/s", "ms_per_iteration": t_iter,
        "stages": stages,
        "config": {"workload": f"{args.spheres} x {args.scene}: {sc.n_tets} tets, {int(tri.shape[0])} surface triangles; {args.views} views x {args.res}^2",
                   "optimizer": "AdamUniform lr 0.2 grad_limit 0.01 (config/gso.yaml:37-41)", "data": "synthetic"},
    }))


if __name__ == "__main__":
    main()
