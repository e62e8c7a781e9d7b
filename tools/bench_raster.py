#!/usr/bin/env python3
"""Bench of the renderer slice: tssplat_amd.dr.rasterize + interpolate forward / backward on the surface
of the headline scene (512 x kuhn19: 2.2 M boundary triangles, 1.1 M surface vertices), 8 views x 512 x 512.

    python tools/bench_raster.py [--spheres 512 --views 8 --res 512 --reps 20]

One JSON line: pixels/s of rasterize (+ interpolate, antialias, the backward passes), and a roofline object for the rasterize pair of kernels with
ALGORITHMIC bytes per call = views x (12 B x triangles + 16 B x vertices + 16 B x pixels) -- every index and clip-space
vertex read once per view, every output pixel written once -- against the 8 TB/s HBM peak.  The first slice is one
lane per (view, triangle) with 64-bit atomics: expect a small fraction.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kuhn19")
    ap.add_argument("--spheres", type=int, default=512)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    import torch
    from tssplat_amd import geometry, scenes
    import tssplat_amd.dr as dr

    sc = scenes.make_scene(args.scene, args.spheres)
    vid, faces = geometry.get_surface_vf(sc.tets)
    v = scenes.deform(sc, 0.02)[np.asarray(vid)]
    mvp = scenes.orbit_mvps(args.views)
    pos = torch.from_numpy(scenes.transform_pos(mvp, v)).cuda()
    tri = torch.from_numpy(np.asarray(faces, dtype=np.int32)).cuda()
    attr = torch.from_numpy(v[None].astype(np.float32)).cuda().requires_grad_(True)
    ctx = dr.RasterizeCudaContext()
    res = [args.res, args.res]

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    rast, _ = dr.rasterize(ctx, pos, tri, resolution=res, grad_db=False)
    cover = float((rast[..., 3] > 0).float().mean())
    t_rast = timed(lambda: dr.rasterize(ctx, pos, tri, resolution=res, grad_db=False), args.reps)
    t_interp = timed(lambda: dr.interpolate(attr.detach(), rast, tri), args.reps)
    g = torch.randn(args.views, args.res, args.res, 3, device="cuda")

    def fwd_bwd():
        attr.grad = None
        out, _ = dr.interpolate(attr, rast, tri)
        out.backward(g)

    t_fb = timed(fwd_bwd, args.reps)

    # the reference's alpha path (mesh_rasterizer.py:103-108): rasterize -> clamp(id) -> antialias, and its backward to pos
    pos_g = pos.clone().requires_grad_(True)
    alpha = torch.clamp(rast[..., -1:], 0, 1).contiguous()
    topo = dr.antialias_construct_topology_hash(tri)
    t_topo = timed(lambda: dr.antialias_construct_topology_hash(tri), max(2, args.reps // 4))
    t_aa = timed(lambda: dr.antialias(alpha, rast, pos, tri, topology_hash=topo), args.reps)
    ga = torch.randn_like(alpha)

    def aa_fwd_bwd():
        pos_g.grad = None
        dr.antialias(alpha, rast, pos_g, tri, topology_hash=topo).backward(ga)

    t_aa_fb = timed(aa_fwd_bwd, args.reps)
    aa = dr.antialias(alpha, rast, pos, tri, topology_hash=topo)
    blended = int(((aa - alpha).abs().sum(-1) > 0).sum())

    def rast_interp_fwd_bwd():
        pos_g.grad = None
        r, _ = dr.rasterize(ctx, pos_g, tri, resolution=res, grad_db=False)
        out, _ = dr.interpolate(attr.detach(), r, tri)
        out.backward(g)

    t_rb = timed(rast_interp_fwd_bwd, args.reps)
    T, V, px = int(tri.shape[0]), int(pos.shape[1]), args.views * args.res * args.res
    b_alg = args.views * (12.0 * T + 16.0 * V) + 16.0 * px
    print(json.dumps({
        "metric": "pixels/s (rasterize, 8 views x 512^2 of the 512-sphere surface)", "value": px / (t_rast * 1e-3), "unit": "pixels/s",
        "rasterize_ms": t_rast, "interpolate_ms": t_interp, "interpolate_fwd_bwd_ms": t_fb,
        "antialias_ms": t_aa, "antialias_fwd_bwd_ms": t_aa_fb, "antialias_topology_ms": t_topo, "antialias_blended_pixels": blended,
        "rasterize_interpolate_fwd_bwd_to_pos_ms": t_rb,
        "triangle_views_per_s": args.views * T / (t_rast * 1e-3),
        "config": {"workload": f"{args.spheres} x {args.scene} surface: {T} triangles, {V} vertices; {args.views} views x {args.res}^2, "
                               f"coverage {cover:.3f}", "dtype": "f32 (coverage / depth test: int64 + f64)"},
        "roofline": {"bound": "hbm", "kernel": "rasterize_bin_kernel + rasterize_resolve_kernel", "achieved": b_alg / (t_rast * 1e-3) / 1e9,
                     "peak": 8000.0, "unit": "GB/s", "frac": b_alg / (t_rast * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_call": b_alg,
                     "traffic": None},
    }))


if __name__ == "__main__":
    main()
