#!/usr/bin/env python3
"""Surface glue: fused per-vertex gather kernels vs the reference's torch op chain
(index, cross, three scatter_add_, where, normalize -- geometry/tetmesh_geometry.py:33,39-66) on the same
GPU, forward + backward, on the boundary of S x kuhn_ball(k).

    python tools/bench_surface.py [--spheres 512] [--k 19] [--steps 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spheres", type=int, default=512)
    ap.add_argument("--k", type=int, default=19)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F
    from tssplat_amd import geometry, scenes
    v, t = scenes.kuhn_ball(args.k)
    vid1, f1 = geometry.get_surface_vf(t)
    S, nv, ns = args.spheres, v.shape[0], vid1.size
    vid = np.concatenate([vid1 + s * nv for s in range(S)]).astype(np.int32)
    faces = np.concatenate([f1 + s * ns for s in range(S)]).astype(np.int32)
    sc = scenes.replicate_spheres(v, t, S)
    x = torch.from_numpy(scenes.deform(sc, 0.02)).cuda()
    vid_t, f_t = torch.from_numpy(vid).cuda(), torch.from_numpy(faces).cuda()
    w = torch.randn(len(vid), 3, device="cuda")
    i0, i1, i2 = (f_t[:, c].long() for c in range(3))
    idx = [i[:, None].repeat(1, 3) for i in (i0, i1, i2)]
    vid_l = vid_t.long()

    def chain(tet_v):
        vp = tet_v[vid_l]
        fn = torch.cross(vp[i1] - vp[i0], vp[i2] - vp[i0], dim=1)
        n = torch.zeros_like(vp)
        for i in idx:
            n = n.scatter_add(0, i, fn)
        n = torch.where((n * n).sum(-1, keepdim=True) > 1e-20, n, torch.tensor([0.0, 0.0, 1.0], device=n.device))
        return F.normalize(n, dim=1)

    ops = geometry.SurfaceOps(vid_t, f_t, x.shape[0])

    def fused(tet_v):
        return geometry.TetMeshGeometryForwardData(tet_v, None, vid_t, f_t, surface_ops=ops)._compute_vertex_normal()

    out = {"surface_vertices": int(len(vid)), "faces": int(len(faces)), "tet_vertices": int(x.shape[0])}
    for name, fn in (("fused", fused), ("torch_ops", chain)):
        p = x.clone().requires_grad_(True)
        for _ in range(5):
            p.grad = None
            (fn(p) * w).sum().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            p.grad = None
            (fn(p) * w).sum().backward()
        torch.cuda.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) / args.steps * 1e3
    out["speedup"] = out["torch_ops_ms"] / out["fused_ms"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
