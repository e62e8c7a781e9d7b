#!/usr/bin/env python3
"""How much does the ORDER of the triangle list matter to dr.rasterize?  512-sphere surface, 8 views x 512^2: mesh order
(boundary faces in tet order), Morton order of the centroids (3-D, per scene), sphere-major Morton, random."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tssplat_amd import geometry, scenes  # noqa: E402
import tssplat_amd.dr as dr  # noqa: E402


def morton3(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = np.minimum(((p - lo) / np.maximum(hi - lo, 1e-9) * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    code = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + d)
    return code


def timed(fn, reps=20):
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


def main():
    for kind, spheres, views in (("kuhn19", 512, 8), ("kuhn8", 20, 120)):
        sc = scenes.make_scene(kind, spheres)
        vid, faces = geometry.get_surface_vf(sc.tets)
        v = scenes.deform(sc, 0.02)[np.asarray(vid)]
        faces = np.asarray(faces, dtype=np.int32)
        pos = torch.from_numpy(scenes.transform_pos(scenes.orbit_mvps(views), v)).cuda()
        cen = v[faces].mean(axis=1)
        n_per = len(faces) // spheres
        sphere_of = np.arange(len(faces)) // n_per
        orders = {"mesh order": np.arange(len(faces)),
                  "morton (scene)": np.argsort(morton3(cen), kind="stable"),
                  "morton within sphere": np.lexsort((morton3(cen - cen.reshape(spheres, n_per, 3).mean(1)[sphere_of]), sphere_of)),
                  "random": np.random.default_rng(0).permutation(len(faces))}
        ctx = dr.RasterizeCudaContext()
        ref = None
        for name, o in orders.items():
            tri = torch.from_numpy(np.ascontiguousarray(faces[o])).cuda()
            ms = timed(lambda: dr.rasterize(ctx, pos, tri, resolution=[512, 512], grad_db=False))
            rast, _ = dr.rasterize(ctx, pos, tri, resolution=[512, 512], grad_db=False)
            cover = float((rast[..., 3] > 0).float().mean())
            print(f"{kind} x {spheres}, {views} views: {name:22s} rasterize {ms:.4f} ms  (coverage {cover:.4f})")


if __name__ == "__main__":
    main()
