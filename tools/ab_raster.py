#!/usr/bin/env python3
"""A/B timing of library variants (tssplat_amd/_build.py::build_variant) on dr.rasterize: the 512-sphere surface in 8 views and
20 x kuhn8 in 120 views, 512^2.  One process per variant (TSSPLAT_AMD_LIB), rounds interleaved.

    python tools/ab_raster.py base old id4 [--rounds 3]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch
    from tssplat_amd import geometry, scenes
    import tssplat_amd.dr as dr
    out = {}
    for name, kind, spheres, views in (("dense", "kuhn19", 512, 8), ("object", "kuhn8", 20, 120)):
        sc = scenes.make_scene(kind, spheres)
        vid, faces = geometry.get_surface_vf(sc.tets)
        v = scenes.deform(sc, 0.02)[np.asarray(vid)]
        pos = torch.from_numpy(scenes.transform_pos(scenes.orbit_mvps(views), v)).cuda()
        tri = torch.from_numpy(np.asarray(faces, dtype=np.int32)).cuda()
        ctx = dr.RasterizeCudaContext()
        for _ in range(3):
            rast, _ = dr.rasterize(ctx, pos, tri, resolution=[512, 512], grad_db=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rast, _ = dr.rasterize(ctx, pos, tri, resolution=[512, 512], grad_db=False)
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 20
        out[name + "_checksum"] = float(rast[..., 3].double().sum())
    print("RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child()
    res = {v: [] for v in args.variants}
    for _ in range(args.rounds):
        for v in args.variants:
            env = dict(os.environ)
            if v != "base":
                env["TSSPLAT_AMD_LIB"] = os.path.join(ROOT, "tssplat_amd", f"libtssplat_amd_{v}.so")
            p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(v, "FAILED", p.stderr[-400:])
                continue
            res[v].append(json.loads(line[0][7:]))
    for v, rs in res.items():
        if rs:
            print(f"{v:12s} dense ms {min(r['dense'] for r in rs):.4f}  object ms {min(r['object'] for r in rs):.4f}  checksums {rs[0]['dense_checksum']:.0f} {rs[0]['object_checksum']:.0f}")


if __name__ == "__main__":
    main()
