#!/usr/bin/env python3
"""A/B timing of library variants (tssplat_amd/_build.py::build_variant) on one scene, tile-kernel ms from the
library's own HIP events.  Every variant runs in its own process (one library per process), several rounds
interleaved so that clock / thermal drift hits all of them alike.

    python tools/ab_variants.py base keepF keepH0 --spheres 512 [--rounds 3] [--opts rebuild_dminv=1]

`base` is tssplat_amd/libtssplat_amd.so, any other name N is tssplat_amd/libtssplat_amd_N.so.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import torch
    from tssplat_amd import _capi, scenes, tet_spheres_ext as T
    lib = _capi.load()
    sc = scenes.make_scene(args.scene, args.spheres)
    kw = {}
    for kv in args.opts:
        k, v = kv.split("=")
        kw[k] = int(v)
    ts = T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), **kw)
    x = torch.from_numpy(scenes.deform(sc, args.sigma)).cuda()
    g = torch.empty_like(x)
    e = torch.empty((), device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run(n):
        for _ in range(n):
            _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, 2e-4 / args.spheres, 2e-4, args.order, st,
                                                   e.data_ptr(), g.data_ptr()))
    run(args.warm)
    torch.cuda.synchronize()
    out = []
    for _ in range(args.rounds):
        run(args.warm // 4 + 1)
        ts.set_timing(True)
        run(args.evals)
        tile_ms, fin_ms, n = ts.get_timing()
        ts.set_timing(False)
        out.append(tile_ms / n)
    print("RESULT " + json.dumps({"tile_ms": out, "finish_ms": fin_ms / n, "energy": float(e), "gsum": float(g.double().abs().sum()),
                                  "slots_per_tet": ts.plan_info()["total_slots"] / ts.plan_info()["n_tets"]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*")
    ap.add_argument("--scene", default="kuhn19")
    ap.add_argument("--spheres", type=int, default=512)
    ap.add_argument("--sigma", type=float, default=0.02)
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--evals", type=int, default=200)
    ap.add_argument("--warm", type=int, default=400)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--opts", nargs="*", default=[])
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    here = os.path.join(ROOT, "tssplat_amd")
    res = {v: [] for v in args.variants}
    info = {}
    for _ in range(args.passes):
        for v in args.variants:
            env = dict(os.environ)
            env["TSSPLAT_AMD_LIB"] = os.path.join(here, "libtssplat_amd.so" if v == "base" else f"libtssplat_amd_{v}.so")
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--scene", args.scene, "--spheres", str(args.spheres),
                   "--sigma", str(args.sigma), "--order", str(args.order), "--evals", str(args.evals), "--warm", str(args.warm),
                   "--rounds", str(args.rounds), "--opts", *args.opts]
            p = subprocess.run(cmd, env=env, capture_output=True, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(f"{v}: FAILED\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}")
                continue
            r = json.loads(line[0][7:])
            res[v] += r["tile_ms"]
            info[v] = r
    for v in args.variants:
        if res[v]:
            t = sorted(res[v])
            print(f"{v:12s} tile ms: min {t[0]:.4f}  median {t[len(t) // 2]:.4f}  max {t[-1]:.4f}   finish {info[v]['finish_ms']:.4f}  "
                  f"E {info[v]['energy']:.9g}  sum|g| {info[v]['gsum']:.9g}  slots/tet {info[v]['slots_per_tet']:.4f}")


if __name__ == "__main__":
    main()
