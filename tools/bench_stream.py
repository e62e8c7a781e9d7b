#!/usr/bin/env python3
"""Tile-kernel time of the STREAMING path (experimental) next to the blob kernel on one scene, HIP events of the library.

    python tools/bench_stream.py [--scene kuhn19 --spheres 512 --evals 100]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kuhn19")
    ap.add_argument("--spheres", type=int, default=512)
    ap.add_argument("--sigma", type=float, default=0.02)
    ap.add_argument("--evals", type=int, default=100)
    ap.add_argument("--skip-blob", action="store_true")
    args = ap.parse_args()
    import time
    import torch
    from tssplat_amd import scenes, tet_spheres_ext as T, _capi
    from tssplat_amd.stream import StreamTetSpheres
    sc = scenes.make_scene(args.scene, args.spheres)
    x = torch.from_numpy(scenes.deform(sc, args.sigma)).cuda()
    c1, c2 = 2e-4 / args.spheres, 2e-4
    t0 = time.time()
    st = StreamTetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    t_plan = time.time() - t0
    info = st.plan_info()
    for _ in range(30):
        e, g = st.forward_backward(x, c1, c2, 2)
    torch.cuda.synchronize()
    st.set_timing(True)
    for _ in range(args.evals):
        e, g = st.forward_backward(x, c1, c2, 2)
    tube_ms, fin_ms, n = st.get_timing()
    st.set_timing(False)
    b_alg = 68.0 * sc.n_tets + 24.0 * sc.n_vertices
    out = {"scene": f"{args.spheres} x {args.scene}", "tets": sc.n_tets, "stream": {
        "tube_kernel_ms": tube_ms / n, "finish_ms": fin_ms / n, "frac_of_8TBs": b_alg / (tube_ms / n * 1e-3) / 8e12, "energy": float(e),
        "plan_s": t_plan, "slots_per_tet": info["total_slots"] / sc.n_tets, "n_tubes": info["n_tubes"], "bands": info["total_bands"],
        "lane_fill": info["total_slots"] / (info["total_bands"] * info["band_slots"]), "blob_bytes_per_tet": info["blob_bytes"] / sc.n_tets,
        "lds_bytes": info["lds_bytes"], "shared_vertex_copies": info["shared_vertex_copies"]}}
    if not args.skip_blob:
        lib = _capi.load()
        ts = T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
        gb, eb = torch.empty_like(x), torch.empty((), device="cuda")
        stream = torch.cuda.current_stream().cuda_stream

        def run(k):
            for _ in range(k):
                _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, c1, c2, 2, stream, eb.data_ptr(), gb.data_ptr()))
        run(30)
        torch.cuda.synchronize()
        ts.set_timing(True)
        run(args.evals)
        tile_ms, fin2, n2 = ts.get_timing()
        out["blob"] = {"tile_kernel_ms": tile_ms / n2, "finish_ms": fin2 / n2, "frac_of_8TBs": b_alg / (tile_ms / n2 * 1e-3) / 8e12,
                       "energy": float(eb), "slots_per_tet": ts.plan_info()["total_slots"] / sc.n_tets}
        out["grad_rel_diff"] = float((gb - g).norm() / gb.norm())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
