#!/usr/bin/env python3
"""Cost of the explicit-operator (L as data) path next to the built-in uniform operator: tile-kernel ms on one scene.

    python tools/bench_operator.py [--scene kuhn19 --spheres 64]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kuhn19")
    ap.add_argument("--spheres", type=int, default=64)
    ap.add_argument("--evals", type=int, default=200)
    args = ap.parse_args()
    import numpy as np
    import scipy.sparse as sp
    import torch
    from tssplat_amd import _capi, scenes, tet_spheres_ext as T
    lib = _capi.load()
    sc = scenes.make_scene(args.scene, args.spheres)
    x = torch.from_numpy(scenes.deform(sc, 0.02)).cuda()
    g = torch.empty_like(x)
    e = torch.empty((), device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def measure(ts, label):
        def run(n):
            for _ in range(n):
                _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, 2e-4 / args.spheres, 2e-4, 2, st,
                                                       e.data_ptr(), g.data_ptr()))
        run(300)
        torch.cuda.synchronize()
        ts.set_timing(True)
        run(args.evals)
        tile_ms, fin_ms, n = ts.get_timing()
        ts.set_timing(False)
        info = ts.plan_info()
        print(f"{label:34s} tile {tile_ms / n:.4f} ms  finish {fin_ms / n:.4f} ms  E {float(e):.8g}  planes {info['n_planes']}  "
              f"threads {info['block_threads']}  lds {info['lds_bytes']}  slots/tet {info['total_slots'] / info['n_tets']:.4f}")

    measure(T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1)), "built-in uniform operator")
    # the same operator handed in as data: L = D - A over face neighbours, built from the plan's own adjacency
    t0 = time.time()
    m = sc.tets.shape[0]
    faces = np.sort(np.stack([sc.tets[:, [1, 2, 3]], sc.tets[:, [0, 2, 3]], sc.tets[:, [0, 1, 3]], sc.tets[:, [0, 1, 2]]], axis=1), axis=2)
    key = (faces[..., 0].astype(np.int64) << 42) | (faces[..., 1].astype(np.int64) << 21) | faces[..., 2].astype(np.int64)
    flat = key.ravel()
    order = np.argsort(flat, kind="stable")
    same = flat[order][1:] == flat[order][:-1]
    a = order[:-1][same] // 4
    b = order[1:][same] // 4
    A = sp.coo_matrix((np.ones(2 * a.size), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(m, m)).tocsr()
    L = (sp.diags(np.asarray(A.sum(axis=1)).ravel()) - A).tocsr()
    print(f"operator built in {time.time() - t0:.1f} s, nnz {L.nnz}")
    measure(T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), operator=L), "explicit operator (same L as data)")
    # a NON-symmetric operator (the row-scaled umbrella D^-1 (D - A)): row and column weights differ, 22 planes per slot
    deg = np.asarray(A.sum(axis=1)).ravel()
    Ls = (sp.diags(1.0 / np.maximum(deg, 1.0)) @ L).tocsr()
    measure(T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), operator=Ls), "explicit, non-symmetric operator")
    for thr, lds in ((768, 81920), (640, 68000), (512, 54400), (448, 48000)):
        try:
            ts = T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), operator=L, max_threads=thr, lds_budget_bytes=lds)
        except RuntimeError as exc:        # a thread count the explicit-operator kernels of this build are not compiled for
            print(f"explicit, {thr} threads, {lds} B: {exc}")
            continue
        measure(ts, f"explicit, {thr} threads, {lds} B")
    measure(T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), max_threads=640, lds_budget_bytes=68000), "built-in, 640 threads / 68000 B")


if __name__ == "__main__":
    main()
