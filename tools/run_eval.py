#!/usr/bin/env python3
"""A few raw C-ABI evaluations of one scene -- the command to put under rocprofv3 (kernel trace / PMC passes).

    python tools/run_eval.py [--scene kuhn19 --spheres 512 --evals 5] [--debug-shuffle N] [--operator scaled]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kuhn19")
    ap.add_argument("--spheres", type=int, default=512)
    ap.add_argument("--evals", type=int, default=5)
    ap.add_argument("--sigma", type=float, default=0.02)
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--debug-shuffle", type=int, default=0)
    ap.add_argument("--max-threads", type=int, default=0)
    ap.add_argument("--lds-budget", type=int, default=0)
    ap.add_argument("--spt", type=int, default=0)
    ap.add_argument("--rebuild-dminv", type=int, default=0)
    args = ap.parse_args()
    import torch
    from tssplat_amd import _capi, scenes, tet_spheres_ext as T
    lib = _capi.load()
    sc = scenes.make_scene(args.scene, args.spheres)
    ts = T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), debug_flags=args.debug_shuffle, max_threads=args.max_threads,
                      lds_budget_bytes=args.lds_budget, slots_per_thread=args.spt, rebuild_dminv=bool(args.rebuild_dminv))
    x = torch.from_numpy(scenes.deform(sc, args.sigma)).cuda()
    g = torch.empty_like(x)
    e = torch.empty((), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(args.evals):
        _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, 2e-4 / args.spheres, 2e-4, args.order, st,
                                               e.data_ptr(), g.data_ptr()))
    torch.cuda.synchronize()
    print(f"{args.spheres} x {args.scene}: E = {float(e):.6g}, plan {ts.plan_info()}")


if __name__ == "__main__":
    main()
