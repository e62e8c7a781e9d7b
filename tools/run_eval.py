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
    ap.add_argument("--rebuild-dminv", type=int, default=-1, help="-1 = the library's choice, 0 = stream, 1 = rebuild")
    ap.add_argument("--preheat-ms", type=float, default=0.0,
                    help="continuous evaluations for about this long before the counted ones, no host sync in between: the kernel-stats "
                         "table then shows the steady state (the bench's protocol), not the clock ramp of a cold device")
    args = ap.parse_args()
    import torch
    from tssplat_amd import _capi, scenes, tet_spheres_ext as T
    lib = _capi.load()
    sc = scenes.make_scene(args.scene, args.spheres)
    ts = T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), debug_flags=args.debug_shuffle, max_threads=args.max_threads,
                      lds_budget_bytes=args.lds_budget, slots_per_thread=args.spt, rebuild_dminv=None if args.rebuild_dminv < 0 else bool(args.rebuild_dminv))
    x = torch.from_numpy(scenes.deform(sc, args.sigma)).cuda()
    g = torch.empty_like(x)
    e = torch.empty((), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def one():
        _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, 2e-4 / args.spheres, 2e-4, args.order, st,
                                               e.data_ptr(), g.data_ptr()))
    n_heat = 0
    if args.preheat_ms > 0:
        import time
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one()
        torch.cuda.synchronize()
        n_heat = int(min(max(args.preheat_ms * 1e-3 / max(time.perf_counter() - t0, 1e-6), 20), 4000))
        for _ in range(n_heat):
            one()
    for _ in range(args.evals):
        _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, 2e-4 / args.spheres, 2e-4, args.order, st,
                                               e.data_ptr(), g.data_ptr()))
    torch.cuda.synchronize()
    print(f"{args.spheres} x {args.scene}: E = {float(e):.6g}, {n_heat} pre-heat + {args.evals} evaluations, plan {ts.plan_info()}")


if __name__ == "__main__":
    main()
