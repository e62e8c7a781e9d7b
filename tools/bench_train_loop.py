#!/usr/bin/env python3
"""Iterations per second of ``energy(x, it, *coeff_scheduler(it)).backward(); optimizer.step()`` (trainer.py:130-133, loss =
the geometry energy) on the reference's own problem sizes, three ways on the same GPU:

  eager           SmoothnessBarrierEnergy + torch.autograd + AdamUniform.step          (what trainer.py's shape gives)
  graph-autograd  SmoothnessBarrierEnergy(graph=True): the evaluation replayed from a HIP graph behind an autograd node
  fused loop      FusedEnergyAdamLoop: n iterations per graph launch, nothing on the host in between

    python tools/bench_train_loop.py [--scene kuhn8 --spheres 64 --iters 512 --chunk 32]
"""
import argparse
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kuhn8")
    ap.add_argument("--spheres", type=int, default=64)
    ap.add_argument("--iters", type=int, default=512)
    ap.add_argument("--chunk", type=int, default=32)
    args = ap.parse_args()
    import torch
    from tssplat_amd import scenes
    from tssplat_amd.energies import FusedEnergyAdamLoop, SmoothnessBarrierEnergy
    from tssplat_amd.utils.optimizer import AdamUniform

    flags = types.SimpleNamespace(smooth_eng_coeff=2e-4, barrier_coeff=2e-4, increase_order_iter=1000)
    sc = scenes.make_scene(args.scene, args.spheres)
    x0 = torch.from_numpy(scenes.deform(sc, 0.05)).cuda()
    kw = dict(lr=0.2, grad_limit=True, grad_limit_values=[0.01, 0.01], grad_limit_iters=[1500])     # config/gso.yaml:37-41

    def timed(fn, n):
        fn(0, min(n, 64))                                          # warm-up
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(64, n)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def eager_like(graph):
        mod = SmoothnessBarrierEnergy(sc.rest, sc.tets, flags, graph=graph)
        x = torch.nn.Parameter(x0.clone())
        opt = AdamUniform([x], **kw)

        def run(first, n):
            for it in range(first, first + n):
                c1, c2 = mod.coeff_scheduler(it)
                opt.zero_grad()
                mod(x, it, c1, c2).backward()
                opt.step()
        return run

    def fused():
        mod = SmoothnessBarrierEnergy(sc.rest, sc.tets, flags)
        x = torch.nn.Parameter(x0.clone())
        opt = AdamUniform([x], **kw)
        loop = FusedEnergyAdamLoop(mod, x, opt, n_iters=args.chunk)

        def run(first, n):
            for it in range(first, first + n, args.chunk):
                loop.run(it)
        return run

    n = (args.iters // args.chunk) * args.chunk
    res = {"eager_ms": timed(eager_like(False), n), "graph_autograd_ms": timed(eager_like(True), n), "fused_loop_ms": timed(fused(), n)}
    print(json.dumps({
        "metric": "iterations/s (energy + backward + AdamUniform.step)", "unit": "it/s", "value": 1e3 / res["fused_loop_ms"],
        "ms_per_iteration": res, "iterations_per_graph_launch": args.chunk,
        "config": {"workload": f"{args.spheres} x {args.scene}: {sc.n_tets} tets, {sc.n_vertices} vertices", "data": "synthetic",
                   "optimizer": "AdamUniform lr 0.2 grad_limit 0.01 (config/gso.yaml:37-41)"},
    }))


if __name__ == "__main__":
    main()
