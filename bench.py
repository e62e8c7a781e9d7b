#!/usr/bin/env python3
"""Benchmark of the tet-sphere geometry-energy hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one energy forward + backward over one batch of synthetic
tet-spheres, through the reference's operator surface
(``SmoothnessBarrierEnergy`` -> ``SmoothnessBarrierFunc`` -> ``tet_spheres_ext``).
Metric (BASELINE.json): tetrahedra/sec, whole job, inputs resident in HBM.

Workload at N=1: the scene BASELINE.json quotes the metric on, the
512-sphere / ~21 M-tet scene (512 x kuhn_ball(19)); it fits one
GPU (1.6 GB of plan data).  N>1: weak scaling -- every rank owns its own 512
spheres (tet-spheres share no vertices, so the path shards with no data-path
collective; the only exchange is the all-reduce of the scalar energy, issued
every step).  ``--scaling strong`` instead splits the 512 spheres over ranks.

Rank 0 prints ONE JSON line; extra keys: ``roofline`` (tile kernel, HIP events
on the launch stream, algorithmic bytes 68 m + 24 n per evaluation) and, at
N=1, ``cpu_baseline`` (the reference formulation in plain PyTorch on the host
cores, on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--scene", default="kuhn19", help="kuhnK | cone (per-sphere template)")
    p.add_argument("--spheres", type=int, default=512, help="spheres per job (strong) / per rank (weak)")
    p.add_argument("--sigma", type=float, default=0.02, help="deformation noise, fraction of sphere radius")
    p.add_argument("--order", type=int, default=2)
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    p.add_argument("--max-threads", type=int, default=0)
    p.add_argument("--lds-budget", type=int, default=0)
    p.add_argument("--target-owned", type=int, default=0)
    p.add_argument("--spt", type=int, default=0, help="slots per thread (2 or 4; 0 = library default)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-spheres", type=int, default=8)
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (bring-up on a 1-GPU box)")
    p.add_argument("--all-ranks-on-device0", action="store_true",
                   help="bring-up only: every rank uses cuda:0 (needs --dist-backend gloo)")
    return p.parse_args()


def cpu_baseline(args, torch, scenes):
    """The reference *formulation* (fp32 sparse M = G^T L^T L G and G, five sparse products per
    forward+backward, tet_spheres_cuda.cu:118-263) in plain PyTorch on the host cores, on a
    bounded sample of the same workload.  Oracle-side code, timed as the baseline only."""
    from oracle import torch_energies as TE
    S = max(1, min(args.cpu_sample_spheres, args.spheres))
    sc = scenes.make_scene(args.scene, S, seed=0)
    t0 = time.time()
    ts = TE.TorchTetSpheres(sc.rest, sc.tets, layout="csr")
    build_s = time.time() - t0
    x = torch.from_numpy(scenes.deform(sc, args.sigma, seed=1))
    c1, c2 = 2e-4 / args.spheres, 2e-4

    def one():
        TE.compute_energy(x, ts, c1, c2, args.order)
        TE.compute_energy_backward(1.0, x, ts, c1, c2, args.order)

    # torch's sparse CSR kernels do not scale with threads; pick the best of a short sweep so the
    # baseline is not handicapped by oversubscription, then time that setting
    ncpu = os.cpu_count() or 1
    best_threads, best_rate = torch.get_num_threads(), 0.0
    for nt in sorted({1, 4, 8, 16, 32, ncpu} & set(range(1, ncpu + 1))):
        torch.set_num_threads(nt)
        one()
        t0 = time.time()
        for _ in range(3):
            one()
        rate = 3 / (time.time() - t0)
        if rate > best_rate:
            best_threads, best_rate = nt, rate
    torch.set_num_threads(best_threads)
    for _ in range(2):
        one()
    reps, t0 = 0, time.time()
    while True:
        one()
        reps += 1
        el = time.time() - t0
        if (reps >= 30 and el > 5.0) or el > 15.0:
            break
    return {
        "value": sc.n_tets * reps / el,
        "unit": "tets/s",
        "cores": int(best_threads),
        "kind": "port",
        "sample": f"{S} x {args.scene} spheres ({sc.n_tets} tets), {reps} fwd+bwd evaluations in {el:.1f} s, "
                  f"torch sparse CSR fp32 (reference formulation M=G'L'LG + G, tet_spheres_cuda.cu:118-263), best of a "
                  f"thread sweep on {ncpu} host cores, operator build {build_s:.1f} s untimed",
    }


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # the module prints "initializing" like the reference does
        from tssplat_amd import scenes
        from tssplat_amd.energies import SmoothnessBarrierEnergy

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tssplat_amd hot path has no CPU fallback")
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)

    # ---- the batch this rank owns ----
    if args.scaling == "weak":
        my_spheres = args.spheres
        seed = 1000 * rank
    else:
        lo = args.spheres * rank // world
        hi = args.spheres * (rank + 1) // world
        my_spheres = hi - lo
        seed = 1000 * rank
    total_spheres = my_spheres * world if args.scaling == "weak" else args.spheres
    t0 = time.time()
    sc = scenes.make_scene(args.scene, my_spheres, seed=seed)
    t_scene = time.time() - t0

    class Flags:
        smooth_eng_coeff = 2e-4 / total_spheres     # geometry/tetmesh_geometry.py:242-243
        barrier_coeff = 2e-4
        increase_order_iter = 1000 if args.order == 2 else -1

    t0 = time.time()
    energy = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags, max_threads=args.max_threads,
                                     lds_budget_bytes=args.lds_budget, target_owned=args.target_owned,
                                     slots_per_thread=args.spt)
    t_plan = time.time() - t0
    info = energy.tet_sp.plan_info()
    if rank == 0:
        log(f"scene {my_spheres} x {args.scene}: n={sc.n_vertices} m={sc.n_tets} ({t_scene:.1f} s); "
            f"plan {t_plan:.1f} s: {info}")
    x = torch.nn.Parameter(torch.from_numpy(scenes.deform(sc, args.sigma, seed=seed + 1)).to(dev))
    m_local, n_local = sc.n_tets, sc.n_vertices
    del sc

    it = 10
    c1, c2 = energy.coeff_scheduler(it)
    e_sum = torch.zeros((), device=dev)

    def step():
        x.grad = None
        e = energy(x, it, c1, c2)          # fused energy+gradient pass, finish kernel
        e.backward()                        # grad_output scale
        if world > 1:                       # the path's only exchange: the scalar energy
            dist.all_reduce(e.detach(), op=dist.ReduceOp.SUM, async_op=True)
        return e

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    e_val = float(e.detach())

    # ---- roofline leg: the tile kernel alone, HIP events on the launch stream ----
    energy.tet_sp.set_timing(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    tile_ms, finish_ms, n_eval = energy.tet_sp.get_timing()
    energy.tet_sp.set_timing(False)
    tile_ms /= max(n_eval, 1)
    finish_ms /= max(n_eval, 1)
    b_alg = 68.0 * m_local + 24.0 * n_local           # bytes per fused fwd+bwd evaluation
    achieved = b_alg / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
    traffic = None
    prof = os.path.join(ROOT, "profiles", "traffic.json")   # written from rocprofv3 --pmc passes, see profiles/README.md
    if os.path.exists(prof):
        try:
            rec = json.load(open(prof))
            key = f"{args.scene}x{my_spheres}"
            if key in rec:
                traffic = rec[key]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None

    # ---- raw C-ABI rate (no autograd / Python operator overhead between launches) ----
    from tssplat_amd import _capi
    lib = _capi.load()
    g = torch.empty_like(x)
    ebuf = torch.empty((), device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    h = energy.tet_sp._handle()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        _capi.check(lib.tsamd_forward_backward(h, x.data_ptr(), None, c1, c2, args.order, stream,
                                               ebuf.data_ptr(), g.data_ptr()))
    torch.cuda.synchronize(dev)
    raw_elapsed = time.perf_counter() - t1

    total_tets = m_local * world if args.scaling == "weak" else None
    if args.scaling == "strong":
        mt = torch.tensor([m_local], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(mt)
        total_tets = int(mt.item())
    value = total_tets * args.steps / elapsed

    out = None
    if rank == 0:
        out = {
            "metric": "tetrahedra/sec (energy fwd+bwd)",
            "value": value,
            "unit": "tets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{total_spheres} tet-spheres x {args.scene} ({total_tets} tets total, "
                            f"{m_local} tets / {n_local} vertices per GPU), sigma={args.sigma}, order={args.order}, "
                            f"energy+gradient through SmoothnessBarrierEnergy (autograd fwd+bwd)",
                "spheres_per_gpu": my_spheres,
                "tets_per_gpu": m_local,
                "vertices_per_gpu": n_local,
                "tiles_per_gpu": info["n_tiles"],
                "slots_per_tet": info["total_slots"] / max(m_local, 1),
                "block_threads": info["block_threads"],
                "lds_bytes": info["lds_bytes"],
                "parallelism": f"spheres sharded over {world} GPU(s), scalar energy all-reduce only",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "tile_energy_kernel<WITH_GRAD=true>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": b_alg,
                "kernel_ms": tile_ms,
                "finish_kernel_ms": finish_ms,
            },
            "raw_c_abi_tets_per_s": m_local * args.steps / raw_elapsed,
            "energy": e_val,
            "plan_build_s": t_plan,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, torch, scenes)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
