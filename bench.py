#!/usr/bin/env python3
"""Benchmark of the tet-sphere geometry-energy hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one energy forward + backward (energy and full gradient) over one batch of synthetic
tet-spheres.  Metric (BASELINE.json): tetrahedra/sec, whole job, inputs resident in HBM.

Workload: the scene BASELINE.json quotes the metric on, 512 tet-spheres x kuhn_ball(19) = 21 070 848 tets
(SURVEY.md 8(d)); it fits one GPU (1.5 GB of plan data).  N > 1 is STRONG scaling by default -- the same 512
spheres split over the ranks by whole spheres (north_star: >= 6x at 8 GPUs on this scene); `--scaling weak`
gives every rank its own 512.  Tet-spheres share no vertices, so there is no data-path collective; the path's
only exchange, the all-reduce of the scalar energy, is issued every step and its result is checked against the
sum of the rank energies.

Launch: one process per GPU.  Under `torchrun` / `python -m torch.distributed.run` the ranks come from the
environment; a bare `python bench.py --gpus N` with N > 1 spawns the N ranks itself (127.0.0.1 rendezvous).

How a step is issued (`--launch`): `graph` replays a HIP graph of the fused evaluation
(tssplat_amd.energies.GraphedSmoothnessBarrier: same kernels, no per-step Python/autograd work) -- 5x faster on
launch-bound batches (64-256 spheres of ~3 k tets: 16-31 us against 88 us per step); `eager` goes through
`SmoothnessBarrierEnergy` + `backward()` (torch.autograd) like the reference trainer -- 1.5 % faster on the
21 M-tet scene, where a replay cannot be queued behind its predecessor the way eager launches are.  `auto`
(default) times a few steps of each during setup and keeps the faster; both are reported at N = 1.

Setup ends with a pre-heat: ~40 back-to-back evaluations, outside every count, straight into the W warm-up steps.
After the idle seconds of plan building the GPU needs ~25 ms of continuous work to reach its steady clocks
(measured: 0.618 -> 0.565 -> 0.550 ms per step over the first three 20-step windows after idle); the driver's
`--warmup 5 --steps 20` would otherwise time the ramp.

Rank 0 prints ONE JSON line; extra keys: `roofline` (tile kernel, HIP events on the launch stream, algorithmic
bytes 68 m + 24 n per evaluation) and, at N = 1, `cpu_baseline` (the reference formulation in plain PyTorch on
the host cores, on a bounded sample).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--scene", default="kuhn19", help="kuhnK | cone | delaunayN | aveg (per-sphere template; aveg = the reference's a.veg, "
                   "952 of them make the headline's 21 M tets)")
    p.add_argument("--spheres", type=int, default=512, help="spheres per job (strong) / per rank (weak)")
    p.add_argument("--sigma", type=float, default=0.02, help="deformation noise, fraction of sphere radius")
    p.add_argument("--order", type=int, default=2)
    p.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                   help="auto = strong (the fixed 512-sphere scene split over the ranks)")
    p.add_argument("--launch", choices=["auto", "graph", "eager", "graph-autograd", "module"], default="auto",
                   help="auto = graph replay or eager autograd, whichever is faster on this batch (measured during setup); "
                        "graph-autograd = SmoothnessBarrierEnergy(graph=True) + backward(): the replay behind an autograd node; "
                        "module = ShardedSmoothnessBarrierEnergy(graph=True, exchange='overlap').forward + backward(): what a trainer "
                        "would call, the energy exchange issued by the module itself (one all-reduce per step, off the training thread)")
    p.add_argument("--max-threads", type=int, default=0)
    p.add_argument("--lds-budget", type=int, default=0)
    p.add_argument("--target-owned", type=int, default=0)
    p.add_argument("--spt", type=int, default=0, help="slots per thread (2 or 4; 0 = library default)")
    p.add_argument("--rebuild-dminv", type=int, default=-1, help="1 = rebuild Dm^-1 in registers from rest positions, 0 = stream it, -1 = library default")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-spheres", type=int, default=0,
                   help="spheres of the SAME scene (the first K, same seeds) the CPU baseline is timed on; 0 = the whole scene "
                        "up to ~2.7 M tets (64 x kuhn19, all of 64/256 x kuhn8)")
    p.add_argument("--cpu-threads", type=int, default=0, help="torch threads of the CPU baseline; 0 = min(16, host cores)")
    p.add_argument("--module-every", type=int, default=1,
                   help="--launch module: ShardedSmoothnessBarrierEnergy(every=N) -- one energy all-reduce per N steps (1 = every step)")
    p.add_argument("--force-collective", action="store_true",
                   help="N = 1 only: create a single-rank process group and issue the per-step energy exchange exactly as the "
                        "N > 1 path does, so that its host cost is inside the timed loop (scaling model, tools/scaling_model.py)")
    p.add_argument("--energy-window", type=int, default=16,
                   help="N > 1: the local energies of this many steps go into a ring of device slots and are all-reduced with ONE "
                        "collective per window (1 = one all-reduce per step)")
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (bring-up on a 1-GPU box)")
    p.add_argument("--all-ranks-on-device0", action="store_true",
                   help="bring-up only: every rank uses cuda:0 (needs --dist-backend gloo)")
    p.add_argument("--dry-run", action="store_true",
                   help="no GPU work: exercise rank launch, process group, energy all-reduce check and the JSON line "
                        "(CPU test of the N > 1 plumbing, gloo)")
    return p.parse_args(argv)


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, torch, scenes, sample, x_sample, c1, c2, total_spheres):
    """The reference *formulation* (fp32 sparse M = G^T L^T L G and G, five sparse products per
    forward+backward, tet_spheres_cuda.cu:118-263) in plain PyTorch on the host cores, on the SAME inputs as the GPU leg:
    the first K spheres of the same scene with the same deformed positions and coefficients (all of them where the scene
    is small enough).  Oracle-side code, timed as the baseline only."""
    from oracle import torch_energies as TE
    sc = sample
    t0 = time.time()
    ts = TE.TorchTetSpheres(sc.rest, sc.tets, layout="csr")
    build_s = time.time() - t0
    x = torch.from_numpy(x_sample)

    def one():
        TE.compute_energy(x, ts, c1, c2, args.order)
        TE.compute_energy_backward(1.0, x, ts, c1, c2, args.order)

    ncpu = os.cpu_count() or 1
    fixed = args.cpu_threads if args.cpu_threads > 0 else min(16, ncpu)
    # secondary: a short thread sweep (torch's sparse CSR kernels do not scale with threads)
    sweep = {}
    for nt in sorted({1, 4, 8, 16, 32, ncpu} & set(range(1, ncpu + 1))):
        torch.set_num_threads(nt)
        one()
        t0 = time.time()
        one()
        sweep[str(nt)] = sc.n_tets / (time.time() - t0)
    torch.set_num_threads(fixed)
    one()
    reps, t0 = 0, time.time()
    while True:
        one()
        reps += 1
        el = time.time() - t0
        if (reps >= 10 and el > 10.0) or el > 25.0:
            break
    # beside it: what the host can do with the SAME energy matrix-free -- the plain-C float64 oracle (oracle/c, OpenMP over
    # tets, rest inverses rebuilt per call), all host threads.  Not the reference's formulation; reported so that the GPU / CPU
    # ratio is not read against torch's sparse kernels alone.
    from oracle import c_oracle as CO
    nbr = CO.face_adjacency(sc.tets)
    CO.energy_and_grad(sc.rest, sc.tets, x_sample, c1, c2, args.order, nbr=nbr)
    c_reps, t0 = 0, time.time()
    while True:
        CO.energy_and_grad(sc.rest, sc.tets, x_sample, c1, c2, args.order, nbr=nbr)
        c_reps += 1
        c_el = time.time() - t0
        if (c_reps >= 5 and c_el > 3.0) or c_el > 8.0:
            break
    return {
        "value": sc.n_tets * reps / el,
        "unit": "tets/s",
        "cores": int(fixed),
        "kind": "port",
        "matrix_free_c_oracle": {"value": sc.n_tets * c_reps / c_el, "unit": "tets/s", "cores": int(os.environ.get("OMP_NUM_THREADS", ncpu)),
                                 "what": f"oracle/c/tet_energy_oracle.c, float64, OpenMP, same sample, {c_reps} fwd+bwd evaluations in {c_el:.1f} s"},
        "cpu_model": _cpu_model(),
        "host_cores": ncpu,
        "sample": f"spheres 0..{sc.n_spheres - 1} of the GPU leg's {total_spheres} x {args.scene} scene (seed 0; same rest mesh, same "
                  f"deformed positions sigma={args.sigma} seed 1, same c1/c2/order): {sc.n_tets} tets, {reps} fwd+bwd evaluations in "
                  f"{el:.1f} s at {fixed} torch threads, torch sparse CSR fp32 (reference formulation M=G'L'LG + G, "
                  f"tet_spheres_cuda.cu:118-263), operator build {build_s:.1f} s untimed",
        "thread_sweep_tets_per_s": sweep,
    }


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawned_rank(rank: int, argv: list, world: int, port: int) -> None:
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    run_rank(parse(argv))


def launch(argv=None) -> None:
    """Entry point: run this rank, or -- `--gpus N > 1` without a launcher's environment -- spawn the N ranks."""
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch.multiprocessing as mp
        mp.spawn(_spawned_rank, args=(list(sys.argv[1:] if argv is None else argv), args.gpus, _free_port()),
                 nprocs=args.gpus, join=True)
        return
    run_rank(args)


def run_rank(args) -> None:
    # stdout carries exactly one JSON line: everything else this process or its libraries print (the module's
    # "initializing", Gloo's connection banner, ...) goes to stderr, at file-descriptor level
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        _run_rank(args, stdout_fd)
    finally:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)


def _emit(stdout_fd: int, record: dict) -> None:
    os.write(stdout_fd, (json.dumps(record) + "\n").encode())


def _run_rank(args, stdout_fd: int) -> None:
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    scaling = "strong" if args.scaling == "auto" else args.scaling
    if world == 1:
        scaling = "weak" if args.scaling == "weak" else "strong"    # identical at one rank; keep the label honest
    use_gpu = not args.dry_run
    if use_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tssplat_amd hot path has no CPU fallback")
    if args.all_ranks_on_device0:
        local_rank = 0
    dev = torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # --force-collective at N = 1: a single-rank group, so that every step pays the host cost of the energy exchange
    use_coll = world > 1 or (args.force_collective and not args.dry_run)
    if use_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if args.dry_run else args.dist_backend
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    with contextlib.redirect_stdout(sys.stderr):     # the module prints "initializing" like the reference does
        from tssplat_amd import scenes
        from tssplat_amd.sharding import partition_spheres

    # ---- the spheres this rank owns (whole spheres, contiguous, balanced on tet count) ----
    v_range = None
    if scaling == "weak":
        my_spheres, total_spheres, seed = args.spheres, args.spheres * world, 1000 * rank
        sc = scenes.make_scene(args.scene, my_spheres, seed=seed) if use_gpu else None
    else:
        total_spheres = args.spheres
        if use_gpu:
            # every rank builds ONLY its own spheres (same radii / centres / noise as in the full scene: scenes.make_scene's
            # sphere_range, scenes.deform's vertex_range) -- eight ranks on one host do not each generate the 21 M-tet batch
            tv, tt = scenes.template_mesh(args.scene, seed=0)
            lo, hi = partition_spheres([int(tt.shape[0])] * total_spheres, world)[rank]
            sc = scenes.make_scene(args.scene, total_spheres, seed=0, sphere_range=(lo, hi))
            v_range = (lo * int(tv.shape[0]), hi * int(tv.shape[0]), total_spheres * int(tv.shape[0]))
            del tv, tt
        else:
            lo, hi = partition_spheres([1] * total_spheres, world)[rank]
            sc = None
        my_spheres, seed = hi - lo, 0
    c1_base = 2e-4 / total_spheres                   # geometry/tetmesh_geometry.py:242-243

    if args.dry_run:
        # plumbing only: a fake local energy, the real collective and check, fake timing
        e_local = torch.tensor([float(my_spheres)])
        # the same windowed exchange as the timed loop (tssplat_amd/sharding.py), a few fake steps
        from tssplat_amd.sharding import WindowedEnergyAllReduce
        reducer = WindowedEnergyAllReduce(max(1, args.energy_window), "cpu")
        for _ in range(max(args.steps, 1)):
            reducer.push(e_local)
        red_all = reducer.results()
        assert red_all.numel() == max(args.steps, 1) and bool((red_all == red_all[0]).all())
        red = red_all[-1:].clone()
        parts = [torch.zeros(1) for _ in range(world)]
        if world > 1:
            dist.all_gather(parts, e_local)
        else:
            parts = [e_local]
        assert abs(float(red) - sum(float(p) for p in parts)) <= 1e-6 * abs(float(red)), "all-reduced energy != sum of rank energies"
        assert abs(float(red) - total_spheres) < 1e-6
        if rank == 0:
            _emit(stdout_fd, {"metric": "tetrahedra/sec (energy fwd+bwd)", "value": 0.0, "unit": "tets/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True,
                              "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "dry-run",
                              "config": {"workload": "dry run (no GPU work)", "spheres_total": total_spheres,
                                         "energy_allreduce_checked": True}})
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    with contextlib.redirect_stdout(sys.stderr):
        from tssplat_amd.energies import GraphedSmoothnessBarrier, SmoothnessBarrierEnergy

    class Flags:
        smooth_eng_coeff = c1_base
        barrier_coeff = 2e-4
        increase_order_iter = 1000 if args.order == 2 else -1

    t0 = time.time()
    energy = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags, max_threads=args.max_threads,
                                     lds_budget_bytes=args.lds_budget, target_owned=args.target_owned,
                                     slots_per_thread=args.spt,
                                     rebuild_dminv=None if args.rebuild_dminv < 0 else bool(args.rebuild_dminv))
    t_plan = time.time() - t0
    info = energy.tet_sp.plan_info()
    if rank == 0:
        log(f"rank 0: {my_spheres} x {args.scene}: n={sc.n_vertices} m={sc.n_tets}; plan {t_plan:.1f} s: {info}")
    x_host = scenes.deform(sc, args.sigma, seed=seed + 1, vertex_range=v_range)
    x = torch.nn.Parameter(torch.from_numpy(x_host).to(dev))
    m_local, n_local = sc.n_tets, sc.n_vertices
    cpu_sample = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        # the CPU baseline runs on the same inputs: the first K spheres of this very scene, with these positions
        tets_per_sphere = max(m_local // max(my_spheres, 1), 1)
        k = args.cpu_sample_spheres if args.cpu_sample_spheres > 0 else max(1, min(my_spheres, 2_700_000 // tets_per_sphere))
        k = min(k, my_spheres)
        sub = sc.slice_spheres(0, k)
        cpu_sample = (sub, x_host[:sub.n_vertices].copy())
    del sc, x_host

    it0 = 10

    def coeffs(i):
        # the reference's schedule moves the coefficients every iteration (smooth_barrier.py:47-58): a timed step does too
        return energy.coeff_scheduler(it0 + i % 900)

    it = it0
    c1, c2 = energy.coeff_scheduler(it)
    from tssplat_amd import _capi
    lib = _capi.load()
    g_raw = torch.empty_like(x)
    e_raw = torch.empty((), device=dev)
    stream0 = torch.cuda.current_stream(dev).cuda_stream
    h = energy.tet_sp._handle()

    def raw():
        _capi.check(lib.tsamd_forward_backward(h, x.data_ptr(), None, c1, c2, args.order, stream0, e_raw.data_ptr(), g_raw.data_ptr()))

    def preheat(ms=120.0):
        """Continuous GPU work (no host sync inside) so that what follows starts at steady clocks."""
        raw()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        raw()
        torch.cuda.synchronize(dev)
        one = max(time.perf_counter() - t, 1e-6)
        for _ in range(int(min(max(ms * 1e-3 / one, 40), 8000))):
            raw()

    graphed = None
    if args.launch in ("auto", "graph", "graph-autograd", "module"):
        try:
            graphed = GraphedSmoothnessBarrier(energy, x)
            graphed.step(it)
            torch.cuda.synchronize(dev)
        except Exception as exc:                       # noqa: BLE001  (a runtime that cannot capture: run eagerly, say so)
            log(f"HIP graph capture failed ({exc!r}); using --launch eager")
            graphed = None

    if world > 1 and args.launch in ("auto", "graph", "graph-autograd", "module"):   # every rank must take the same path through the probes below
        ok = torch.tensor([1 if graphed is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            graphed = None

    from tssplat_amd.sharding import ShardedSmoothnessBarrierEnergy, WindowedEnergyAllReduce
    reducer = WindowedEnergyAllReduce(max(1, args.energy_window), dev) if use_coll and args.launch != "module" else None
    sharded = None
    if args.launch == "module":
        import numpy as np
        # The module a trainer would hold: this rank's spheres are all this process built, so the module sees them as ITS share
        # (rank 0 of 1 for the partition) while its energy exchange runs over the job's process group.
        energy.graph = graphed is not None
        sharded = ShardedSmoothnessBarrierEnergy(np.zeros((n_local, 3), np.float32), np.zeros((0, 4), np.int32), Flags, [0, n_local], [0, 0],
                                                 rank=0, world_size=1, local_factory=lambda v, f, F: energy, exchange="overlap",
                                                 every=max(1, args.module_every))

    def step_module(i):
        x.grad = None
        a, b = coeffs(i)
        e = sharded(x, it0 + i % 900, a, b)   # JobWideEnergy: the all-reduce is on its way, nobody waits for it here
        e.backward()
        return e

    def step_eager(i):
        x.grad = None
        a, b = coeffs(i)
        e = energy(x, it0 + i % 900, a, b)   # fused energy+gradient pass, finish kernel
        e.backward()                          # grad_output scale
        return e.detach()

    def step_graph(i):
        if reducer is not None:                          # the replay writes its energy into the exchange's ring slot itself
            e = graphed.step(it0 + i % 900, energy_copy=reducer.slot())[0]
            reducer.commit()
            return e
        return graphed.step(it0 + i % 900)[0]            # (coefficients refreshed on the device whenever they change: every step here)

    def step_graph_autograd(i):
        # code shaped like the reference trainer (trainer.py:94-130), replayed: SmoothnessBarrierEnergy(graph=True)
        x.grad = None
        a, b = coeffs(i)
        energy.graph = True
        try:
            e = energy(x, it0 + i % 900, a, b)
        finally:
            energy.graph = False
        e.backward()
        return e.detach()

    def step(fn, i):
        e = fn(i)
        if reducer is not None and fn is not step_graph:   # the path's only exchange: the scalar energy (never on the gradient's path)
            reducer.push(e)
        return e

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(fn, steps, warmup):
        for i in range(warmup):
            step(fn, i)
        if reducer is not None:
            reducer.results()               # (warm-up energies are not part of the check below)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            e = step(fn, warmup + i)
        if reducer is not None:
            reducer.flush()                 # the last (partial) window's collective is issued inside the timed region
        if sharded is not None and sharded._overlap is not None:
            sharded.flush_exchange()        # (every > 1: the last, partial window)
            sharded._overlap.drain()        # every collective issued and completed inside the timed region
        fence()
        el = time.perf_counter() - t0
        timed.local = el
        if world > 1:
            tmax = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, e

    launch_mode = args.launch if graphed is not None or args.launch == "module" else "eager"
    probe = {}
    if launch_mode == "auto":                          # a few steps of each, after a pre-heat; ranks agree on rank 0's choice
        for name, fn in (("graph", step_graph), ("eager", step_eager)):
            preheat()
            probe[name] = timed(fn, 30, 5)[0] / 30
            if reducer is not None:
                reducer.results()
        # (the faster of the two: the replay saves the autograd node and two launch gaps per step -- 60 % on a 64-sphere batch,
        # 3-4 % on the 21 M-tet scene)
        pick = torch.tensor([0 if probe["graph"] < probe["eager"] else 1], device=dev)
        if world > 1:
            dist.broadcast(pick, src=0)
        launch_mode = "graph" if int(pick.item()) == 0 else "eager"
    main_fn = {"graph": step_graph, "eager": step_eager, "graph-autograd": step_graph_autograd, "module": step_module}[launch_mode]
    preheat()
    elapsed, e_last = timed(main_fn, args.steps, args.warmup)
    elapsed_local = timed.local
    e_val = float(e_last)
    # the collective's result = sum of the rank energies (checked, not just issued)
    e_global = e_val
    if sharded is not None:
        # the module's own exchange: the value of the LAST step's JobWideEnergy (read now, long after the step) against the sum of
        # the rank-local energies of that step
        from tssplat_amd.sharding import JobWideEnergy
        assert isinstance(e_last, JobWideEnergy) and sharded._overlap.collectives >= (args.steps // max(1, args.module_every) if use_coll else 0)
        e_loc = e_last.as_subclass(torch.Tensor).detach().clone().reshape(1)
        parts = [torch.zeros(1, device=dev) for _ in range(world)]
        if world > 1:
            dist.all_gather(parts, e_loc)
        else:
            parts = [e_loc]
        e_sum = sum(float(p) for p in parts)
        assert abs(e_global - e_sum) <= 1e-5 * abs(e_sum) + 1e-30, f"job-wide energy {e_global} != sum of rank energies {e_sum}"
    if reducer is not None:
        reduced = reducer.results()
        assert reduced.numel() == args.steps, (reduced.numel(), args.steps)
        e_global = float(reduced[-1])
        parts = [torch.zeros(1, device=dev) for _ in range(world)]
        if world > 1:
            dist.all_gather(parts, e_last.clone().reshape(1))
        else:
            parts = [e_last.clone().reshape(1)]
        e_sum = sum(float(p) for p in parts)
        assert abs(e_global - e_sum) <= 1e-5 * abs(e_sum) + 1e-30, f"all-reduced energy {e_global} != sum of rank energies {e_sum}"

    others = {}
    if world == 1:                                     # the other launch modes, for the record
        for name, fn in (("eager_autograd", step_eager), ("graph_replay", step_graph), ("graph_autograd", step_graph_autograd)):
            if fn is main_fn or (graphed is None and fn is not step_eager) or sharded is not None:
                continue
            preheat()
            others[name + "_ms_per_step"] = 1e3 * timed(fn, args.steps, min(args.warmup, 5))[0] / args.steps
            if reducer is not None:
                reducer.results()

    # ---- roofline leg: the tile kernel alone, HIP events on the launch stream (raw C-ABI evaluations) ----
    preheat()
    torch.cuda.synchronize(dev)
    energy.tet_sp.set_timing(True)
    for _ in range(max(args.steps, 50)):               # (at least 50 launches: a 20-launch average moves by 3 % between runs)
        raw()
    torch.cuda.synchronize(dev)
    tile_ms, finish_ms, n_eval = energy.tet_sp.get_timing()
    energy.tet_sp.set_timing(False)
    tile_ms /= max(n_eval, 1)
    finish_ms /= max(n_eval, 1)
    b_alg = 68.0 * m_local + 24.0 * n_local           # bytes per fused fwd+bwd evaluation
    achieved = b_alg / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters cannot be collected inside this process (rocprofv3 wraps the command: separate
    # --pmc passes, tools/profile_round.sh); the figure of the last profiled run is reported only while the kernel and the plan
    # layout it was measured on are the ones running now (fingerprint of kernels.hip + plan.cpp + plan.h), otherwise null
    traffic, traffic_note = None, "no PMC pass on record for this workload (tools/profile_round.sh)"
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):
        try:
            from tssplat_amd import _build
            rec = json.load(open(prof)).get(f"{args.scene}x{my_spheres}")
            if rec is not None:
                if rec.get("sources") == _build.traffic_digest():
                    traffic, traffic_note = rec["hbm_bytes_per_launch"], f"rocprofv3 PMC passes of {rec['round']}, same kernel and plan sources"
                else:
                    traffic_note = (f"dropped: profiles/traffic.json ({rec.get('round')}) was measured on other kernel / plan sources "
                                    f"({rec.get('sources', 'unstamped')} != {_build.traffic_digest()}); re-run tools/profile_round.sh")
                    print("bench.py: " + traffic_note, file=sys.stderr)
        except Exception as exc:                       # noqa: BLE001
            traffic, traffic_note = None, f"profiles/traffic.json unreadable: {exc}"
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        raw()
    torch.cuda.synchronize(dev)
    raw_elapsed = time.perf_counter() - t1

    mt = torch.tensor([m_local], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(mt)
    total_tets = int(mt.item())
    value = total_tets * args.steps / elapsed
    # what every rank did: its own (un-maximised) step time, its tets, its kernel times -- imbalance shows here, not in `value`
    mine = torch.tensor([1e3 * elapsed_local / args.steps, float(m_local), tile_ms, finish_ms], device=dev, dtype=torch.float64)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    per_rank = [[float(v) for v in t.tolist()] for t in per_rank]

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
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{total_spheres} tet-spheres x {args.scene} ({total_tets} tets total, "
                            f"{m_local} tets / {n_local} vertices on rank 0), sigma={args.sigma}, order={args.order}, "
                            f"energy + full gradient per step",
                "launch": {"graph": "HIP-graph replay of the fused evaluation (GraphedSmoothnessBarrier.step)",
                           "eager": "eager: SmoothnessBarrierEnergy + backward() through torch.autograd",
                           "graph-autograd": "SmoothnessBarrierEnergy(graph=True) + backward(): HIP-graph replay behind an autograd node",
                           "module": "ShardedSmoothnessBarrierEnergy(graph=%s, exchange='overlap', every=%d).forward + backward()"
                                     % (graphed is not None, max(1, args.module_every))}[launch_mode],
                "schedule": "coefficients follow coeff_scheduler(it) and change every step (replays update them as kernel-node arguments, no device traffic)",
                "energy_exchange": (f"per-step local energies into a ring of {reducer.window} device slots, one all-reduce per window "
                                    f"({reducer.collectives} collectives so far, backend {dist.get_backend()}, {dist.get_world_size()} rank(s))"
                                    if reducer is not None else
                                    (f"the module's own: one all-reduce per {max(1, args.module_every)} step(s) issued by a helper thread on a side stream, the job-wide value waited for "
                                     f"when read ({sharded._overlap.collectives} collectives, "
                                     + (f"backend {dist.get_backend()}, {dist.get_world_size()} rank(s))" if use_coll else "no process group)"))
                                    if sharded is not None else "none (one rank, no process group)"),
                "spheres_rank0": my_spheres,
                "tets_rank0": m_local,
                "vertices_rank0": n_local,
                "tiles_rank0": info["n_tiles"],
                "slots_per_tet": info["total_slots"] / max(m_local, 1),
                "block_threads": info["block_threads"],
                "lds_bytes": info["lds_bytes"],
                "parallelism": f"whole spheres sharded over {world} GPU(s); one scalar-energy all-reduce per step, "
                               f"result checked against the sum of rank energies",
                "parity": "partial (L unpinned): the smoothness operator L is the ASSUMED uniform face-adjacency umbrella -- libpgo's "
                          "pgo_create_tet_biharmonic_gradient_matrix (tet_spheres.cpp:148) is not in the reference; G, det, cofactor "
                          "and the penalty are pinned to reference code; tools/pin_L_with_pypgo.py settles it where libpgo exists",
            },
            "per_rank": {"ms_per_step": [r[0] for r in per_rank], "tets": [int(r[1]) for r in per_rank],
                         "tile_kernel_ms": [r[2] for r in per_rank], "finish_kernel_ms": [r[3] for r in per_rank],
                         "ms_per_step_min": min(r[0] for r in per_rank), "ms_per_step_max": max(r[0] for r in per_rank)},
            "roofline": {
                "bound": "hbm",
                "kernel": "tile_energy_kernel<WITH_GRAD=true>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "algorithmic_bytes_per_launch": b_alg,
                "kernel_ms": tile_ms,
                "finish_kernel_ms": finish_ms,
                "launches": int(n_eval),               # evaluations behind kernel_ms (at least 50, whatever --steps says)
                # SURVEY 8(d) defines the roofline on t(fwd+bwd), i.e. on the whole step -- tile kernel + finish kernel +
                # whatever the launch path adds -- not on the dominant kernel alone: this rank's bytes / the timed step
                "achieved_step": b_alg / (elapsed / args.steps) / 1e9,
                "frac_step": b_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                "frac_kernels": b_alg / ((tile_ms + finish_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS if tile_ms > 0 else 0.0,
            },
            "raw_c_abi_tets_per_s": m_local * args.steps / raw_elapsed,
            "energy": e_global,
            "plan_build_s": t_plan,
        }
        out.update(others)
        if cpu_sample is not None:
            out["cpu_baseline"] = cpu_baseline(args, torch, scenes, cpu_sample[0], cpu_sample[1], c1, c2, total_spheres)
    if use_coll:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        _emit(stdout_fd, out)


if __name__ == "__main__":
    launch()
