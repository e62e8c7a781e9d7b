#!/usr/bin/env python3
"""Where the sharded module's step differs from bench.py's replay loop, at the 8-way share of the headline scene (64 x kuhn19, one GPU,
single-rank RCCL group): the same loop built up piece by piece, microseconds per step INCLUDING the GPU (3 000 steps between two syncs).

    python tools/module_breakdown.py [--scene kuhn19 --spheres 64]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
ap = argparse.ArgumentParser(); ap.add_argument("--scene", default="kuhn19"); ap.add_argument("--spheres", type=int, default=64); ap.add_argument("--steps", type=int, default=3000)
args = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29653")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from tssplat_amd import scenes
from tssplat_amd.energies import SmoothnessBarrierEnergy, GraphedSmoothnessBarrier
from tssplat_amd.sharding import ShardedSmoothnessBarrierEnergy, WindowedEnergyAllReduce, OverlappedEnergyAllReduce, JobWideEnergy
S = args.spheres
class F: smooth_eng_coeff = 2e-4 / 512; barrier_coeff = 2e-4; increase_order_iter = 1000
sc = scenes.make_scene(args.scene, S)
en = SmoothnessBarrierEnergy(sc.rest, sc.tets, F, graph=True)
x = torch.nn.Parameter(torch.from_numpy(scenes.deform(sc, 0.02)).cuda())
gr = GraphedSmoothnessBarrier(en, x)
pc = time.perf_counter
def run(name, fn, N=args.steps):
    for i in range(300): fn(i)
    torch.cuda.synchronize(); t = pc()
    for i in range(N): fn(i)
    h = pc() - t; torch.cuda.synchronize()
    print(f"{name:100s} host {1e6 * h / N:6.1f} us   total {1e6 * (pc() - t) / N:6.1f} us", flush=True)
win = WindowedEnergyAllReduce(16, x.device)
def loop_bench(i):
    gr.step(10 + i % 900, energy_copy=win.slot()); win.commit()
run("bench.py's loop: replay, energy into the window slot, one all-reduce per 16 steps", loop_bench); win.results()
def loop_plain(i):
    gr.step(10 + i % 900)
run("replay alone (no exchange)", loop_plain)
def loop_fresh(i):
    x.grad = None
    c1, c2 = en.coeff_scheduler(10 + i % 900)
    e, g = gr.evaluate(c1, c2, 2, win.slot(), torch.empty_like(gr.x)); win.commit()
    x.grad = g
run("+ gradient into a fresh tensor that becomes x.grad (x.grad = None first)", loop_fresh); win.results()
ex = OverlappedEnergyAllReduce(x.device, dist.new_group(), 256, every=16)
def loop_exchange(i):
    x.grad = None
    c1, c2 = en.coeff_scheduler(10 + i % 900)
    t, slot = ex.reserve()
    e, g = gr.evaluate(c1, c2, 2, slot, torch.empty_like(gr.x)); ex.commit(t)
    x.grad = g
run("... with the C++ exchange (every=16) instead of the windowed reducer", loop_exchange); ex.flush(); ex.drain()
def loop_wrap(i):
    x.grad = None
    c1, c2 = en.coeff_scheduler(10 + i % 900)
    t, slot = ex.reserve()
    d = en.evaluate_direct(x, 10 + i % 900, c1, c2, energy_copy=slot); ex.commit(t)
    JobWideEnergy.wrap_direct(d[0], ex, t, x, d[1], d[2]).backward()
run("... through evaluate_direct + JobWideEnergy.wrap_direct(...).backward()", loop_wrap); ex.flush(); ex.drain(); ex.close()
vo = sc.sphere_vertex_offsets; to = np.arange(S + 1) * (sc.n_tets // S)
for every in (16, 1):
    mod = ShardedSmoothnessBarrierEnergy(np.zeros((sc.n_vertices, 3), np.float32), np.zeros((0, 4), np.int32), F, [0, sc.n_vertices], [0, 0], rank=0, world_size=1,
                                         local_factory=lambda v, f, FL: en, every=every)
    def loop_mod(i):
        x.grad = None
        c1, c2 = mod.coeff_scheduler(10 + i % 900)
        mod(x, 10 + i % 900, c1, c2).backward()
    run(f"ShardedSmoothnessBarrierEnergy(graph=True, every={every}) forward + backward()", loop_mod)
    mod.flush_exchange(); mod._overlap.drain(); mod._overlap.close()
dist.destroy_process_group()
