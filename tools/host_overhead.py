#!/usr/bin/env python3
"""Host-side cost of one energy evaluation on a small scene (64 x kuhn8: ~12 us of GPU work), layer by layer:
direct operator calls, the autograd Function, the nn.Module.  Steady state on the MI355X box: 16 us direct,
~47 us through autograd (of which ~25 us is an empty autograd.Function round trip).  Note the first ~2000
autograd steps of a process run at about twice that (runtime warm-up), which is what short benchmarks see.

    python tools/host_overhead.py
"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from tssplat_amd import scenes, _capi, tet_spheres_ext as T
from tssplat_amd.energies import SmoothnessBarrierEnergy, SmoothnessBarrierFunc
class F: smooth_eng_coeff=2e-4/64; barrier_coeff=2e-4; increase_order_iter=1000
sc = scenes.make_scene("kuhn8", 64)
en = SmoothnessBarrierEnergy(sc.rest, sc.tets, F)
x = torch.nn.Parameter(torch.from_numpy(scenes.deform(sc, 0.02)).cuda())
ts = en.tet_sp
one = torch.ones((), device="cuda")
pc = time.perf_counter
def run(name, fn, N=2000):
    for i in range(100): fn(i)
    torch.cuda.synchronize()
    t = pc()
    for i in range(N): fn(i)
    h = pc() - t
    torch.cuda.synchronize()
    print(f"{name:52s} host {1e6 * h / N:6.1f} us  total {1e6 * (pc() - t) / N:6.1f} us")
run("ext.forward(x requires_grad) + ext.backward direct", lambda i: (T.forward(x, ts, 1e-4, 2e-4, 2), T.backward(one, x, ts, 1e-4, 2e-4, 2)))
run("ext.forward only (fused, cache overwritten)", lambda i: T.forward(x, ts, 1e-4, 2e-4, 2))
def f_apply(i):
    return SmoothnessBarrierFunc.apply(x, ts, 1e-4, 2e-4, 2)
run("Func.apply forward only (graph built, dropped)", f_apply)
def f_full(i):
    x.grad = None
    SmoothnessBarrierFunc.apply(x, ts, 1e-4, 2e-4, 2).backward()
run("Func.apply + backward()", f_full)
def f_full_keepgrad(i):
    SmoothnessBarrierFunc.apply(x, ts, 1e-4, 2e-4, 2).backward()
x.grad = None
run("Func.apply + backward(), grad accumulates", f_full_keepgrad)
g1 = torch.ones((), device="cuda")
def f_full_g(i):
    x.grad = None
    SmoothnessBarrierFunc.apply(x, ts, 1e-4, 2e-4, 2).backward(g1)
run("Func.apply + backward(given ones)", f_full_g)
def f_mod(i):
    x.grad = None
    en(x, i, 1e-4, 2e-4).backward()
run("module call + backward()", f_mod)
en_g = SmoothnessBarrierEnergy(sc.rest, sc.tets, F, graph=True)
def f_mod_graph(i):
    x.grad = None
    en_g(x, i, 1e-4, 2e-4).backward()
run("module(graph=True) call + backward()  [replay behind autograd]", f_mod_graph)
def f_mod_graph_sched(i):
    x.grad = None
    c1, c2 = en_g.coeff_scheduler(i % 900)
    en_g(x, i % 900, c1, c2).backward()
run("same, coefficients from the schedule (change every step)", f_mod_graph_sched)
from tssplat_amd.energies import GraphedSmoothnessBarrier
gr = GraphedSmoothnessBarrier(en, x)
run("GraphedSmoothnessBarrier.step(it) (no autograd)", lambda i: gr.step(i % 900))

# ---- the trainer's shape, `loss = a + energy(x, it, c1, c2); loss.backward()`, C++ nodes against Python nodes ----
w = torch.randn_like(x.detach())
def trainer_step(mod):
    def f(i):
        x.grad = None
        it = i % 900
        c1, c2 = mod.coeff_scheduler(it)
        loss = (w * x).sum() + mod(x, it, c1, c2)
        loss.backward()
    return f
for graph in (False, True):
    m_cpp = SmoothnessBarrierEnergy(sc.rest, sc.tets, F, graph=graph)
    m_py = SmoothnessBarrierEnergy(sc.rest, sc.tets, F, graph=graph)
    m_py._ext = None
    tag = "graph=True " if graph else "eager      "
    run(f"trainer-shaped step, {tag} C++ autograd node", trainer_step(m_cpp), N=4000)
    run(f"trainer-shaped step, {tag} Python autograd Function", trainer_step(m_py), N=4000)
def only_energy(mod):
    def f(i):
        x.grad = None
        it = i % 900
        c1, c2 = mod.coeff_scheduler(it)
        mod(x, it, c1, c2).backward()
    return f
m_cpp = SmoothnessBarrierEnergy(sc.rest, sc.tets, F, graph=True)
run("energy(x).backward() alone, graph=True, C++ node", only_energy(m_cpp), N=4000)

a_const = torch.tensor(0.37, device="cuda")
def with_const(mod):
    def f(i):
        x.grad = None
        it = i % 900
        c1, c2 = mod.coeff_scheduler(it)
        (a_const + mod(x, it, c1, c2)).backward()
    return f
for graph in (False, True):
    m_cpp = SmoothnessBarrierEnergy(sc.rest, sc.tets, F, graph=graph)
    run(f"loss = a + energy(x, it, c1, c2); loss.backward()   graph={graph}, C++ node", with_const(m_cpp), N=4000)


# ---- round 6: the sharded module with its own energy exchange, a single-rank RCCL group so that the helper thread
#      and its collectives are real ----
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29641")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from tssplat_amd.sharding import ShardedSmoothnessBarrierEnergy
vo = sc.sphere_vertex_offsets; to = np.arange(65) * (sc.n_tets // 64)
def module_step(mod):
    def f(i):
        x.grad = None
        it = i % 900
        c1, c2 = mod.coeff_scheduler(it)
        mod(x, it, c1, c2).backward()
    return f
for kw in (dict(graph=True), dict(graph=False), dict(graph=True, exchange="step"), dict(graph=True, exchange="window")):
    mod = ShardedSmoothnessBarrierEnergy(sc.rest, sc.tets, F, vo, to, **kw)
    run(f"ShardedSmoothnessBarrierEnergy({kw}) forward + backward()", module_step(mod), N=4000)
    if getattr(mod, "_overlap", None) is not None:
        mod._overlap.drain(); mod._overlap.close()
# the pieces of the default module's step, host time only
mod = ShardedSmoothnessBarrierEnergy(sc.rest, sc.tets, F, vo, to, graph=True)
module_step(mod)(0)
red, loc = mod._overlap, mod.local
def piece(name, fn, N=4000):
    for i in range(50): fn(i)
    t = pc()
    for i in range(N): fn(i)
    print(f"    piece: {name:60s} {1e6 * (pc() - t) / N:6.1f} us")
piece("coeff_scheduler", lambda i: mod.coeff_scheduler(i % 900))
piece("reducer.reserve + commit (event record, queue)", lambda i: red.commit(red.reserve()[0]))
_t, slot = red.reserve(); red.commit(_t)
piece("local.evaluate_direct (replay, energy into a slot)", lambda i: loc.evaluate_direct(x, i % 900, 1e-4, 2e-4, energy_copy=slot))
e = mod(x, 0, 1e-4, 2e-4)
from tssplat_amd.sharding import JobWideEnergy
piece("JobWideEnergy.wrap_direct", lambda i: JobWideEnergy.wrap_direct(loc._graphed.energy, red, 0, x, loc._graphed.grad, lambda: True))
def bw(i):
    x.grad = None
    JobWideEnergy.wrap_direct(loc._graphed.energy, red, 0, x, loc._graphed.grad, lambda: True).backward()
piece("x.grad = None; wrap_direct(...).backward()", bw)
piece("module.__call__ alone (forward, no backward)", lambda i: mod(x, i % 900, 1e-4, 2e-4))
torch.cuda.synchronize()
red.drain(); red.close()
dist.destroy_process_group()
