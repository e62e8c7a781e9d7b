"""world_size-2 `gloo` test of the N>1 path on CPU: sphere partitioning, rank-local slices, the
scalar energy all-reduce and the rank-local gradient.

The rank-local evaluator on a GPU box is the HIP-backed SmoothnessBarrierEnergy; here it is an
oracle-backed stand-in with the same interface (this is a TEST -- the product never imports the
oracle), so what is exercised is exactly the host logic of tssplat_amd/sharding.py.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tssplat_amd import scenes
from tssplat_amd.sharding import ShardedSmoothnessBarrierEnergy, partition_spheres


class _Flags:
    smooth_eng_coeff = 2e-4 / 5
    barrier_coeff = 2e-4
    increase_order_iter = 1000


class _OracleFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cache, c1, c2, order):
        from oracle import tet_energy_oracle as O
        E, _, _, g = O.energy_and_grad(x.detach().numpy(), cache, c1, c2, order)
        ctx.g = torch.from_numpy(g.astype(np.float32))
        return torch.tensor(E, dtype=torch.float32)

    @staticmethod
    def backward(ctx, go):
        return ctx.g * go, None, None, None, None


class _OracleEnergy(torch.nn.Module):
    """Stand-in for SmoothnessBarrierEnergy (same constructor / forward signature)."""

    def __init__(self, tet_v, tet_f, FLAGS):
        super().__init__()
        from oracle import tet_energy_oracle as O
        self.cache = O.prepare(np.asarray(tet_v, np.float32), np.asarray(tet_f, np.int32))
        self.FLAGS = FLAGS

    def coeff_scheduler(self, it):
        return self.FLAGS.smooth_eng_coeff, self.FLAGS.barrier_coeff

    def forward(self, x, it, c1, c2):
        return _OracleFunc.apply(x, self.cache, c1, c2, 4 if it > self.FLAGS.increase_order_iter else 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene():
    # five spheres of different sizes so that the balanced cut is not the trivial one
    parts = [scenes.kuhn_ball(k) for k in (3, 2, 4, 2, 3)]
    rest, tets, vo, to = [], [], [0], [0]
    rng = np.random.default_rng(0)
    for v, t in parts:
        rest.append((v * rng.uniform(0.1, 0.3) + rng.uniform(-0.5, 0.5, 3)).astype(np.float32))
        tets.append(t + vo[-1])
        vo.append(vo[-1] + v.shape[0])
        to.append(to[-1] + t.shape[0])
    rest = np.concatenate(rest)
    tets = np.concatenate(tets).astype(np.int32)
    x = (rest + 0.05 * rng.standard_normal(rest.shape)).astype(np.float32)
    return rest, tets, np.array(vo), np.array(to), x


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rest, tets, vo, to, x = _scene()
        mod = ShardedSmoothnessBarrierEnergy(rest, tets, _Flags, vo, to, local_factory=_OracleEnergy, exchange="step")
        lo, hi = mod.vertex_range
        xl = torch.from_numpy(x[lo:hi]).requires_grad_(True)
        c1, c2 = mod.coeff_scheduler(0)
        e = mod(xl, 0, c1, c2)
        (2.0 * e).backward()
        # replicated parameter: same energy, full gradient on every rank (one all-gather in backward)
        xf = torch.from_numpy(x).requires_grad_(True)
        ef = mod.forward_replicated(xf, 0, c1, c2)
        (2.0 * ef).backward()
        out[rank] = (float(e), mod.sphere_range, (lo, hi), xl.grad.numpy().copy(), float(ef), xf.grad.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_partition_properties():
    counts = [3072] * 64
    for w in (1, 2, 4, 8):
        r = partition_spheres(counts, w)
        assert r[0][0] == 0 and r[-1][1] == 64 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sizes = [hi - lo for lo, hi in r]
        assert max(sizes) - min(sizes) <= 1
    r = partition_spheres([100, 1, 1, 1, 1, 1], 3)
    assert r[0] == (0, 1) and r[-1][1] == 6 and all(hi > lo for lo, hi in r)
    r = partition_spheres([5, 5], 4)                       # more ranks than spheres
    assert sum(hi - lo for lo, hi in r) == 2 and all(hi - lo <= 1 for lo, hi in r)


def test_two_ranks_match_unsharded_oracle():
    from oracle import tet_energy_oracle as O
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    rest, tets, vo, to, x = _scene()
    cache = O.prepare(rest, tets)
    E, _, _, g = O.energy_and_grad(x, cache, _Flags.smooth_eng_coeff, _Flags.barrier_coeff, 2, grad_output=2.0)
    covered = np.zeros(rest.shape[0], dtype=bool)
    for rank in range(world):
        e, srange, (lo, hi), grad, ef, grad_full = out[rank]
        assert abs(ef - E) <= 2e-6 * abs(E) and np.abs(grad_full - g).max() <= 1e-5 * np.abs(g).max()
        assert abs(e - E) <= 2e-6 * abs(E)                 # every rank sees the job-wide energy
        assert np.abs(grad - g[lo:hi]).max() <= 1e-5 * np.abs(g).max()   # and only its own gradient slice
        assert not covered[lo:hi].any()
        covered[lo:hi] = True
    assert covered.all()
    assert out[0][1][1] == out[1][1][0]                    # contiguous sphere ranges


def test_single_process_is_identity():
    rest, tets, vo, to, x = _scene()
    mod = ShardedSmoothnessBarrierEnergy(rest, tets, _Flags, vo, to, local_factory=_OracleEnergy)
    assert mod.exchange == "overlap"                       # the job-wide value by default (waited for when read); windowed = opt-in
    assert mod.world_size == 1 and mod.vertex_range == (0, rest.shape[0])
    xl = torch.from_numpy(x).requires_grad_(True)
    e = mod(xl, 0, _Flags.smooth_eng_coeff, _Flags.barrier_coeff)
    e.backward()
    from oracle import tet_energy_oracle as O
    E, _, _, g = O.energy_and_grad(x, O.prepare(rest, tets), _Flags.smooth_eng_coeff, _Flags.barrier_coeff, 2)
    assert abs(float(e) - E) <= 2e-6 * abs(E)
    assert np.abs(xl.grad.numpy() - g).max() <= 1e-5 * np.abs(g).max()
    xf = torch.from_numpy(x).requires_grad_(True)
    mod.forward_replicated(xf, 0, _Flags.smooth_eng_coeff, _Flags.barrier_coeff).backward()
    assert np.abs(xf.grad.numpy() - g).max() <= 1e-5 * np.abs(g).max()


def _overlap_worker(rank, world, port, out):
    """exchange="overlap" (the default): a collective per call, issued by the helper thread; the value waits, backward() does not."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tssplat_amd.sharding import JobWideEnergy
        rest, tets, vo, to, x = _scene()
        mod = ShardedSmoothnessBarrierEnergy(rest, tets, _Flags, vo, to, local_factory=_OracleEnergy, depth=4)
        assert mod.exchange == "overlap"
        lo, hi = mod.vertex_range
        c1, c2 = mod.coeff_scheduler(0)
        rec = {}
        # (1) a training-shaped loop that never reads the value: six steps with different positions
        grads, kept = [], []
        for k in range(6):
            xl = torch.from_numpy(x[lo:hi] * (1.0 + 0.01 * k)).requires_grad_(True)
            e = mod(xl, 0, c1, c2)
            assert isinstance(e, JobWideEnergy) and e._tsamd_resolved is None
            e.backward()                                       # rank-local gradient; must not resolve the value
            assert e._tsamd_resolved is None
            grads.append(xl.grad.numpy().copy())
            kept.append(e)
        rec["grads"] = grads
        # (2) the values of the last `depth` calls are readable afterwards, older ones have expired (loudly)
        rec["late"] = [float(e) for e in kept[2:]]
        try:
            float(kept[0])
            rec["expired"] = False
        except RuntimeError as exc:
            rec["expired"] = "expired" in str(exc)
        # (3) reading inside the step -- the reference trainer's `loss = image_loss + energy` (trainer.py:115) -- gives the job-wide
        #     value AND the rank-local gradient, scaled
        xl = torch.from_numpy(x[lo:hi]).requires_grad_(True)
        e = mod(xl, 0, c1, c2)
        loss = 3.0 + 2.0 * e
        loss.backward()
        rec["loss"], rec["g_loss"] = float(loss), xl.grad.numpy().copy()
        # (4) under no_grad the call is still a collective and returns the plain job-wide value
        with torch.no_grad():
            en = mod(torch.from_numpy(x[lo:hi]), 0, c1, c2)
        assert type(en) is torch.Tensor
        rec["nograd"] = float(en)
        # (5) replicated parameter through the same exchange
        xf = torch.from_numpy(x).requires_grad_(True)
        ef = mod.forward_replicated(xf, 0, c1, c2)
        ef.backward()
        rec["ef"], rec["g_full"] = float(ef), xf.grad.numpy().copy()
        rec["collectives"] = mod._overlap.collectives
        rec["range"] = (lo, hi)
        rec["cxx"] = mod._overlap._cxx is not None             # the helper thread of csrc/torch_exchange.cpp (the extension builds on CPU)
        mod._overlap.drain()
        mod._overlap.close()
        # the Python helper thread (the fallback without the extension): same protocol, same values, on a group of its own
        from tssplat_amd.sharding import OverlappedEnergyAllReduce
        red = OverlappedEnergyAllReduce("cpu", dist.new_group(), depth=4, use_extension=False)
        assert red._cxx is None
        tickets = [red.submit(torch.tensor(float(rank + 1) * (k + 1))) for k in range(6)]
        rec["py_values"] = [float(red.value(t)) for t in tickets[2:]]
        try:
            red.value(tickets[0])
            rec["py_expired"] = False
        except RuntimeError:
            rec["py_expired"] = True
        red.drain()
        red.close()
        out[rank] = rec
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_overlapped_exchange_job_wide_values_and_local_gradients(world):
    from oracle import tet_energy_oracle as O
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, port, out), nprocs=world, join=True)
    rest, tets, vo, to, x = _scene()
    cache = O.prepare(rest, tets)
    c1, c2 = _Flags.smooth_eng_coeff, _Flags.barrier_coeff
    Es, gs = zip(*[(lambda r: (r[0], r[3]))(O.energy_and_grad((x * (1.0 + 0.01 * k)).astype(np.float32), cache, c1, c2, 2)) for k in range(6)])
    E0, _, _, g0 = O.energy_and_grad(x, cache, c1, c2, 2)
    for rank in range(world):
        rec = out[rank]
        lo, hi = rec["range"]
        for k in range(6):
            assert np.abs(rec["grads"][k] - gs[k][lo:hi]).max(initial=0.0) <= 1e-5 * np.abs(gs[k]).max() + 1e-12   # (ranks beyond the spheres own nothing)
        assert rec["expired"] is True
        for k, v in zip(range(2, 6), rec["late"]):
            assert abs(v - Es[k]) <= 3e-6 * abs(Es[k]), (rank, k, v, Es[k])            # job-wide, on every rank, steps later
        assert abs(rec["loss"] - (3.0 + 2.0 * E0)) <= 1e-5 * abs(3.0 + 2.0 * E0)
        assert np.abs(rec["g_loss"] - 2.0 * g0[lo:hi]).max(initial=0.0) <= 2e-5 * np.abs(g0).max() + 1e-12
        assert abs(rec["nograd"] - E0) <= 3e-6 * abs(E0) and abs(rec["ef"] - E0) <= 3e-6 * abs(E0)
        assert np.abs(rec["g_full"] - g0).max() <= 1e-5 * np.abs(g0).max()
        assert rec["collectives"] == 9                                                   # one per call, read or not
        assert rec["cxx"] is True
        tri = world * (world + 1) / 2
        assert rec["py_values"] == [tri * (k + 1) for k in range(2, 6)] and rec["py_expired"] is True


def test_overlapped_exchange_single_process():
    """No process group: the exchange is the identity, the lazy tensor behaves like the local energy."""
    from tssplat_amd.sharding import JobWideEnergy
    rest, tets, vo, to, x = _scene()
    mod = ShardedSmoothnessBarrierEnergy(rest, tets, _Flags, vo, to, local_factory=_OracleEnergy)
    xl = torch.from_numpy(x).requires_grad_(True)
    e = mod(xl, 0, _Flags.smooth_eng_coeff, _Flags.barrier_coeff)
    assert isinstance(e, JobWideEnergy) and e.shape == () and e.requires_grad and e.dtype == torch.float32
    assert e._tsamd_resolved is None                                                     # metadata reads do not resolve
    torch.autograd.backward((e, (xl * 0).sum()))                                         # trainer-shaped: two losses, no value read
    assert e._tsamd_resolved is None and xl.grad is not None
    assert "%.3e" % e == "%.3e" % float(e.detach()) and type(e + 1.0) is torch.Tensor   # %-formatting (trainer.py:118-125), arithmetic
    assert mod._overlap.collectives == 0


class _OracleEnergyDirect(_OracleEnergy):
    """The stand-in with the engine-free protocol of SmoothnessBarrierEnergy(graph=True).evaluate_direct: static buffers, a validity
    callable, the energy written into the exchange's slot as well."""

    def __init__(self, tet_v, tet_f, FLAGS):
        super().__init__(tet_v, tet_f, FLAGS)
        self.ticket, self.e_buf, self.g_buf, self.direct_calls = 0, torch.zeros(()), None, 0

    def evaluate_direct(self, x, it, c1, c2, energy_copy=None):
        from oracle import tet_energy_oracle as O
        E, _, _, g = O.energy_and_grad(x.detach().numpy(), self.cache, c1, c2, 4 if it > self.FLAGS.increase_order_iter else 2)
        if self.g_buf is None:
            self.g_buf = torch.zeros_like(x.detach())
        self.e_buf.fill_(float(E))
        self.g_buf.copy_(torch.from_numpy(g.astype(np.float32)))
        if energy_copy is not None:
            energy_copy.fill_(float(E))
        self.ticket += 1
        self.direct_calls += 1
        t = self.ticket
        return self.e_buf, self.g_buf.clone(), (lambda: self.ticket == t)   # (a fresh gradient tensor per evaluation, like the replay's)


def _direct_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tssplat_amd.sharding import JobWideEnergy
        rest, tets, vo, to, x = _scene()
        mod = ShardedSmoothnessBarrierEnergy(rest, tets, _Flags, vo, to, local_factory=_OracleEnergyDirect)
        lo, hi = mod.vertex_range
        c1, c2 = mod.coeff_scheduler(0)
        rec = {}
        xl = torch.nn.Parameter(torch.from_numpy(x[lo:hi].copy()))
        e = mod(xl, 0, c1, c2)
        assert isinstance(e, JobWideEnergy) and not e.requires_grad and e._tsamd_direct is not None
        e.backward()                                           # engine-free: the evaluation's gradient tensor goes to x.grad
        assert e._tsamd_resolved is None
        g1 = xl.grad.numpy().copy()
        try:
            e.backward()                                       # the graph is gone (as after a backward() without retain_graph)
            rec["twice"] = False
        except RuntimeError as exc:
            rec["twice"] = "already" in str(exc)
        ea = mod(xl, 0, c1, c2)
        ea.backward()                                          # x.grad is set: a further evaluation accumulates, as autograd would
        rec["g1"], rec["g2"] = g1, xl.grad.numpy().copy()
        rec["value"] = float(e)                                # read late: job-wide
        # trainer.py:115 shape: the value takes part in a loss -> attached through an ordinary node, scaled gradient
        xl.grad = None
        e2 = mod(xl, 0, c1, c2)
        (3.0 + 2.0 * e2).backward()
        rec["g_loss"] = xl.grad.numpy().copy()
        # a newer evaluation invalidates the older one's gradient, loudly
        e3 = mod(xl, 0, c1, c2)
        _ = mod(xl, 0, c1, c2)
        try:
            e3.backward()
            rec["stale"] = False
        except RuntimeError as exc:
            rec["stale"] = "overwritten" in str(exc)
        # a parameter with a hook is none of the fast path's business: the ordinary autograd path, same numbers
        calls = mod.local.direct_calls if mod.local is not None else 0
        xh = torch.nn.Parameter(torch.from_numpy(x[lo:hi].copy()))
        seen = []
        xh.register_hook(lambda g: seen.append(1))
        eh = mod(xh, 0, c1, c2)
        assert eh.requires_grad and (mod.local is None or mod.local.direct_calls == calls)
        eh.backward()
        rec["g_hook"], rec["hook_ran"] = xh.grad.numpy().copy(), len(seen) == 1
        rec["range"] = (lo, hi)
        out[rank] = rec
        mod._overlap.drain()
        mod._overlap.close()
    finally:
        dist.destroy_process_group()


def test_engine_free_backward_of_the_overlapped_module():
    from oracle import tet_energy_oracle as O
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_direct_worker, args=(world, port, out), nprocs=world, join=True)
    rest, tets, vo, to, x = _scene()
    E0, _, _, g0 = O.energy_and_grad(x, O.prepare(rest, tets), _Flags.smooth_eng_coeff, _Flags.barrier_coeff, 2)
    tol = 1e-5 * np.abs(g0).max()
    for rank in range(world):
        rec = out[rank]
        lo, hi = rec["range"]
        assert np.abs(rec["g1"] - g0[lo:hi]).max() <= tol and np.abs(rec["g2"] - 2.0 * g0[lo:hi]).max() <= 2 * tol
        assert abs(rec["value"] - E0) <= 3e-6 * abs(E0)
        assert np.abs(rec["g_loss"] - 2.0 * g0[lo:hi]).max() <= 2 * tol
        assert rec["stale"] is True and rec["hook_ran"] is True and rec["twice"] is True
        assert np.abs(rec["g_hook"] - g0[lo:hi]).max() <= tol


def _every_worker(rank, world, port, out):
    """exchange="overlap" with every=4: one collective per four calls; values readable from the end of their window on."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rest, tets, vo, to, x = _scene()
        rec = {}
        for use_ext in (True, False):
            mod = ShardedSmoothnessBarrierEnergy(rest, tets, _Flags, vo, to, local_factory=_OracleEnergy, depth=8, every=4)
            lo, hi = mod.vertex_range
            c1, c2 = mod.coeff_scheduler(0)
            if not use_ext:                                    # the Python helper thread: same protocol
                from tssplat_amd.sharding import OverlappedEnergyAllReduce
                mod._overlap = OverlappedEnergyAllReduce("cpu", mod._energy_group, 8, use_extension=False, every=4)
            kept = []
            for k in range(6):
                xl = torch.from_numpy(x[lo:hi] * (1.0 + 0.01 * k)).requires_grad_(True)
                e = mod(xl, 0, c1, c2)
                e.backward()
                kept.append(e)
                if k == 1:
                    try:
                        float(e)                               # its window (calls 0-3) has not gone out yet: loud, no deadlock
                        early = False
                    except RuntimeError as exc:
                        early = "not on its way" in str(exc)
            vals = [float(e) for e in kept[:4]]                # window 0 went out with call 3
            coll_before = mod._overlap.collectives
            mod.flush_exchange()                               # calls 4, 5: a partial window, sent on request (every rank)
            vals += [float(e) for e in kept[4:]]
            mod._overlap.drain()
            rec[use_ext] = dict(early=early, vals=vals, coll=(coll_before, mod._overlap.collectives), cxx=mod._overlap._cxx is not None)
            mod._overlap.close()
        out[rank] = rec
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_one_collective_per_window():
    from oracle import tet_energy_oracle as O
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_every_worker, args=(world, port, out), nprocs=world, join=True)
    rest, tets, vo, to, x = _scene()
    cache = O.prepare(rest, tets)
    Es = [O.energy_and_grad((x * (1.0 + 0.01 * k)).astype(np.float32), cache, _Flags.smooth_eng_coeff, _Flags.barrier_coeff, 2)[0] for k in range(6)]
    for rank in range(world):
        for use_ext in (True, False):
            r = out[rank][use_ext]
            assert r["early"] is True and r["cxx"] is use_ext
            assert r["coll"] == (1, 2)                                                   # six calls: one full window + the flushed rest
            assert all(abs(v - E) <= 3e-6 * abs(E) for v, E in zip(r["vals"], Es)), (rank, use_ext, r["vals"], Es)


def test_windowed_reducer_slots_filled_by_the_evaluation():
    """slot() / commit(): the evaluation writes its energy into the ring itself (a replay's second energy destination) -- same values and
    window boundaries as push()."""
    from tssplat_amd.sharding import WindowedEnergyAllReduce
    a, b = WindowedEnergyAllReduce(4, "cpu"), WindowedEnergyAllReduce(4, "cpu")
    for k in range(10):
        a.push(torch.tensor(float(k) * 1.5))
        b.slot().fill_(float(k) * 1.5)
        b.commit()
    assert torch.equal(a.results(), b.results()) and a.results().numel() == 0


def test_rejects_spheres_that_share_vertices():
    rest, tets, vo, to, _ = _scene()
    bad = tets.copy()
    bad[0, 0] = vo[-1] - 1        # first sphere now references a vertex of the last one
    with pytest.raises(ValueError):
        ShardedSmoothnessBarrierEnergy(rest, bad, _Flags, vo, to, rank=0, world_size=2, local_factory=_OracleEnergy)


def test_bench_spawns_its_own_ranks(capfd):
    """`python bench.py --gpus 2` with no launcher environment spawns the two ranks itself; `--dry-run` runs the rank
    launch, the gloo process group, the per-step energy all-reduce with its sum-of-ranks check and the JSON line
    without touching a GPU (the driver launches N > 1 through torch.distributed.run; this is the bare-python route)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--spheres", "7",
                          "--steps", "3", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                       # exactly one JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["config"]["energy_allreduce_checked"] and rec["config"]["spheres_total"] == 7
    # under a launcher's environment the same file runs as ONE rank of the job (no nested spawn)
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run", "--spheres", "3"],
                         capture_output=True, text=True, env=env1, timeout=300)
    assert out.returncode == 0 and json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1


def test_scene_slices_are_self_contained():
    sc = scenes.make_scene("kuhn3", 5)
    part = sc.slice_spheres(1, 4)
    assert part.n_spheres == 3 and part.tets.min() == 0 and part.tets.max() == part.n_vertices - 1
    assert np.array_equal(part.rest, sc.rest[sc.sphere_vertex_offsets[1]:sc.sphere_vertex_offsets[4]])


def _window_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tssplat_amd.sharding import WindowedEnergyAllReduce
        red = WindowedEnergyAllReduce(8, "cpu")
        for i in range(37):                                   # four full windows + a partial one
            red.push(torch.tensor(float((rank + 1) * 1000 + i)))
        first = red.results()
        for i in range(3):                                    # reusable after results()
            red.push(torch.tensor(float(rank + i)))
        out[rank] = (first.numpy().copy(), red.results().numpy().copy(), red.collectives)
    finally:
        dist.destroy_process_group()


def test_windowed_energy_all_reduce_two_ranks():
    """One collective per window of steps instead of one per step: reduced values, order,
    partial windows, reuse."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_window_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        for rank in range(world):
            first, second, ncoll = out[rank]
            assert np.array_equal(first, np.array([1000 + 2000 + 2 * i for i in range(37)], dtype=np.float32))
            assert np.array_equal(second, np.array([0 + 1 + 2 * i for i in range(3)], dtype=np.float32))
            assert ncoll == 5 + 1                              # ceil(37 / 8) + the partial window of 3


def test_windowed_energy_all_reduce_without_process_group():
    from tssplat_amd.sharding import WindowedEnergyAllReduce
    red = WindowedEnergyAllReduce(4, "cpu")
    for i in range(10):
        red.push(torch.tensor(float(i)))
    assert np.array_equal(red.results().numpy(), np.arange(10, dtype=np.float32)) and red.collectives == 0


def _scene_uneven(n_spheres, seed=3):
    """Spheres of very different sizes (kuhn 1 .. 4: 6 .. 384 tets) so that the balanced cut is far from equal counts."""
    rng = np.random.default_rng(seed)
    ks = rng.integers(1, 5, size=n_spheres)
    rest, tets, vo, to = [], [], [0], [0]
    for k in ks:
        v, t = scenes.kuhn_ball(int(k))
        rest.append((v * rng.uniform(0.1, 0.3) + rng.uniform(-0.5, 0.5, 3)).astype(np.float32))
        tets.append(t + vo[-1])
        vo.append(vo[-1] + v.shape[0])
        to.append(to[-1] + t.shape[0])
    rest = np.concatenate(rest)
    x = (rest + 0.03 * rng.standard_normal(rest.shape)).astype(np.float32)
    return rest, np.concatenate(tets).astype(np.int32), np.array(vo), np.array(to), x


def _worker8(rank, world, port, n_spheres, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rest, tets, vo, to, x = _scene_uneven(n_spheres)
        # windowed exchange (opt-in): rank-local value per step, job-wide energies from reduced_energies()
        mod = ShardedSmoothnessBarrierEnergy(rest, tets, _Flags, vo, to, local_factory=_OracleEnergy, exchange="window", window=4)
        lo, hi = mod.vertex_range
        xl = torch.from_numpy(x[lo:hi].copy()).requires_grad_(True)
        local, grads = [], None
        for it in range(6):                                  # one full window + a partial one; order 2 throughout
            c1, c2 = mod.coeff_scheduler(it)
            e = mod(xl, it, c1 * (1 + it), c2)
            xl.grad = None
            e.backward()
            local.append(float(e.detach()))
            grads = xl.grad.numpy().copy()
            if rank == 0:                                    # a log line on ONE rank: under no_grad nothing is filed, the windows stay in step
                with torch.no_grad():
                    assert abs(float(mod(xl, it, c1 * (1 + it), c2)) - local[-1]) <= 1e-6 * abs(local[-1]) + 1e-12
        reduced = mod.reduced_energies().numpy().copy()
        # replicated-parameter mode on top of the per-step exchange
        mod2 = ShardedSmoothnessBarrierEnergy(rest, tets, _Flags, vo, to, local_factory=_OracleEnergy, exchange="step")
        xf = torch.from_numpy(x).requires_grad_(True)
        c1, c2 = mod2.coeff_scheduler(0)
        ef = mod2.forward_replicated(xf, 0, c1, c2)
        ef.backward()
        out[rank] = (mod.sphere_range, (lo, hi), local, reduced, grads, float(ef), xf.grad.numpy().copy(),
                     mod._reducer.collectives if mod._reducer is not None else 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_spheres", [13, 5])
def test_eight_ranks_uneven_spheres_and_more_ranks_than_spheres(n_spheres):
    """world_size 8 on CPU (gloo): 13 spheres of very different sizes, and 5 spheres for 8 ranks (three ranks own nothing,
    contribute zero and still take part in every collective).  Windowed exchange: the per-step values are rank-local and
    sum to the oracle's energy; reduced_energies() returns the job-wide energies of all six steps from two collectives;
    replicated-parameter mode returns the full gradient on every rank."""
    from oracle import tet_energy_oracle as O
    world = 8
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker8, args=(world, _free_port(), n_spheres, out), nprocs=world, join=True)
        rest, tets, vo, to, x = _scene_uneven(n_spheres)
        cache = O.prepare(rest, tets)
        E = [O.energy_and_grad(x, cache, _Flags.smooth_eng_coeff * (1 + it), _Flags.barrier_coeff, 2)[0] for it in range(6)]
        E0, _, _, g0 = O.energy_and_grad(x, cache, _Flags.smooth_eng_coeff, _Flags.barrier_coeff, 2)
        g5 = O.energy_and_grad(x, cache, _Flags.smooth_eng_coeff * 6, _Flags.barrier_coeff, 2)[3]
        covered = np.zeros(rest.shape[0], dtype=bool)
        owners = 0
        for rank in range(world):
            srange, (lo, hi), local, reduced, grads, ef, grad_full, ncoll = out[rank]
            owners += srange[1] > srange[0]
            assert ncoll == 2                                        # 6 steps, window 4: one full + one partial window
            assert np.allclose(reduced, E, rtol=3e-6, atol=0)        # job-wide energies, every rank, in step order
            assert grads.shape == (hi - lo, 3)
            if hi > lo:                                              # (a rank without spheres owns no rows)
                assert np.abs(grads - g5[lo:hi]).max() <= 1e-5 * np.abs(g5).max()   # rank-local gradient of the last step
            assert abs(ef - E0) <= 3e-6 * abs(E0) and np.abs(grad_full - g0).max() <= 1e-5 * np.abs(g0).max()
            assert not covered[lo:hi].any()
            covered[lo:hi] = True
        assert covered.all() and owners == min(world, n_spheres)
        # the rank-local values of a step sum to the job-wide energy
        for it in range(6):
            assert abs(sum(out[r][2][it] for r in range(world)) - E[it]) <= 3e-6 * abs(E[it])


def test_bench_spawns_eight_ranks_dry_run():
    """`python bench.py --gpus 8 --dry-run`: the self-spawn route at the node's full width, more spheres than ranks and fewer
    (5 spheres on 8 ranks: three ranks hold empty shards and still meet every collective)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for spheres in (19, 5):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--spheres", str(spheres),
                              "--steps", "3", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 8 and rec["config"]["energy_allreduce_checked"] and rec["config"]["spheres_total"] == spheres


def test_windowed_reducer_bounds_what_it_keeps():
    """A loop that never asks for the reduced energies must not accumulate device tensors: with max_pending the oldest
    windows are dropped (with one warning), the newest are still delivered in order."""
    import warnings
    from tssplat_amd.sharding import WindowedEnergyAllReduce
    red = WindowedEnergyAllReduce(2, "cpu", max_inflight=1, max_pending=3)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for k in range(20):
            red.push(torch.tensor(float(k)))
    assert any("reduced windows waiting" in str(w.message) for w in rec) and red.dropped > 0
    got = red.results().tolist()
    assert len(got) <= 2 * (3 + 1) + 2 and got == sorted(got) and got[-1] == 19.0


def test_rank_scene_slices_equal_the_full_scene():
    """bench.py builds only a rank's spheres (scenes.make_scene(..., sphere_range=...), scenes.deform(..., vertex_range=...)): they
    must be exactly the slice of the full scene that TetScene.slice_spheres / the full deformation give."""
    from tssplat_amd import scenes
    from tssplat_amd.sharding import partition_spheres
    for kind, total, world in (("kuhn3", 13, 8), ("cone", 5, 2), ("aveg", 3, 2)):
        full = scenes.make_scene(kind, total, seed=0)
        xf = scenes.deform(full, 0.02, seed=1)
        tv, tt = scenes.template_mesh(kind, seed=0)
        for lo, hi in partition_spheres([int(tt.shape[0])] * total, world):
            part = scenes.make_scene(kind, total, seed=0, sphere_range=(lo, hi))
            ref = full.slice_spheres(lo, hi)
            assert np.array_equal(part.rest, ref.rest) and np.array_equal(part.tets, ref.tets)
            assert np.array_equal(part.sphere_vertex_offsets, ref.sphere_vertex_offsets) and np.array_equal(part.radii, ref.radii)
            v0, v1 = lo * tv.shape[0], hi * tv.shape[0]
            assert np.array_equal(scenes.deform(part, 0.02, seed=1, vertex_range=(v0, v1, full.n_vertices)), xf[v0:v1])
