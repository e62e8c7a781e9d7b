"""Sharding tet-spheres over the GPUs of a node (one process per GPU, torch.distributed / RCCL).

Tet-spheres share no vertices and no faces: the reference concatenates them with a running
vertex offset (/root/reference/geometry/tetmesh_geometry.py:310-331), so the operators are
block-diagonal and ``E = sum_s E_s``.  Each rank therefore owns a contiguous range of whole
spheres -- its slice of ``x`` and of the gradient -- and the path needs exactly one exchange
per evaluation: the sum of the scalar energies (8 bytes over xGMI, latency bound, issued
asynchronously so it never sits on the gradient's critical path).  No gradient exchange.

The reference itself has no distributed code; this is new host logic, covered
by world_size-2 and -8 gloo tests on CPU (tests/test_sharding_gloo.py).

Three forms of that exchange: ``WindowedEnergyAllReduce`` (one collective per window of steps, issued by the caller: bench.py's loop),
``OverlappedEnergyAllReduce`` (issued by a helper thread -- csrc/torch_exchange.cpp when the in-tree extension is there --, per step or
per window, values readable by ticket) and the plain per-call all-reduce; ``ShardedSmoothnessBarrierEnergy`` is the module a trainer
holds, ``JobWideEnergy`` the tensor its ``forward`` returns.
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Sequence

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["partition_spheres", "ShardedSmoothnessBarrierEnergy", "all_reduce_energy", "slice_replicated",
           "WindowedEnergyAllReduce", "OverlappedEnergyAllReduce", "JobWideEnergy"]


def partition_spheres(tets_per_sphere: Sequence[int], world_size: int) -> list[tuple[int, int]]:
    """Contiguous sphere ranges ``[lo, hi)`` per rank, balanced on tet count.

    Greedy prefix cut at multiples of ``total / world_size``; every rank gets at least one sphere
    while spheres last, ranks beyond the sphere count get an empty range.
    """
    counts = np.asarray(tets_per_sphere, dtype=np.int64)
    S = int(counts.size)
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if S <= world_size:          # fewer spheres than ranks: one each (a sphere cannot be split), the rest own nothing
        return [(r, r + 1) if r < S else (S, S) for r in range(world_size)]
    prefix = np.concatenate([[0], np.cumsum(counts)])
    total = int(prefix[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        c = int(np.argmin(np.abs(prefix - target)))          # nearest sphere boundary
        lo = cuts[-1] + 1 if S >= world_size else cuts[-1]   # non-empty ranges while spheres last
        hi = S - (world_size - r) if S >= world_size else S
        cuts.append(int(min(max(c, lo), max(hi, lo) if S >= world_size else S)))
    cuts.append(S)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


class _AddGlobal(torch.autograd.Function):
    """``local + (global - local)`` with the gradient of ``local`` only: the forward value is the
    job-wide energy, the backward pass stays rank-local (each rank owns its vertices)."""

    @staticmethod
    def forward(ctx, local, global_value):
        return global_value.to(local.device, local.dtype).reshape(local.shape).clone()

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out, None


class _SliceReplicated(torch.autograd.Function):
    """``x_full[lo:hi]`` whose backward returns the FULL gradient on every rank: each rank contributes the
    gradient of its own vertex range and the ranges are exchanged with one all-gather (uneven sizes, RCCL
    all-gather over xGMI on a GPU node).  For trainers that keep ``tet_v`` replicated on all ranks
    (the reference trainer unchanged, e.g. with view-parallel rendering)."""

    @staticmethod
    def forward(ctx, x_full, ranges, rank, group):
        ctx.ranges, ctx.rank, ctx.group, ctx.shape = ranges, rank, group, x_full.shape
        lo, hi = ranges[rank]
        return x_full[lo:hi].contiguous()

    @staticmethod
    def backward(ctx, grad_local):
        rows = [hi - lo for lo, hi in ctx.ranges]
        make = torch.empty if sum(rows) == ctx.shape[0] else torch.zeros    # rows no rank owns get a zero gradient
        full = make(ctx.shape, dtype=grad_local.dtype, device=grad_local.device)
        lo, hi = ctx.ranges[ctx.rank]
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(ctx.group) == 1:
            full[lo:hi] = grad_local
        elif len(set(rows)) == 1 and sum(rows) == ctx.shape[0]:
            dist.all_gather_into_tensor(full, grad_local.contiguous(), group=ctx.group)   # lands in place
        else:
            # uneven ranges: gather equal-sized (padded) blocks, then copy each rank's rows to their place
            # (all_gather of uneven tensor lists is not available on every backend)
            mx = max(rows)
            block = torch.zeros((mx,) + tuple(ctx.shape[1:]), dtype=grad_local.dtype, device=grad_local.device)
            block[: hi - lo] = grad_local
            gathered = torch.empty((len(rows) * mx,) + tuple(ctx.shape[1:]), dtype=grad_local.dtype, device=grad_local.device)
            dist.all_gather_into_tensor(gathered, block, group=ctx.group)
            for r, (a, b) in enumerate(ctx.ranges):
                full[a:b] = gathered[r * mx: r * mx + (b - a)]
        return full, None, None, None


def slice_replicated(x_full: torch.Tensor, vertex_ranges, rank: int, group=None) -> torch.Tensor:
    """This rank's rows of a replicated ``[n, 3]`` tensor; its gradient comes back all-gathered to ``[n, 3]``."""
    return _SliceReplicated.apply(x_full, [tuple(map(int, r)) for r in vertex_ranges], int(rank), group)


def all_reduce_energy(local_energy: torch.Tensor, group=None, async_op: bool = False):
    """Sum the scalar energies over ranks.  Returns the reduced tensor (and the work handle when
    ``async_op``).  With one rank or no initialised process group it is the identity."""
    e = local_energy.detach().clone().reshape(1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return (e.reshape(()), None) if async_op else e.reshape(())
    work = dist.all_reduce(e, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return (e.reshape(()), work) if async_op else e.reshape(())


class WindowedEnergyAllReduce:
    """The path's only exchange, batched over steps.

    The gradient of a rank's vertices never depends on the job-wide energy, which a trainer only logs
    (/root/reference/trainer.py:118-125).  At 8-way strong scaling of the 512-sphere scene a step is ~70 us of
    kernels, and enqueueing one RCCL all-reduce per step costs the host a comparable amount (measured on one GPU with a
    single-rank group: profiles/r03_scaling_model.json).  So the local energies of ``window`` consecutive steps go
    into a ring of device slots -- ``push`` is one 4-byte device copy, no collective -- and every ``window``-th push
    issues ONE asynchronous all-reduce over the whole window.  ``results()`` returns the reduced energies of every
    pushed step, in push order (it flushes a partial window and waits for the collectives in flight).

    With one rank or no initialised process group the collective is the identity; everything else runs unchanged.
    """

    def __init__(self, window: int, device, group=None, max_inflight: int = 2, max_pending: int = 0):
        if window < 1:
            raise ValueError("window must be >= 1")
        self.window, self.group, self.max_inflight = int(window), group, max(1, int(max_inflight))
        self.max_pending, self.dropped = int(max_pending), 0   # 0 = keep every reduced window until results() is called
        self._bufs = [torch.zeros(self.window, dtype=torch.float32, device=device) for _ in range(self.max_inflight + 1)]
        self._cur, self._fill = 0, 0
        self._inflight: list[tuple[int, int, object]] = []   # (buffer index, entries, work handle)
        self._done: list[torch.Tensor] = []
        self._active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= 1
        self.collectives = 0

    def push(self, local_energy: torch.Tensor) -> None:
        self._bufs[self._cur][self._fill:self._fill + 1].copy_(local_energy.detach().reshape(1), non_blocking=True)
        self.commit()

    def slot(self) -> torch.Tensor:
        """The ring slot of the NEXT step (a one-element view): an evaluation that can write its energy to a second address -- a
        HIP-graph replay, ``GraphedSmoothnessBarrier.step(..., energy_copy=slot)`` -- fills it itself on the current stream; then
        :meth:`commit`.  Saves ``push``'s 4-byte copy kernel: 2-3 us of a 66 us step at 8-way strong scaling."""
        return self._bufs[self._cur][self._fill:self._fill + 1]

    def commit(self) -> None:
        self._fill += 1
        if self._fill == self.window:
            self.flush()

    def _retire(self) -> None:
        b, cnt, work = self._inflight.pop(0)
        if work is not None:
            work.wait()                                   # (stream-side for RCCL: the host does not block)
        self._done.append(self._bufs[b][:cnt].clone())
        if self.max_pending > 0 and len(self._done) > self.max_pending:
            if self.dropped == 0:
                import warnings
                warnings.warn(f"WindowedEnergyAllReduce: more than {self.max_pending} reduced windows waiting for results(); "
                              "dropping the oldest (call reduced_energies() / results() from the training loop, or use exchange='step')")
            self.dropped += len(self._done) - self.max_pending
            del self._done[:len(self._done) - self.max_pending]

    def flush(self) -> None:
        """All-reduce what the current window holds (also called by ``push`` when a window is full)."""
        if self._fill == 0:
            return
        buf = self._bufs[self._cur]
        work = None
        if self._active:
            work = dist.all_reduce(buf[:self._fill] if self._fill < self.window else buf, op=dist.ReduceOp.SUM,
                                   group=self.group, async_op=True)
            self.collectives += 1
        self._inflight.append((self._cur, self._fill, work))
        while len(self._inflight) > self.max_inflight:
            self._retire()
        busy = {b for b, _, _ in self._inflight}
        self._cur = next(i for i in range(len(self._bufs)) if i not in busy)
        self._fill = 0

    def results(self) -> torch.Tensor:
        """Reduced energies of all steps pushed since the last call (1-D, push order)."""
        self.flush()
        while self._inflight:
            self._retire()
        out = torch.cat(self._done) if self._done else self._bufs[0][:0].clone()
        self._done = []
        return out


class OverlappedEnergyAllReduce:
    """One all-reduce per step (or per ``every`` steps) that costs the training THREAD nothing: issued by a helper thread on a side
    stream, waited for only by whoever reads its result.  (What a collective per step still costs is on the GPU's timeline: measured
    +28 us per step at the 8-way share of the 512-sphere scene -- hence ``every``.)

    ``submit(local_energy)`` (the training thread) copies the scalar into a ring slot on the current stream, records an event
    and hands (slot, event) to the helper thread -- a few microseconds, no collective call.  The helper makes a side stream
    wait for the event and enqueues ``dist.all_reduce(slot, async_op=True)`` there (RCCL on a GPU node: the collective's own
    stream then waits for the side stream, never for the compute stream, and the compute stream never waits for it); the
    20-40 us a collective call costs the host (profiles/r05_scaling_model.json: 63.5 -> 89.7 us per step at 64 spheres per
    rank) are spent next to the training thread, not in it.  ``value(ticket)`` waits -- host-side until the helper has
    issued the collective, stream-side for its completion -- and returns the job-wide energy.  Collectives are issued in
    ticket order by ONE thread, so every rank issues the same sequence as long as every rank submits the same sequence
    (the rule of any collective; ``every`` > 1: ONE collective per window of ``every`` evaluations, issued with the window's last one --
    for steps so short that even an asynchronous collective per step shows on the GPU's timeline; a value is readable once its
    window has gone out, a read before that raises) -- on ``group``, which must carry NOTHING else: the training thread's own collectives would
    interleave with the helper's in a different order on every rank (``ShardedSmoothnessBarrierEnergy`` creates a group for
    it).  A slot is re-used after ``depth`` tickets: older values have expired.

    With one rank or no initialised process group the collective is the identity; everything else runs unchanged.
    """

    def __init__(self, device, group=None, depth: int = 256, use_extension: bool = True, every: int = 1):
        if depth < 2:
            raise ValueError("depth must be >= 2")
        if every < 1 or depth % every or depth < 2 * every:
            raise ValueError("`every` must divide the ring depth at least twice")
        self.device, self.group, self.depth, self.every = torch.device(device), group, int(depth), int(every)
        self._committed = self._issued_upto = 0
        self.ring = torch.zeros(self.depth, dtype=torch.float32, device=self.device)
        self._slots = [self.ring[s:s + 1] for s in range(self.depth)]
        self._cuda = self.device.type == "cuda"
        self._side = torch.cuda.Stream(self.device) if self._cuda else None
        self._events = [torch.cuda.Event() for _ in range(self.depth)] if self._cuda else None
        self._works: list = [None] * self.depth
        self._issued = [threading.Event() for _ in range(self.depth)]
        for ev in self._issued:
            ev.set()
        self._ticket_of_slot = [-1] * self.depth
        self._next = 0
        self._q: queue.SimpleQueue = queue.SimpleQueue()
        self._thread: threading.Thread | None = None
        self._error: BaseException | None = None
        self._active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= 1
        self._collectives = 0
        # The helper thread lives in the in-tree C++ extension when that is there (csrc/torch_exchange.cpp: same protocol, a
        # std::thread, c10d called without the interpreter); the Python thread below is the fallback.  A Python helper competes
        # with the training thread for the interpreter lock: 45 us of a 64 us step on the MI355X box (tools/host_overhead.py).
        self._cxx = None
        if use_extension:
            from . import _capi
            ext = _capi.autograd_ext()
            if ext is not None and hasattr(ext, "EnergyExchange"):
                pg = None
                if self._active:
                    pg = group if group is not None else dist.distributed_c10d._get_default_group()
                self._cxx = ext.EnergyExchange(self.ring, pg, self.every)

    @property
    def collectives(self) -> int:
        return int(self._cxx.collectives) if self._cxx is not None else self._collectives

    # ---- helper thread (fallback) ----
    def _worker(self) -> None:
        # What this thread does per collective while it holds the interpreter lock is what the training thread loses: the
        # process group's own allreduce (one pybind call that releases the lock) instead of dist.all_reduce (argument checks,
        # two logging decorators), the side stream made current ONCE for the thread, the slot views cut beforehand.
        pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
        if self._cuda:
            torch.cuda.set_device(self.device)
            torch.cuda.set_stream(self._side)
        opts = dist.AllreduceOptions()
        opts.reduceOp = dist.ReduceOp.SUM
        while True:
            item = self._q.get()
            if item is None:
                return
            s, count = item
            try:
                if self._cuda:
                    self._side.wait_event(self._events[s])
                w = pg.allreduce([self.ring[s:s + count]], opts)
                for k in range(count):
                    self._works[s + k] = w                    # (every slot of the window waits on the same handle)
                self._collectives += 1
            except BaseException as exc:                      # noqa: BLE001  (handed to the reader)
                self._error = exc
            finally:
                for k in range(count):
                    self._issued[s + k].set()

    def _settle(self, s: int) -> None:
        """The collective that last used slot ``s`` has been issued and the CURRENT stream is ordered behind its completion."""
        self._issued[s].wait()
        if self._error is not None:
            raise RuntimeError("the energy all-reduce failed in the helper thread") from self._error
        w = self._works[s]
        if w is not None:
            if not w.is_completed():                          # (a finished collective needs no wait on the calling stream)
                w.wait()                                      # (RCCL: stream-side; gloo: the host waits)
            w0 = s - s % self.every                           # (every slot of a window shares the handle: one wait serves them all)
            for k in range(w0, w0 + self.every):
                if self._works[k] is w:
                    self._works[k] = None

    # ---- training thread ----
    def reserve(self) -> tuple[int, torch.Tensor]:
        """A ticket and its ring slot (a one-element view): the caller has the local energy written there ON THE CURRENT STREAM --
        a replay does it itself, ``tsamd_graph_launch_to`` -- and then calls :meth:`commit`."""
        if self._cxx is not None:
            return self._cxx.reserve()
        t = self._next
        self._next += 1
        s = t % self.depth
        if self._ticket_of_slot[s] >= 0:
            self._settle(s)                                   # (depth tickets old: long done; orders the overwrite behind it)
        self._ticket_of_slot[s] = t
        return t, self._slots[s]

    def commit(self, ticket: int) -> None:
        """Tickets are committed in the order they were reserved.  With ``every`` > 1 the collective covers the slots of a window
        and goes out with the window's last ticket."""
        if self._cxx is not None:
            return self._cxx.commit(ticket)
        if ticket != self._committed:
            raise RuntimeError("tickets are committed in the order they were reserved")
        self._committed += 1
        if self._committed % self.every == 0:
            self._issue(self._issued_upto, self._committed - self._issued_upto)

    def flush(self) -> None:
        """The all-reduce of the committed part of the current window now (a collective: every rank, at the same ticket)."""
        if self._cxx is not None:
            return self._cxx.flush()
        if self._committed > self._issued_upto:
            self._issue(self._issued_upto, self._committed - self._issued_upto)

    def _issue(self, first: int, count: int) -> None:
        self._issued_upto = first + count
        if not self._active:
            return
        s0 = first % self.depth
        for k in range(count):
            self._issued[s0 + k].clear()
        if self._cuda:
            self._events[s0].record(torch.cuda.current_stream(self.device))
        if self._thread is None:
            self._thread = threading.Thread(target=self._worker, name="tssplat_amd-energy-allreduce", daemon=True)
            self._thread.start()
        self._q.put((s0, count))

    def submit(self, local_energy: torch.Tensor) -> int:
        t, slot = self.reserve()
        slot.copy_(local_energy.detach().reshape(1), non_blocking=True)
        self.commit(t)
        return t

    def value(self, ticket: int) -> torch.Tensor:
        """Job-wide energy of ``ticket`` (0-dim, a fresh tensor)."""
        if self._cxx is not None:
            return self._cxx.value(ticket)
        if not 0 <= ticket < self._next:
            raise ValueError(f"unknown ticket {ticket}")
        s = ticket % self.depth
        if self._ticket_of_slot[s] != ticket:
            raise RuntimeError(f"the job-wide energy of evaluation {ticket} has expired: {self._next - ticket} evaluations ago, the ring keeps "
                               f"{self.depth} (read it sooner, or build the module with a larger `depth`)")
        if ticket >= self._issued_upto:
            raise RuntimeError(f"the job-wide energy of evaluation {ticket} is not on its way yet: with every = {self.every} the all-reduce of a "
                               f"window is issued with its last evaluation ({self._issued_upto} evaluations are covered so far).  Read it later, "
                               "call flush() on EVERY rank, or build the module with every = 1")
        self._settle(s)
        return self.ring[s].clone()

    def drain(self) -> None:
        """Every collective submitted so far has been issued and the current stream is ordered behind all of them."""
        if self._cxx is not None:
            return self._cxx.drain()
        for s in range(self.depth):
            if 0 <= self._ticket_of_slot[s] < self._issued_upto:
                self._settle(s)

    def close(self) -> None:
        if self._cxx is not None:
            self._cxx.close()
        if self._thread is not None:
            self._q.put(None)
            self._thread.join(timeout=10)
            self._thread = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                     # noqa: BLE001
            pass


class _AttachGradient(torch.autograd.Function):
    """Forward value = the job-wide energy; backward = ``grad_output x`` the gradient an engine-free evaluation left in its
    buffer (``still_valid()``: the buffer has not been overwritten by a newer evaluation)."""

    @staticmethod
    def forward(ctx, x, value, grad, still_valid):
        ctx.grad, ctx.still_valid = grad, still_valid
        return value.to(x.device, torch.float32).reshape(()).clone()

    @staticmethod
    def backward(ctx, go):
        if not ctx.still_valid():
            raise RuntimeError("backward() of an energy whose gradient buffer a newer evaluation has overwritten: call backward() before "
                               "the next forward, or build the module with graph=False")
        g = ctx.grad
        return g * go.detach().to(device=g.device, dtype=g.dtype), None, None, None


def _plain(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, torch.Tensor) and type(t) is not torch.Tensor else t


class JobWideEnergy(torch.Tensor):
    """What ``ShardedSmoothnessBarrierEnergy.forward`` returns with ``exchange="overlap"``: a 0-dim tensor whose VALUE is the
    job-wide energy and whose gradient is this rank's.  The all-reduce that produces the value is already on its way
    (:class:`OverlappedEnergyAllReduce`); the tensor waits for it the first time something READS the value -- ``float(e)``,
    ``print``, ``loss + e`` -- and never when nothing does: ``e.backward()`` (or ``torch.autograd.backward((image_loss, e))``)
    needs no value, so a training loop whose step only back-propagates and logs every n-th energy never has the exchange on its
    path.  Reading goes through ``__torch_function__``: every torch function except the few that only touch metadata or
    start the backward pass gets the resolved plain tensor (forward value = job-wide energy, backward = identity to the local
    energy)."""

    _NO_VALUE = {"backward", "register_hook", "retain_grad", "requires_grad_", "ones_like", "zeros_like", "empty_like", "size", "dim",
                 "numel", "is_floating_point", "is_complex", "element_size", "get_device", "data_ptr", "_version", "type", "stride",
                 "storage_offset", "is_contiguous", "untyped_storage", "__hash__", "__reduce_ex__"}
    _NO_VALUE_GETTERS = {"shape", "dtype", "device", "requires_grad", "grad_fn", "is_leaf", "grad", "ndim", "is_cuda", "is_cpu", "layout",
                         "names", "is_sparse", "is_quantized", "is_meta", "output_nr", "_backward_hooks", "retains_grad", "is_nested", "_grad"}

    @staticmethod
    def wrap(local: torch.Tensor, reducer: "OverlappedEnergyAllReduce", ticket: int) -> "JobWideEnergy":
        t = local.as_subclass(JobWideEnergy)                  # (an alias of `local`: stays attached to its autograd node)
        t._tsamd_reducer, t._tsamd_ticket, t._tsamd_resolved = reducer, ticket, None
        return t

    @staticmethod
    def wrap_direct(energy: torch.Tensor, reducer: "OverlappedEnergyAllReduce", ticket: int, x: torch.Tensor, grad: torch.Tensor,
                    still_valid) -> "JobWideEnergy":
        """The engine-free flavour (``graph=True``): the evaluation ran without an autograd node and left dE/dx in ``grad``.
        ``backward()`` with no arguments adds it to ``x.grad`` itself -- what the autograd engine would do for this one-node
        graph, minus the engine's round trip (40-50 us of host time per step, more than a 64-sphere share of the headline scene
        takes on the GPU).  Anything else -- arithmetic, ``backward(gradient=...)``, ``inputs=`` -- first attaches the tensor to
        ``x`` through an ordinary autograd node (:meth:`resolve`).  The tensor itself does NOT require grad: hand it to
        ``torch.autograd.backward`` / ``grad`` as ``e.resolve()``."""
        t = energy.detach().as_subclass(JobWideEnergy)
        t._tsamd_reducer, t._tsamd_ticket, t._tsamd_resolved = reducer, ticket, None
        t._tsamd_direct = (x, grad, still_valid)
        return t

    def resolve(self) -> torch.Tensor:
        """The plain tensor: job-wide value (waits for the exchange), rank-local gradient."""
        red = getattr(self, "_tsamd_reducer", None)
        with torch._C.DisableTorchFunctionSubclass():
            plain = self.as_subclass(torch.Tensor)
            if red is None:                                   # (a by-product such as ones_like(e): an ordinary tensor)
                return plain
            if self._tsamd_resolved is None:
                direct = getattr(self, "_tsamd_direct", None)
                if direct is not None and direct[1] is None:
                    return red.value(self._tsamd_ticket)      # (already back-propagated: the value is all that is left)
                if direct is not None and torch.is_grad_enabled():
                    self._tsamd_resolved = _AttachGradient.apply(direct[0], red.value(self._tsamd_ticket), direct[1], direct[2])
                elif direct is not None:
                    return red.value(self._tsamd_ticket)      # (under no_grad: the value only; a later read attaches)
                else:
                    self._tsamd_resolved = _AddGlobal.apply(plain, red.value(self._tsamd_ticket))
            return self._tsamd_resolved

    def _backward_direct(self) -> None:
        x, grad, still_valid = self._tsamd_direct
        if grad is None:
            raise RuntimeError("this energy has been back-propagated already and its gradient tensor handed to x.grad (the engine-free path "
                               "keeps no graph): evaluate again, or use e.resolve().backward(retain_graph=True)")
        if not still_valid():
            raise RuntimeError("backward() of an energy whose gradient buffer a newer evaluation has overwritten: call backward() before "
                               "the next forward, or build the module with graph=False")
        with torch._C.DisableTorchFunctionSubclass(), torch.no_grad():
            if x.grad is None:
                x.grad = grad                                 # (the evaluation's own tensor, written by the replay: handed over, not copied)
            else:
                x.grad.add_(grad)
        self._tsamd_direct = (x, None, still_valid)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        owner = getattr(func, "__self__", None)               # property getters arrive as `<getset_descriptor>.__get__`
        if name == "backward" and args and getattr(args[0], "_tsamd_direct", None) is not None:
            me = args[0]
            if len(args) == 1 and not any(kwargs.get(k) for k in ("gradient", "inputs", "create_graph")):
                return me._backward_direct()                  # d e / d e = 1: the gradient buffer goes to x.grad as it is
            with torch._C.DisableTorchFunctionSubclass():
                return me.resolve().backward(*args[1:], **kwargs)
        if name in cls._NO_VALUE or (name == "__get__" and getattr(owner, "__name__", "") in cls._NO_VALUE_GETTERS):
            with torch._C.DisableTorchFunctionSubclass():
                return _plain(func(*args, **kwargs))
        scalar_read = name in ("item", "__float__", "__int__", "__bool__", "__format__", "tolist", "__repr__", "__str__")
        res = lambda a: (a.resolve().detach() if scalar_read else a.resolve()) if isinstance(a, JobWideEnergy) else a   # noqa: E731
        args = tuple(type(a)(res(b) for b in a) if isinstance(a, (list, tuple)) else res(a) for a in args)
        kwargs = {k: res(v) for k, v in kwargs.items()}
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class ShardedSmoothnessBarrierEnergy(torch.nn.Module):
    """``SmoothnessBarrierEnergy`` over this rank's spheres; ``forward`` returns the JOB-WIDE energy -- one scalar all-reduce per
    call, issued next to the training thread and waited for only when the value is read (``exchange="overlap"``, the default:
    :class:`JobWideEnergy`), or inside the call (``exchange="step"``) -- or, as an explicit opt-in, the rank-local one with the
    reduction batched over steps (``exchange="window"``).  ``graph=True`` replays the rank's evaluation from a HIP graph behind
    the autograd node (``SmoothnessBarrierEnergy(graph=True)``).

    Parameters mirror the reference module (/root/reference/energies/smooth_barrier.py:34-45) plus
    the sphere layout: ``sphere_vertex_offsets`` / ``sphere_tet_offsets`` (``S+1`` entries each,
    as produced by the multi-sphere geometry's ``base_vid`` bookkeeping).  ``forward`` takes the
    rank-local slice ``x[v_lo:v_hi]`` (use :attr:`vertex_range`); its gradient w.r.t. the local slice is the
    local gradient.  How the scalar energies of the ranks meet is ``exchange`` (see :meth:`forward`): one all-reduce per
    step by default -- the returned tensor is then what the reference's module returns, whoever calls it and however
    often -- or, opted into by a training loop that calls ``forward`` once per step on EVERY rank, windowed and
    asynchronous (at 8-way strong scaling a step is ~70 us and enqueueing a collective per step costs the host about as much).

    ``local_factory(rest_local, tets_local, FLAGS)`` builds the rank-local evaluator; it defaults
    to the HIP-backed ``SmoothnessBarrierEnergy`` and exists so the CPU tests can exercise the
    partition / collective logic with an oracle-backed stand-in.
    """

    def __init__(self, tet_v, tet_f, FLAGS, sphere_vertex_offsets, sphere_tet_offsets, group=None,
                 rank: int | None = None, world_size: int | None = None,
                 local_factory: Callable | None = None, exchange: str = "overlap", window: int = 16, depth: int = 256,
                 every: int = 1, graph: bool = False, **local_kwargs):
        super().__init__()
        if exchange not in ("overlap", "window", "step"):
            raise ValueError("exchange must be 'overlap' (one all-reduce per call issued off the training thread, the job-wide value "
                             "waited for when it is read), 'step' (the same all-reduce inside the call) or 'window' (rank-local "
                             "value; one collective per `window` steps, job-wide values from reduced_energies())")
        self.exchange, self.window, self.depth, self.every = exchange, int(window), int(depth), int(every)
        self._overlap = None                     # OverlappedEnergyAllReduce, created on the first forward (needs the device)
        self._energy_group = group
        self.max_pending = 64                    # reduced windows kept for reduced_energies() (exchange="window")
        self._reducer = None                     # WindowedEnergyAllReduce, created on the first forward (needs the device)
        initialised = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = rank if rank is not None else (dist.get_rank(group) if initialised else 0)
        self.world_size = world_size if world_size is not None else (dist.get_world_size(group) if initialised else 1)
        vo = np.asarray(sphere_vertex_offsets, dtype=np.int64)
        to = np.asarray(sphere_tet_offsets, dtype=np.int64)
        if vo.size != to.size or vo.size < 1:
            raise ValueError("sphere offset arrays must both have S+1 entries")
        if exchange == "overlap" and initialised and dist.get_world_size(group) > 1:
            # The helper thread's all-reduces need a communicator of their own: collectives of one group are matched by the ORDER in
            # which each rank issues them, and the training thread issues others meanwhile (forward_replicated's all-gather, a
            # data-parallel wrapper's gradient buckets) -- which of the two threads comes first differs from rank to rank.
            # (Collective: every rank of `group` constructs the module.)
            if group is None:
                self._energy_group = dist.new_group()
            else:
                self._energy_group = dist.new_group(ranks=dist.get_process_group_ranks(group), use_local_synchronization=True)
        self.ranges = partition_spheres(np.diff(to), self.world_size)
        lo, hi = self.ranges[self.rank]
        self.sphere_range = (lo, hi)
        self.vertex_range = (int(vo[lo]), int(vo[hi]))
        self.vertex_ranges = [(int(vo[a]), int(vo[b])) for a, b in self.ranges]   # of every rank (slice_replicated)
        self.tet_range = (int(to[lo]), int(to[hi]))
        v = np.asarray(tet_v).reshape(-1, 3)[self.vertex_range[0]:self.vertex_range[1]]
        f = np.asarray(tet_f).reshape(-1, 4)[self.tet_range[0]:self.tet_range[1]] - self.vertex_range[0]
        if f.size and (f.min() < 0 or f.max() >= max(v.shape[0], 1)):
            raise ValueError("a tet references a vertex outside its sphere range: spheres must not share vertices")
        if local_factory is None:
            from .energies import SmoothnessBarrierEnergy
            self.local = SmoothnessBarrierEnergy(v, f, FLAGS, graph=graph, **local_kwargs) if v.shape[0] else None
        else:
            if graph or local_kwargs:
                raise TypeError("graph= and TetSpheres options go to the default local evaluator; a local_factory builds its own")
            self.local = local_factory(v, f, FLAGS) if v.shape[0] else None
        self.FLAGS = FLAGS

    def coeff_scheduler(self, it):
        if self.local is not None:
            return self.local.coeff_scheduler(it)
        from .energies import SmoothnessBarrierEnergy
        return SmoothnessBarrierEnergy.coeff_scheduler(self, it)

    def forward_replicated(self, x_full: torch.Tensor, it, c1, c2):
        """Same energy from a ``tet_v`` replicated on every rank: evaluates this rank's spheres and hands
        every rank the full ``[n, 3]`` gradient (one all-gather of the rank slices in backward)."""
        return self.forward(slice_replicated(x_full, self.vertex_ranges, self.rank, self.group), it, c1, c2)

    def forward(self, x_local: torch.Tensor, it, c1, c2):
        """``exchange="overlap"`` (default): one all-reduce per call, the JOB-WIDE energy is the value of the returned
        :class:`JobWideEnergy`, its gradient the rank-local gradient.  The call itself only files the local energy (a 4-byte
        device copy and an event); a helper thread issues the collective on a side stream, and the returned tensor waits for it
        when -- and only if -- its value is read.  ``e.backward()`` reads nothing.  Every rank must make the call, in the same
        order (it is a collective); calls under ``torch.no_grad()`` are collectives like any other and return the plain job-wide
        value.  The values of the last ``depth`` calls stay readable.  ``every=n`` (constructor): one collective per ``n`` calls, issued
        with the n-th -- for steps of a few tens of microseconds, where even an asynchronous collective per step is visible (8-way
        strong scaling of the 512-sphere scene: 70 us per step, ~20 us of them the per-step exchange); a value is then readable from
        the end of its window on, an earlier read raises (``flush_exchange()`` -- on every rank -- sends a partial window).
        ``exchange="step"``: the same all-reduce issued and waited for inside the call (stream-side); returns a plain tensor.
        ``exchange="window"`` (opt-in): returns THIS RANK's energy -- same gradient, which is all an optimiser needs --
        and files it into a :class:`WindowedEnergyAllReduce`: one asynchronous collective per ``window`` steps, nothing on
        the step's critical path; :meth:`reduced_energies` hands out the job-wide energies of the steps evaluated so far
        (what the reference only logs, trainer.py:118-125).  Calls under ``torch.no_grad()`` (logging, validation) are NOT
        filed -- they return the local value and leave the hidden collective state alone, so a rank-0-only log line
        cannot put the windows of the ranks out of step; at most ``max_pending`` reduced windows are kept for
        :meth:`reduced_energies` (the oldest are dropped with a warning: a loop that never asks for them does not
        accumulate device tensors)."""
        if self.exchange == "overlap":
            if self._overlap is None:
                self._overlap = OverlappedEnergyAllReduce(x_local.device, self._energy_group, self.depth, every=self.every)
            # Engine-free evaluation where nothing can tell the difference: a local evaluator that offers it (graph=True), a leaf
            # parameter without hooks.  The replay writes the energy into the exchange's ring slot itself.
            if (self.local is not None and torch.is_grad_enabled() and x_local.requires_grad and x_local.is_leaf
                    and hasattr(self.local, "evaluate_direct") and not x_local._backward_hooks
                    and not getattr(x_local, "_post_accumulate_grad_hooks", None)):
                ticket, slot = self._overlap.reserve()
                direct = self.local.evaluate_direct(x_local, it, c1, c2, energy_copy=slot)
                if direct is not None:
                    self._overlap.commit(ticket)
                    return JobWideEnergy.wrap_direct(direct[0], self._overlap, ticket, x_local, direct[1], direct[2])
                e_local = self.local(x_local, it, c1, c2)     # (this x cannot be replayed: the ordinary path into the same slot)
                slot.copy_(e_local.detach().reshape(1), non_blocking=True)
                self._overlap.commit(ticket)
                return JobWideEnergy.wrap(e_local, self._overlap, ticket)
        if self.local is not None:
            e_local = self.local(x_local, it, c1, c2)
        else:                                   # more ranks than spheres: contribute zero
            e_local = x_local.sum() * 0.0
        if self.exchange == "step":
            e_global = all_reduce_energy(e_local, self.group)
            return _AddGlobal.apply(e_local, e_global)
        if self.exchange == "overlap":
            ticket = self._overlap.submit(e_local)
            if not (torch.is_grad_enabled() and e_local.requires_grad) and self.every == 1:
                return self._overlap.value(ticket)
            return JobWideEnergy.wrap(e_local, self._overlap, ticket)
        if not torch.is_grad_enabled():
            return e_local
        if self._reducer is None:
            self._reducer = WindowedEnergyAllReduce(self.window, e_local.device, self.group, max_pending=self.max_pending)
        self._reducer.push(e_local)
        return e_local

    def flush_exchange(self) -> None:
        """``exchange="overlap"`` with ``every`` > 1: issue the all-reduce of the current, partial window now.  A collective: every rank."""
        if self._overlap is not None:
            self._overlap.flush()

    def reduced_energies(self) -> torch.Tensor:
        """Job-wide energies of every step evaluated since the last call (1-D, evaluation order; flushes a partial window
        and waits for the collectives in flight).  Every rank must call it at the same step count."""
        if self._reducer is None:
            return torch.zeros(0)
        return self._reducer.results()
