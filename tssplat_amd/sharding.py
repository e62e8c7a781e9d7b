"""Sharding tet-spheres over the GPUs of a node (one process per GPU, torch.distributed / RCCL).

Tet-spheres share no vertices and no faces: the reference concatenates them with a running
vertex offset (/root/reference/geometry/tetmesh_geometry.py:310-331), so the operators are
block-diagonal and ``E = sum_s E_s``.  Each rank therefore owns a contiguous range of whole
spheres -- its slice of ``x`` and of the gradient -- and the path needs exactly one exchange
per evaluation: the sum of the scalar energies (8 bytes over xGMI, latency bound, issued
asynchronously so it never sits on the gradient's critical path).  No gradient exchange.

The reference itself has no distributed code; this is new host logic, covered
by world_size-2 gloo tests on CPU (tests/test_sharding_gloo.py).
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["partition_spheres", "ShardedSmoothnessBarrierEnergy", "all_reduce_energy", "slice_replicated",
           "WindowedEnergyAllReduce"]


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


class ShardedSmoothnessBarrierEnergy(torch.nn.Module):
    """``SmoothnessBarrierEnergy`` over this rank's spheres; ``forward`` returns the JOB-WIDE energy (one scalar all-reduce per
    step, ``exchange="step"``, the default) or, as an explicit opt-in, the rank-local one with the reduction batched
    over steps (``exchange="window"``).

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
                 local_factory: Callable | None = None, exchange: str = "step", window: int = 16):
        super().__init__()
        if exchange not in ("window", "step"):
            raise ValueError("exchange must be 'window' (one collective per `window` steps, off the step's path) or "
                             "'step' (one blocking-order all-reduce per step, job-wide value returned at once)")
        self.exchange, self.window = exchange, int(window)
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
            local_factory = SmoothnessBarrierEnergy
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
        """``exchange="step"`` (default): one all-reduce per call, the JOB-WIDE energy is the returned value; its gradient is
        the rank-local gradient.  Every rank must make the call (it is a collective).
        ``exchange="window"`` (opt-in): returns THIS RANK's energy -- same gradient, which is all an optimiser needs --
        and files it into a :class:`WindowedEnergyAllReduce`: one asynchronous collective per ``window`` steps, nothing on
        the step's critical path; :meth:`reduced_energies` hands out the job-wide energies of the steps evaluated so far
        (what the reference only logs, trainer.py:118-125).  Calls under ``torch.no_grad()`` (logging, validation) are NOT
        filed -- they return the local value and leave the hidden collective state alone, so a rank-0-only log line
        cannot put the windows of the ranks out of step; at most ``max_pending`` reduced windows are kept for
        :meth:`reduced_energies` (the oldest are dropped with a warning: a loop that never asks for them does not
        accumulate device tensors)."""
        if self.local is not None:
            e_local = self.local(x_local, it, c1, c2)
        else:                                   # more ranks than spheres: contribute zero
            e_local = x_local.sum() * 0.0
        if self.exchange == "step":
            e_global = all_reduce_energy(e_local, self.group)
            return _AddGlobal.apply(e_local, e_global)
        if not torch.is_grad_enabled():
            return e_local
        if self._reducer is None:
            self._reducer = WindowedEnergyAllReduce(self.window, e_local.device, self.group, max_pending=self.max_pending)
        self._reducer.push(e_local)
        return e_local

    def reduced_energies(self) -> torch.Tensor:
        """Job-wide energies of every step evaluated since the last call (1-D, evaluation order; flushes a partial window
        and waits for the collectives in flight).  Every rank must call it at the same step count."""
        if self._reducer is None:
            return torch.zeros(0)
        return self._reducer.results()
