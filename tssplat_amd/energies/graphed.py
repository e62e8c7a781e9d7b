"""HIP-graph replay of the energy forward+backward for launch-bound batches.

For the reference's own problem sizes (64-256 tet-spheres of ~3 k tets, BASELINE.json configs 2-3) one
evaluation is 10-30 us of kernels, while the eager route through ``torch.autograd`` costs 70-80 us of host
time per step (tools/host_overhead.py) -- the reference pays that, plus three blocking device reads, at
/root/reference/trainer.py:94,130 in every iteration.  ``GraphedSmoothnessBarrier`` replays the fused evaluation
(the tile and finish kernels) from a HIP graph that the library builds once per order (``tsamd_graph_create``); the
schedule of ``SmoothnessBarrierEnergy.coeff_scheduler`` (smooth_barrier.py:47-58) keeps working because the
coefficients are kernel arguments of the graph's nodes, updated on the host at every launch.  The order switch 2 -> 4
(smooth_barrier.py:61-63) selects a second graph, built on first use.

The result is the same energy and the same gradient as ``SmoothnessBarrierEnergy`` + ``backward()`` with
``grad_output = grad_scale``; it is written to ``self.energy`` / ``self.grad`` (static buffers, overwritten by
every ``step``).  ``step`` itself does not go through autograd: add ``self.grad`` to the parameter's ``.grad`` (or
hand it to ``AdamUniform``) yourself.  For code shaped like the reference trainer (``loss = ... + energy(x, it, c1, c2)``,
``loss.backward()``, /root/reference/trainer.py:94-130) ``SmoothnessBarrierEnergy(..., graph=True)`` wraps the same
replay in an autograd node (``GraphReplayFunc``): the trainer keeps its shape and gets the replayed kernels -- though
on this path the ~70 us that ``torch.autograd`` itself spends on a custom Function round trip dominate the step.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _capi, tet_spheres_ext
from .smooth_barrier import SmoothnessBarrierEnergy

__all__ = ["GraphedSmoothnessBarrier", "GraphReplayFunc"]

_lib = _capi.load()


class GraphedSmoothnessBarrier:
    """``step(it) -> (energy, grad)`` by HIP-graph replay; ``x`` must keep its storage (in-place updates only).

    Parameters: ``energy`` -- a ``SmoothnessBarrierEnergy`` (supplies the ``TetSpheres`` object, the schedule and
    ``increase_order_iter``); ``x`` -- the ``[n, 3]`` float32 parameter on the GPU; ``grad_scale`` -- the factor
    the reference applies as ``grad_output`` (tet_spheres_cuda.cu:257-258), e.g. the weight of the
    regularisation term in the loss.

    The graph lives in the library (``tsamd_graph_create`` / ``tsamd_graph_launch``): two kernel nodes whose coefficient
    arguments are updated on the host at every launch, so the reference's schedule (coefficients change every iteration
    up to it = 1200, smooth_barrier.py:47-58) costs nothing on the device.
    """

    def __init__(self, energy: SmoothnessBarrierEnergy, x: torch.Tensor, grad_scale: float = 1.0):
        ts = energy.tet_sp
        if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.numel() != ts.n3:
            raise RuntimeError("GraphedSmoothnessBarrier needs a contiguous float32 GPU tensor of 3 * n_vertices elements")
        if x.device != ts.device:
            raise RuntimeError(f"x is on {x.device} but the TetSpheres object lives on {ts.device}")
        self.module = energy
        self.x = x
        dev = x.device
        self.energy = torch.zeros((), dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(x.detach())
        self._scale = torch.full((1,), float(grad_scale), dtype=torch.float32, device=dev)
        self._graphs: dict[int, C.c_void_p] = {}
        # evaluations so far (a backward must belong to the latest one): a one-element CPU tensor so that the C++ autograd node
        # (csrc/torch_autograd.cpp) and GraphReplayFunc count on the same cell
        self.ticket_tensor = torch.zeros(1, dtype=torch.int64)
        self._ticket_cell = self.ticket_tensor.numpy()      # (the same memory: a numpy scalar access costs 0.1 us, a tensor index 3 us)

    @property
    def ticket(self) -> int:
        return int(self._ticket_cell[0])

    @ticket.setter
    def ticket(self, value: int) -> None:
        self._ticket_cell[0] = value

    def graph_address(self, order: int) -> int:
        """Address of the library graph for ``order`` (created on first use) -- what the C++ autograd node launches."""
        g = self._graphs.get(order)
        if g is None:
            g = self._graphs[order] = self._create(order)
        return int(g.value)

    def _create(self, order: int) -> C.c_void_p:
        g = C.c_void_p()
        _capi.check(_lib.tsamd_graph_create(self.module.tet_sp._handle(), self.x.data_ptr(), self._scale.data_ptr(), int(order),
                                            self.energy.data_ptr(), self.grad.data_ptr(), C.byref(g)))
        return g

    def close(self) -> None:
        for g in self._graphs.values():
            _lib.tsamd_graph_destroy(g)
        self._graphs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def evaluate(self, c1: float, c2: float, order: int, energy_copy: torch.Tensor | None = None, grad_out: torch.Tensor | None = None):
        """Replay the fused evaluation with these coefficients; returns ``(energy, grad)``: the static energy buffer and the
        gradient (already multiplied by ``grad_scale``) -- in the static buffer, or in ``grad_out`` when one is given (a contiguous
        float32 tensor shaped like ``x``; a per-launch argument of the replay, ``tsamd_graph_launch_to``: the caller keeps that
        tensor, no copy).  ``energy_copy``: a one-element float32 device tensor that receives the energy as well (an energy
        exchange's ring slot)."""
        g = self._graphs.get(order)
        if g is None:
            g = self._graphs[order] = self._create(order)
        if energy_copy is None and grad_out is None:
            _capi.check(_lib.tsamd_graph_launch(g, c1, c2, tet_spheres_ext._stream_ptr(self.x.device)))
            return self.energy, self.grad
        _capi.check(_lib.tsamd_graph_launch_to(g, c1, c2, tet_spheres_ext._stream_ptr(self.x.device),
                                               None if energy_copy is None else energy_copy.data_ptr(),
                                               None if grad_out is None else grad_out.data_ptr()))
        return self.energy, (self.grad if grad_out is None else grad_out)

    def step(self, it: int, c1: float | None = None, c2: float | None = None, energy_copy: torch.Tensor | None = None):
        """One evaluation at iteration ``it``: coefficients from ``coeff_scheduler(it)`` unless given, order 4 after
        ``FLAGS.increase_order_iter``.  Returns ``(energy, grad)`` -- the static device buffers.  ``energy_copy``: see
        :meth:`evaluate`."""
        if c1 is None or c2 is None:
            c1, c2 = self.module.coeff_scheduler(it)
        order = 4 if it > self.module.FLAGS.increase_order_iter else 2
        return self.evaluate(c1, c2, order, energy_copy)


class GraphReplayFunc(torch.autograd.Function):
    """The autograd node of ``SmoothnessBarrierEnergy(..., graph=True)``: same ``apply(x, ., c1, c2, order)`` shape as
    ``SmoothnessBarrierFunc`` (reference energies/smooth_barrier.py:9-31), with the evaluation replayed from a HIP graph.

    forward: one replay (tile + finish kernels: energy AND the unscaled gradient, into static buffers), returns the
    energy buffer (a fresh tensor object on the same storage -- valid until the next evaluation, like every output of a
    graphed callable).  backward: ``grad_output * gradient`` -- one small elementwise kernel, no second evaluation, and a
    fresh tensor, so ``x.grad`` never aliases the static buffer."""

    @staticmethod
    def forward(ctx, x_cur, graphed, c1, c2, order):
        energy, grad = graphed.evaluate(c1, c2, order)
        ctx.graphed = graphed
        ctx.ticket = graphed.ticket = graphed.ticket + 1
        return energy.detach()

    @staticmethod
    def backward(ctx, grad_output):
        if grad_output is None:
            return None, None, None, None, None
        graphed = ctx.graphed
        if ctx.ticket != graphed.ticket:
            raise RuntimeError("backward() of a graph-replayed energy after a newer evaluation overwrote its static gradient "
                               "buffer: call backward() before the next forward, or use graph=False")
        go = grad_output
        if go.device != graphed.grad.device or go.dtype != torch.float32:
            go = go.detach().to(device=graphed.grad.device, dtype=torch.float32, non_blocking=True)
        return graphed.grad * go, None, None, None, None
