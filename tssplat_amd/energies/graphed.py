"""HIP-graph replay of the energy forward+backward for launch-bound batches.

For the reference's own problem sizes (64-256 tet-spheres of ~3 k tets, BASELINE.json configs 2-3) one
evaluation is 10-30 us of kernels, while the eager route through ``torch.autograd`` costs 50-70 us of host
time per step (tools/host_overhead.py) -- the reference pays that, plus three blocking device reads, at
/root/reference/trainer.py:94,130 in every iteration.  ``GraphedSmoothnessBarrier`` captures the fused
evaluation once (the tile and finish kernels) and replays it; the schedule of
``SmoothnessBarrierEnergy.coeff_scheduler`` (smooth_barrier.py:47-58) keeps working because the kernels read ``c1, c2`` from device memory (``tsamd_evaluate_dev_coef``), refreshed by one small asynchronous
copy in front of the replay whenever they change (never while they are constant).  The order switch 2 -> 4
(smooth_barrier.py:61-63) selects a second graph, captured on first use.

The result is the same energy and the same gradient as ``SmoothnessBarrierEnergy`` + ``backward()`` with
``grad_output = grad_scale``; it is written to ``self.energy`` / ``self.grad`` (static buffers, overwritten by
every ``step``).  ``step`` itself does not go through autograd: add ``self.grad`` to the parameter's ``.grad`` (or
hand it to ``AdamUniform``) yourself.  For code shaped like the reference trainer (``loss = ... + energy(x, it, c1, c2)``,
``loss.backward()``, /root/reference/trainer.py:94-130) ``SmoothnessBarrierEnergy(..., graph=True)`` wraps the same
replay in an autograd node (``GraphReplayFunc``): the trainer keeps its shape and gets the replayed kernels.
"""
from __future__ import annotations

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
        self._coef = torch.zeros(2, dtype=torch.float32, device=dev)
        # pinned staging ring for the coefficients: a slot is rewritten only after the copy that read it has run
        self._ring = [torch.zeros(2, dtype=torch.float32).pin_memory() for _ in range(8)]
        self._ring_ev = [None] * len(self._ring)
        self._ring_pos = 0
        self._last = None
        self._scale = torch.full((1,), float(grad_scale), dtype=torch.float32, device=dev)
        self._graphs: dict[int, torch.cuda.CUDAGraph] = {}
        self._stream = torch.cuda.Stream(device=dev)
        self.ticket = 0            # evaluations so far (GraphReplayFunc: a backward must belong to the latest one)

    def _launch(self, order: int) -> None:
        stream = tet_spheres_ext._stream_ptr(self.x.device)
        _capi.check(_lib.tsamd_evaluate_dev_coef(self.module.tet_sp._handle(), self.x.data_ptr(), self._scale.data_ptr(),
                                                 self._coef.data_ptr(), int(order), stream, self.energy.data_ptr(),
                                                 self.grad.data_ptr()))

    def _capture(self, order: int) -> torch.cuda.CUDAGraph:
        dev = self.x.device
        with torch.cuda.device(dev):
            self._stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._stream):        # warm-up outside capture (first-launch module loading)
                self._launch(order)
            torch.cuda.current_stream(dev).wait_stream(self._stream)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._stream):
                self._launch(order)
        return g

    def evaluate(self, c1: float, c2: float, order: int):
        """Replay the fused evaluation with these coefficients; returns the static ``(energy, grad)`` buffers
        (``grad`` already multiplied by ``grad_scale``)."""
        if self._last != (c1, c2):
            k = self._ring_pos
            self._ring_pos = (k + 1) % len(self._ring)
            if self._ring_ev[k] is not None:
                self._ring_ev[k].synchronize()
            self._ring[k][0] = float(c1)
            self._ring[k][1] = float(c2)
            self._coef.copy_(self._ring[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.x.device))
            self._ring_ev[k] = ev
            self._last = (c1, c2)
        g = self._graphs.get(order)
        if g is None:
            g = self._graphs[order] = self._capture(order)
        g.replay()
        return self.energy, self.grad

    def step(self, it: int, c1: float | None = None, c2: float | None = None):
        """One evaluation at iteration ``it``: coefficients from ``coeff_scheduler(it)`` unless given, order 4 after
        ``FLAGS.increase_order_iter``.  Returns ``(energy, grad)`` -- the static device buffers."""
        if c1 is None or c2 is None:
            c1, c2 = self.module.coeff_scheduler(it)
        order = 4 if it > self.module.FLAGS.increase_order_iter else 2
        return self.evaluate(c1, c2, order)


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
