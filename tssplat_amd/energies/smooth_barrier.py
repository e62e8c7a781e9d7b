"""Host-side mirror of /root/reference/energies/smooth_barrier.py (:9-67).

Same class names, constructor arguments, schedule and order switch, so a
trainer written against the reference's ``energies.smooth_barrier`` runs on
the MI355X kernels by changing one import.  The only dependency dropped is the
module-level ``import pypgo`` (smooth_barrier.py:1), which the reference file
never uses.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import tet_spheres_ext

__all__ = ["SmoothnessBarrierFunc", "SmoothnessBarrierEnergy"]


def _ALWAYS() -> bool:
    return True


class SmoothnessBarrierFunc(torch.autograd.Function):
    """autograd bridge: saves ``x`` only, constants ride on ctx (smooth_barrier.py:9-31).

    Same ``apply(x_cur, tet_sp, c1, c2, order)`` call and same saved state as the reference class.  The
    reference spells it ``forward`` + ``setup_context``; that form makes ``Function.apply`` run
    ``inspect.signature(forward).bind(...)`` on every call (torch/autograd/function.py, ``bind_default_args``),
    measured at ~45 us per step on the MI355X box -- four times the 12 us of GPU work of a 64-sphere batch.
    The ``forward(ctx, ...)`` spelling below is the same autograd node without that cost.
    """

    @staticmethod
    def forward(ctx, x_cur, tet_sp, c1, c2, order):
        ctx.save_for_backward(x_cur)
        ctx.constants = (tet_sp, c1, c2, order)
        return tet_spheres_ext.forward(x_cur, tet_sp, c1, c2, order)

    @staticmethod
    def backward(ctx, grad_output):
        if grad_output is None:
            return None, None, None, None, None
        (x_cur,) = ctx.saved_tensors
        tet_sp, c1, c2, order = ctx.constants
        grad_final = tet_spheres_ext.backward(grad_output, x_cur, tet_sp, c1, c2, int(order))
        return grad_final, None, None, None, None


class SmoothnessBarrierEnergy(torch.nn.Module):
    """Builds the device state from the rest mesh and evaluates the scheduled energy
    (smooth_barrier.py:34-67).  ``FLAGS`` needs ``smooth_eng_coeff``, ``barrier_coeff`` and
    ``increase_order_iter`` (config/gso.yaml:8-11)."""

    def __init__(self, tet_v, tet_f, FLAGS, graph: bool = False, **tet_spheres_kwargs) -> None:
        super().__init__()
        # graph=True (not in the reference): every differentiable evaluation is a HIP-graph replay behind an autograd
        # node (energies/graphed.py) -- for batches whose 10-30 us of kernels drown in 80 us of per-step host work.
        # The parameter must keep its storage (in-place optimiser updates, as torch optimisers and AdamUniform do).
        self.graph = bool(graph)
        self._graphed = None
        v_flat = np.asarray(tet_v).flatten().astype(np.float32)      # smooth_barrier.py:38
        f_flat = np.asarray(tet_f).flatten().astype(np.int32)        # smooth_barrier.py:39
        self.tet_sp = tet_spheres_ext.TetSpheres(v_flat, f_flat, **tet_spheres_kwargs)
        self.FLAGS = FLAGS
        self._ext = False                                     # C++ autograd extension: looked up on the first differentiable call
        self.smooth_eng_func = SmoothnessBarrierFunc          # (the reference instantiates the Function, smooth_barrier.py:45; torch deprecates that)

    def coeff_scheduler(self, it):
        """x1 at it=0 rising to x16 from it=1200 on (smooth_barrier.py:47-58)."""
        smooth_coeff = self.FLAGS.smooth_eng_coeff
        barrier_coeff = self.FLAGS.barrier_coeff
        multiplier = math.pow(2, abs(math.sin(min(it / 300.0 / 4 * 0.5 * math.pi, 0.5 * math.pi))) * 4)
        return smooth_coeff * multiplier, barrier_coeff * multiplier

    def evaluate_direct(self, x, it, c1, c2, energy_copy=None):
        """``graph=True`` only, not in the reference: the evaluation WITHOUT an autograd node -- one replay, returns
        ``(energy, grad, still_valid)``: the replay's static energy buffer, dE/dx (upstream gradient 1) in a FRESH tensor the
        replay wrote directly (the caller may keep it: no copy), and a callable that says whether ``grad`` still holds this
        evaluation (always: it is the evaluation's own).  For callers that apply the gradient themselves (tssplat_amd.sharding:
        ``JobWideEnergy.backward``); ``energy_copy`` as in ``GraphedSmoothnessBarrier.evaluate``.  None when ``x`` cannot be replayed
        (not this module's device / dtype / size)."""
        if not (self.graph and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() == self.tet_sp.n3
                and x.device == self.tet_sp.device):
            return None
        from .graphed import GraphedSmoothnessBarrier
        gr = self._graphed
        if gr is None or gr.x.data_ptr() != x.data_ptr() or gr.x.shape != x.shape:
            gr = self._graphed = GraphedSmoothnessBarrier(self, x.detach())
        order = 4 if it > self.FLAGS.increase_order_iter else 2
        energy, grad = gr.evaluate(c1, c2, order, energy_copy, torch.empty_like(gr.x))
        gr.ticket = gr.ticket + 1                             # (the autograd nodes' counter: a pending backward() of theirs sees a newer evaluation)
        return energy, grad, _ALWAYS

    def forward(self, x, it, c1, c2):
        order = 4 if it > self.FLAGS.increase_order_iter else 2      # smooth_barrier.py:61-63
        if not torch.is_grad_enabled():
            # logging / validation under torch.no_grad(): energy only -- no fused gradient pass, and a gradient kept for a
            # pending backward() of an earlier evaluation stays where it is
            return tet_spheres_ext.forward(x, self.tet_sp, c1, c2, order, fuse=False)
        # The autograd node is a C++ torch::autograd::Function when the in-tree extension is there (csrc/torch_autograd.cpp: one
        # call into the C ABI per evaluation, no Python frame in forward or backward); the Python Functions of this package
        # are the fallback and the reference-spelled surface.
        ext = self._ext
        if ext is False:
            from .. import _capi
            ext = self._ext = _capi.autograd_ext()
        if self.graph and x.requires_grad:
            from .graphed import GraphedSmoothnessBarrier, GraphReplayFunc
            gr = self._graphed
            if gr is None or gr.x.data_ptr() != x.data_ptr() or gr.x.shape != x.shape:
                gr = self._graphed = GraphedSmoothnessBarrier(self, x.detach())   # (re)capture for this storage
            if ext is not None:
                return ext.energy_replay(x, gr.graph_address(order), c1, c2, gr.energy, gr.grad, gr.ticket_tensor)
            return GraphReplayFunc.apply(x, gr, c1, c2, order)
        # (the extension always runs the fused forward+backward pass and allocates the gradient: only where that is what the
        # Python path would do as well -- a differentiable input and fusion not switched off)
        if (ext is not None and x.requires_grad and self.tet_sp.fuse_forward_backward and not tet_spheres_ext.cpu_energy_mode()
                and x.is_cuda and x.dtype == torch.float32 and x.numel() == self.tet_sp.n3 and x.device == self.tet_sp.device):
            self.tet_sp._cache = None      # (a fused result kept by the operator functions belongs to an older evaluation now)
            return ext.energy_eval(x, int(self.tet_sp._handle().value), c1, c2, order)
        return SmoothnessBarrierFunc.apply(x, self.tet_sp, c1, c2, order)
