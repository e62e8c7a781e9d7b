"""``n`` optimisation steps -- energy + gradient, then ``AdamUniform.step`` -- as ONE HIP graph.

The reference's iteration (/root/reference/trainer.py:94-133) is ``out = renderer(...)``, ``loss.backward()``,
``optimizer.step()``.  For the geometry energy alone (the regulariser-only phases, and the benchmark configurations 2-3 of
BASELINE.json: 64-256 tet-spheres of ~3 k tets) one iteration is 15-25 us of kernels under 80-100 us of per-step host work
(autograd round trip, optimizer bookkeeping).  ``FusedEnergyAdamLoop`` hands the library the parameter, its gradient buffer
and the optimiser's moments once; every ``run(first_iteration)`` then replays ``n`` whole iterations from one graph launch,
with the coefficient schedule (``coeff_scheduler``, the order switch) and the optimiser's schedule (bias corrections,
``grad_limit``) written into the graph's kernel-node arguments beforehand.

The arithmetic is the eager path's, kernel for kernel (``tests/test_gpu_parity.py::test_fused_train_loop_equals_eager``:
bit-identical parameters after the same number of steps).  The loss is the energy alone (upstream gradient 1): a loop with an
image loss goes through ``tssplat_amd.renderers`` and autograd instead.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _capi, tet_spheres_ext
from ..utils.optimizer import AdamUniform
from .smooth_barrier import SmoothnessBarrierEnergy

__all__ = ["FusedEnergyAdamLoop"]

_lib = _capi.load()


class FusedEnergyAdamLoop:
    """``run(it0) -> energies[n]``: iterations ``it0 .. it0 + n - 1`` of ``energy(x, it, *coeff_scheduler(it)).backward();
    optimizer.step()`` from one graph launch.  ``param`` must keep its storage and be the optimiser's only parameter."""

    def __init__(self, energy: SmoothnessBarrierEnergy, param: torch.Tensor, optimizer: AdamUniform, n_iters: int = 16):
        ts = energy.tet_sp
        if not isinstance(optimizer, AdamUniform):
            raise TypeError("FusedEnergyAdamLoop folds tssplat_amd.utils.optimizer.AdamUniform (utils/optimizer.py of the reference)")
        params = [p for g in optimizer.param_groups for p in g["params"]]
        if len(params) != 1 or params[0] is not param:
            raise RuntimeError("the optimiser must hold exactly the parameter the loop updates")
        if not param.is_cuda or param.dtype != torch.float32 or not param.is_contiguous() or param.numel() != ts.n3:
            raise RuntimeError("FusedEnergyAdamLoop needs a contiguous float32 GPU parameter of 3 * n_vertices elements")
        if param.device != ts.device:
            raise RuntimeError(f"the parameter is on {param.device} but the TetSpheres object lives on {ts.device}")
        self.module, self.param, self.optimizer, self.n_iters = energy, param, optimizer, int(n_iters)
        state = optimizer.state[param]
        if len(state) == 0:                                        # the optimiser's lazy initialisation (optimizer.py:48-51)
            state["step"] = 0
            state["g1"] = torch.zeros_like(param.data)
            state["g2"] = torch.zeros_like(param.data)
        dev = param.device
        self.grad = torch.zeros_like(param.data)
        self.energies = torch.zeros(self.n_iters, dtype=torch.float32, device=dev)
        self._ws = torch.zeros(int(_lib.tsamd_train_loop_workspace_bytes(self.n_iters)), dtype=torch.uint8, device=dev)
        self._loop = C.c_void_p()
        with torch.cuda.device(dev):
            _capi.check(_lib.tsamd_train_loop_create(ts._handle(), param.data.data_ptr(), self.grad.data_ptr(), state["g1"].data_ptr(),
                                                     state["g2"].data_ptr(), self.energies.data_ptr(), self._ws.data_ptr(), self.n_iters,
                                                     C.byref(self._loop)))
        self._keep = (state["g1"], state["g2"])                     # the graph holds their addresses

    def close(self) -> None:
        if self._loop:
            _lib.tsamd_train_loop_destroy(self._loop)
            self._loop = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, first_iteration: int) -> torch.Tensor:
        """Replays ``n_iters`` iterations starting at ``first_iteration``; returns the (static, device) vector of their energies.
        The optimiser's counters (``state['step']``, ``cc``, ``grad_limit_ptr``) advance exactly as ``n_iters`` calls of
        ``step()`` would advance them."""
        opt, n = self.optimizer, self.n_iters
        state = opt.state[self.param]
        if state["g1"] is not self._keep[0] or state["g2"] is not self._keep[1]:
            raise RuntimeError("the optimiser's moment tensors were replaced (reset()?): build a new FusedEnergyAdamLoop")
        group = opt.param_groups[0]
        c1 = (C.c_float * n)()
        c2 = (C.c_float * n)()
        order = (C.c_int32 * n)()
        limit = (C.c_float * n)()
        for k in range(n):
            it = first_iteration + k
            c1[k], c2[k] = self.module.coeff_scheduler(it)
            order[k] = 4 if it > self.module.FLAGS.increase_order_iter else 2
            limit[k] = -1.0
            if opt.grad_limit:                                      # AdamUniform.step: the pointer advances after the read
                limit[k] = float(opt.grad_limit_values[opt.grad_limit_ptr])
                if opt.grad_limit_ptr < len(opt.grad_limit_iters) and opt.cc >= opt.grad_limit_iters[opt.grad_limit_ptr]:
                    opt.grad_limit_ptr += 1
            opt.cc += 1
        first_step = int(state["step"]) + 1
        state["step"] += n
        b1, b2 = group["betas"]
        with torch.cuda.device(self.param.device):
            _capi.check(_lib.tsamd_train_loop_launch(self._loop, c1, c2, order, float(group["lr"]), float(b1), float(b2), first_step, limit,
                                                     tet_spheres_ext._stream_ptr(self.param.device)))
        return self.energies
