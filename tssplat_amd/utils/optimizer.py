"""Mirror of /root/reference/utils/optimizer.py (`AdamUniform`, :4-89) on one fused HIP step.

Same constructor arguments, state keys (``step``, ``g1``, ``g2``), ``reset()`` and grad-limit
schedule (``grad_limit_values`` / ``grad_limit_iters`` / ``grad_limit_ptr`` / ``cc``) as the reference
class; the arithmetic of ``step()`` (optimizer.py:61-88) runs in ``tsamd_adam_uniform_step`` -- two
kernels, no host synchronisation, where the reference issues ~10 elementwise torch ops, two ``.max()``
reductions and, with ``grad_limit``, a Python-side ``if s > m`` (a device->host sync every step).

SURVEY.md 8(f) row 3: the step right after the energy backward, over the same ``[n, 3]`` array.
Parameters must be contiguous float32 GPU tensors; there is no CPU path.
"""
from __future__ import annotations

import torch

from .. import _capi

__all__ = ["AdamUniform"]


class AdamUniform(torch.optim.Optimizer):
    """Variant of Adam with uniform scaling by the second moment (optimizer.py:5-10)."""

    def __init__(self, params, grad_limit=False, grad_limit_values=[0.05, 0.01], grad_limit_iters=[4000],
                 lr=0.1, betas=(0.9, 0.999)):
        defaults = dict(lr=lr, betas=betas)
        self.grad_limit = grad_limit
        super().__init__(params, defaults)
        self.cc = 0
        if grad_limit:
            self.grad_limit_values = grad_limit_values
            self.grad_limit_iters = grad_limit_iters
            self.grad_limit_ptr = 0
        self._lib = _capi.load()
        self._ws = {}

    @torch.no_grad()
    def reset(self):
        for group in self.param_groups:
            for p in group["params"]:
                state = self.state[p]
                state["step"] = 0
                state["g1"] = torch.zeros_like(p.data)
                state["g2"] = torch.zeros_like(p.data)

    def _workspace(self, device):
        ws = self._ws.get(device)
        if ws is None:
            ws = torch.empty(16, dtype=torch.uint8, device=device)
            self._ws[device] = ws
        return ws

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            lr = group["lr"]
            b1, b2 = group["betas"]
            for p in group["params"]:
                state = self.state[p]
                if len(state) == 0:                       # lazy initialisation, optimizer.py:48-51
                    state["step"] = 0
                    state["g1"] = torch.zeros_like(p.data)
                    state["g2"] = torch.zeros_like(p.data)
                state["step"] += 1
                grad = p.grad.data
                if not (p.is_cuda and p.dtype == torch.float32 and p.data.is_contiguous()):
                    raise RuntimeError("tssplat_amd AdamUniform needs contiguous float32 GPU parameters (no CPU path)")
                if not grad.is_contiguous():
                    grad = grad.contiguous()
                limit = -1.0
                if self.grad_limit:                       # optimizer.py:76-81 (pointer advances after the read)
                    limit = float(self.grad_limit_values[self.grad_limit_ptr])
                    if self.grad_limit_ptr < len(self.grad_limit_iters):
                        if self.cc >= self.grad_limit_iters[self.grad_limit_ptr]:
                            self.grad_limit_ptr += 1
                stream = int(torch.cuda.current_stream(p.device).cuda_stream)
                with torch.cuda.device(p.device):
                    _capi.check(self._lib.tsamd_adam_uniform_step(
                        p.data.data_ptr(), grad.data_ptr(), state["g1"].data_ptr(), state["g2"].data_ptr(),
                        p.numel(), float(lr), float(b1), float(b2), int(state["step"]), limit,
                        self._workspace(p.device).data_ptr(), stream))
                self.cc += 1
