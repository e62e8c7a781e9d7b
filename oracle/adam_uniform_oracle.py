"""CPU float64 ORACLE for the AdamUniform step -- TEST INFRASTRUCTURE (never imported by the product).

Restates /root/reference/utils/optimizer.py:38-89 on numpy float64 arrays:
  :61-62  g1 = b1*g1 + (1-b1)*grad ;  g2 = b2*g2 + (1-b2)*grad^2
  :67-68  m1 = g1/(1-b1^t) ; m2 = g2/(1-b2^t)
  :74     gr = m1 / (1e-8 + max(sqrt(m2)))
  :76-86  optional clamp: m = grad_limit_values[ptr] (ptr advances when cc >= grad_limit_iters[ptr]);
          if max|gr| > m: gr *= m / max|gr|
  :88     p -= lr*gr ;  :89 cc += 1
Pinned by tests/golden/adam_uniform_golden.npz, produced by running the reference class itself
(tests/golden/make_golden.py; utils/optimizer.py needs only torch and imports here).
"""
from __future__ import annotations

import numpy as np


class AdamUniformOracle:
    def __init__(self, p, grad_limit=False, grad_limit_values=(0.05, 0.01), grad_limit_iters=(4000,), lr=0.1,
                 betas=(0.9, 0.999)):
        self.p = np.asarray(p, dtype=np.float64).copy()
        self.g1 = np.zeros_like(self.p)
        self.g2 = np.zeros_like(self.p)
        self.t = 0
        self.cc = 0
        self.lr, self.b1, self.b2 = float(lr), float(betas[0]), float(betas[1])
        self.grad_limit = grad_limit
        self.values, self.iters, self.ptr = list(grad_limit_values), list(grad_limit_iters), 0

    def step(self, grad):
        g = np.asarray(grad, dtype=np.float64)
        self.t += 1
        self.g1 = self.b1 * self.g1 + (1 - self.b1) * g
        self.g2 = self.b2 * self.g2 + (1 - self.b2) * g * g
        m1 = self.g1 / (1 - self.b1 ** self.t)
        m2 = self.g2 / (1 - self.b2 ** self.t)
        gr = m1 / (1e-8 + np.sqrt(m2).max())
        if self.grad_limit:
            m = self.values[self.ptr]
            if self.ptr < len(self.iters) and self.cc >= self.iters[self.ptr]:
                self.ptr += 1
            s = np.abs(gr).max()
            if s > m:
                gr = gr * (m / s)
        self.p = self.p - self.lr * gr
        self.cc += 1
        return self.p
