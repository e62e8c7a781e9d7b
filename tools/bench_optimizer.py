#!/usr/bin/env python3
"""AdamUniform step: fused HIP step vs the reference's sequence of torch ops on the
same GPU, on an [n,3] fp32 parameter.  Algorithmic bytes per step = 8 floats per element
(read grad, g1, g2, p; write g1, g2, p; g1 read again by the apply pass) = 32 B/element.

    python tools/bench_optimizer.py [--vertices 4096000] [--steps 50]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def torch_reference_step(p, grad, g1, g2, step, lr, b1, b2, limit):
    """utils/optimizer.py:61-88 verbatim in torch ops (two device->host syncs when limit is set)."""
    import torch
    g1.mul_(b1).add_(grad, alpha=1 - b1)
    g2.mul_(b2).add_(grad.square(), alpha=1 - b2)
    m1 = g1 / (1 - (b1 ** step))
    m2 = g2 / (1 - (b2 ** step))
    gr = m1 / (1e-8 + m2.sqrt().max())
    if limit > 0:
        s = torch.max(torch.abs(gr))
        if s > limit:
            gr.mul_(limit / s)
    p.sub_(gr, alpha=lr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vertices", type=int, default=4096000)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    from tssplat_amd.utils import AdamUniform
    n = args.vertices
    p = torch.nn.Parameter(torch.randn(n, 3, device="cuda"))
    grad = torch.randn(n, 3, device="cuda")
    opt = AdamUniform([p], lr=0.1, grad_limit=True, grad_limit_values=[0.05], grad_limit_iters=[])
    p.grad = grad
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.step()
    torch.cuda.synchronize()
    fused = (time.perf_counter() - t0) / args.steps
    q = torch.randn(n, 3, device="cuda")
    g1, g2 = torch.zeros_like(q), torch.zeros_like(q)
    for s in range(1, 6):
        torch_reference_step(q, grad, g1, g2, s, 0.1, 0.9, 0.999, 0.05)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(6, 6 + args.steps):
        torch_reference_step(q, grad, g1, g2, s, 0.1, 0.9, 0.999, 0.05)
    torch.cuda.synchronize()
    ref = (time.perf_counter() - t0) / args.steps
    b = 32.0 * 3 * n
    print(json.dumps({"elements": 3 * n, "fused_ms": fused * 1e3, "torch_ops_ms": ref * 1e3, "speedup": ref / fused,
                      "fused_GBps_algorithmic": b / fused / 1e9, "hbm_frac_of_8TBps": b / fused / 8e12}))


if __name__ == "__main__":
    main()
