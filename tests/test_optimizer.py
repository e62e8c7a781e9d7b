"""AdamUniform: oracle pinned to the reference class's own trajectory, host-side
schedule logic of the mirror, and (GPU) the fused HIP step against the oracle."""
import os

import numpy as np
import pytest

from oracle.adam_uniform_oracle import AdamUniformOracle

CASES = (("plain", dict(lr=0.2)),
         ("limited", dict(lr=0.2, grad_limit=True, grad_limit_values=[0.05, 0.01], grad_limit_iters=[3])))


def test_oracle_matches_reference_trajectory(golden_dir):
    """adam_uniform_golden.npz was produced by /root/reference/utils/optimizer.py::AdamUniform itself."""
    z = np.load(os.path.join(golden_dir, "adam_uniform_golden.npz"))
    for name, kw in CASES:
        o = AdamUniformOracle(z["p0"], **kw)
        for it in range(z[f"{name}_grads"].shape[0]):
            p = o.step(z[f"{name}_grads"][it])
            assert np.abs(p - z[f"{name}_traj"][it]).max() <= 1e-14
    # the limited run really clamps: its steps are bounded by lr * limit
    lim = z["limited_traj"]
    steps = np.abs(np.diff(np.concatenate([z["p0"][None], lim]), axis=0)).max(axis=(1, 2))
    assert np.all(steps <= 0.2 * 0.05 * (1 + 1e-12))


def test_mirror_schedule_bookkeeping(monkeypatch):
    """grad_limit pointer / counter logic of optimizer.py:76-89, checked without running a kernel."""
    torch = pytest.importorskip("torch")
    from tssplat_amd.utils import optimizer as mod

    calls = []

    class FakeLib:
        def tsamd_adam_uniform_step(self, p, g, g1, g2, n, lr, b1, b2, step, limit, ws, stream):
            calls.append((n, round(lr, 6), step, round(limit, 6)))
            return 0

    opt = mod.AdamUniform([torch.nn.Parameter(torch.zeros(4, 3))], grad_limit=True, grad_limit_values=[0.05, 0.01],
                          grad_limit_iters=[2], lr=0.3)
    assert set(opt.defaults) == {"lr", "betas"} and opt.cc == 0 and opt.grad_limit_ptr == 0
    opt._lib = FakeLib()
    p = opt.param_groups[0]["params"][0]
    # pretend the parameter lives on the GPU: only the launch is faked
    monkeypatch.setattr(type(p), "is_cuda", property(lambda self: True), raising=False)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: type("S", (), {"cuda_stream": 0})())
    monkeypatch.setattr(torch.cuda, "device", lambda d: __import__("contextlib").nullcontext())
    monkeypatch.setattr(mod.AdamUniform, "_workspace", lambda self, dev: torch.zeros(16, dtype=torch.uint8))
    for _ in range(5):
        p.grad = torch.ones(4, 3)
        opt.step()
    # limit used: 0.05 while cc < 2 and on the step where cc == 2 (pointer advances after the read), then 0.01
    assert [c[3] for c in calls] == [0.05, 0.05, 0.05, 0.01, 0.01]
    assert [c[2] for c in calls] == [1, 2, 3, 4, 5] and opt.cc == 5 and opt.grad_limit_ptr == 1
    state = opt.state[p]
    assert set(state) == {"step", "g1", "g2"} and state["step"] == 5
    opt.reset()
    assert opt.state[p]["step"] == 0 and float(opt.state[p]["g1"].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", CASES)
def test_fused_step_matches_oracle_on_gpu(golden_dir, name, kw):
    """fp32 fused HIP step vs float64 oracle on the reference trajectory's gradients:
    |p - p64| <= 4e-6 * (|p64| + lr) per entry after every step."""
    torch = pytest.importorskip("torch")
    from tssplat_amd.utils import AdamUniform
    z = np.load(os.path.join(golden_dir, "adam_uniform_golden.npz"))
    p = torch.nn.Parameter(torch.from_numpy(z["p0"]).float().cuda())
    opt = AdamUniform([p], **kw)
    oracle = AdamUniformOracle(z["p0"].astype(np.float32), **kw)
    for it in range(z[f"{name}_grads"].shape[0]):
        g32 = z[f"{name}_grads"][it].astype(np.float32)
        p.grad = torch.from_numpy(g32).cuda()
        opt.step()
        ref = oracle.step(g32)
        got = p.detach().cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - ref) <= 4e-6 * (np.abs(ref) + kw["lr"])), (name, it, np.abs(got - ref).max())
    assert opt.cc == 6


@pytest.mark.gpu
def test_fused_step_large_and_loud(golden_dir):
    torch = pytest.importorskip("torch")
    from tssplat_amd.utils import AdamUniform
    n = 1_000_003
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal((n, 3)).astype(np.float32)
    g = (rng.standard_normal((n, 3)) * 3).astype(np.float32)
    p = torch.nn.Parameter(torch.from_numpy(p0).cuda())
    opt = AdamUniform([p], lr=0.1, grad_limit=True, grad_limit_values=[0.02], grad_limit_iters=[])
    oracle = AdamUniformOracle(p0, lr=0.1, grad_limit=True, grad_limit_values=[0.02], grad_limit_iters=[])
    for _ in range(3):
        p.grad = torch.from_numpy(g).cuda()
        opt.step()
        ref = oracle.step(g)
    got = p.detach().cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() <= 4e-6 * (np.abs(ref).max() + 0.1)
    with pytest.raises(RuntimeError):
        q = torch.nn.Parameter(torch.zeros(4, 3))            # CPU parameter: no fallback
        o2 = AdamUniform([q])
        q.grad = torch.ones(4, 3)
        o2.step()
