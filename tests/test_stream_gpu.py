"""GPU parity of the streaming-tile kernel (csrc/stream_kernels.hip, EXPERIMENTAL) against the float64 oracle -- the same three
regression guards as tests/test_gpu_parity.py (energy, gradient norm, worst vertex against the predicted fp32 rounding
error) -- and against the blob kernel (the product path) on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GUARD_E, GUARD_G, GUARD_V = 40.0, 8.0, 30.0


def _check(kind, S, sigma, order, go=1.0):
    from oracle import tet_energy_oracle as O
    from tssplat_amd import scenes
    from tssplat_amd.stream import StreamTetSpheres
    sc = scenes.make_scene(kind, S)
    st = StreamTetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    x_np = scenes.deform(sc, sigma)
    c1, c2 = 2e-4 / S, 2e-4
    x = torch.from_numpy(x_np).cuda()
    e, g = st.forward_backward(x, c1, c2, order, None if go == 1.0 else torch.tensor(go))
    e2, g2 = st.forward_backward(x, c1, c2, order, None if go == 1.0 else torch.tensor(go))
    assert float(e) == float(e2) and torch.equal(g, g2), "evaluation must be deterministic"
    cache = O.prepare(sc.rest, sc.tets)
    E, Es, Eb, gr = O.energy_and_grad(x_np, cache, c1, c2, order, grad_output=go)
    std_e, std_gv = O.rounding_error_model(x_np, cache, c1, c2, order)
    es_gpu, eb_gpu = st.energy_terms()
    e_terms = float(np.float32(c1)) * es_gpu + float(np.float32(c2)) * eb_gpu
    assert abs(e_terms - float(e)) <= 2.0 ** -24 * abs(e_terms) * 1.0001
    g_gpu = g.cpu().numpy().astype(np.float64)
    assert np.isfinite(g_gpu).all()
    err_e = abs(e_terms - E)
    err_v = np.linalg.norm(g_gpu - gr, axis=1)
    err_g = float(np.sqrt(np.sum(err_v ** 2)))
    std_g = abs(go) * float(np.sqrt(np.sum(std_gv ** 2)))
    floor_v = abs(go) * (std_gv + 1e-3 * float(np.sqrt(np.mean(std_gv ** 2))))
    ratio_v = float(np.max(err_v / np.maximum(floor_v, 1e-300)))
    print(f"[stream {kind}x{S} s={sigma} p={order}] E={E:.6e} err={err_e:.2e} |g|={np.linalg.norm(gr):.4e} err={err_g:.2e} "
          f"ratios E {err_e / max(std_e, 1e-300):.2f} g {err_g / max(std_g, 1e-300):.2f} vertex-max {ratio_v:.2f} "
          f"slots/tet {st.plan_info()['total_slots'] / sc.n_tets:.4f}")
    assert err_e <= GUARD_E * std_e, (err_e, std_e)
    assert err_g <= GUARD_G * std_g, (err_g, std_g)
    assert ratio_v <= GUARD_V, ratio_v
    return sc, x, c1, c2, float(e), g


@pytest.mark.parametrize("kind,S,sigma,order", [("kuhn4", 3, 0.3, 2), ("kuhn8", 8, 0.02, 2), ("kuhn8", 8, 0.3, 4), ("kuhn12", 2, 0.0, 2),
                                               ("kuhn19", 3, 0.02, 2), ("kuhn19", 2, 0.3, 4), ("cone", 2, 0.3, 2),
                                               ("delaunay1500", 3, 0.3, 2), ("kuhn3", 40, 0.3, 4)])
def test_stream_kernel_parity(kind, S, sigma, order):
    _check(kind, S, sigma, order, go=1.0 if order == 2 else 0.37)


def test_stream_kernel_equals_blob_kernel():
    from tssplat_amd import tet_spheres_ext as T
    sc, x, c1, c2, e, g = _check("kuhn19", 4, 0.02, 2)
    ts = T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    xr = x.clone().requires_grad_(True)
    eb = T.forward(xr, ts, c1, c2, 2)
    gb = T.backward(torch.tensor(1.0), xr, ts, c1, c2, 2)
    assert abs(float(eb) - e) <= 2e-6 * abs(e)
    assert float((gb - g).norm()) <= 2e-6 * float(gb.norm())


def test_stream_real_mesh():
    import os
    from tssplat_amd.stream import StreamTetSpheres
    from oracle import tet_energy_oracle as O
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "aveg_mesh.npz"))
    rest, tets = g["rest"], g["tets"]
    st = StreamTetSpheres(rest.reshape(-1), tets.reshape(-1))
    rng = np.random.default_rng(3)
    x_np = (rest + 0.01 * rng.standard_normal(rest.shape)).astype(np.float32)
    e, gr = st.forward_backward(torch.from_numpy(x_np).cuda(), 2e-4, 2e-4, 2)
    cache = O.prepare(rest, tets)
    E, _, _, gref = O.energy_and_grad(x_np, cache, 2e-4, 2e-4, 2)
    _, tol_g = O.factored_tolerances(x_np, cache, 2e-4, 2e-4, 2)
    assert abs(float(e) - E) <= 1e-5 * abs(E)
    assert np.linalg.norm(gr.cpu().numpy() - gref) <= tol_g
