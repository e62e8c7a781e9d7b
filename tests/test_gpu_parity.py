"""GPU parity tests: the HIP path (through the C ABI / operator surface) against the CPU oracle.

Bar (BASELINE.json north_star): fp32 results within a stated tolerance of the
float64 oracle on identical fp32 inputs.  Two tolerances are asserted everywhere:

* ``factored_tolerances`` (oracle/tet_energy_oracle.py): rtol 1e-5 plus the propagated
  16-ulp error model of a factored fp32 evaluation -- the bar for OUR kernels;
* SURVEY.md 8(c)'s band for the reference formulation, ``1e-5*|E| + 8*eps*c1*A`` and the gradient
  analogue, which the fp32 ``M = G^T L^T L G`` path itself needs;
* a REGRESSION GUARD: both of the above are worst-case bounds that sit 3-4 orders above the measured
  errors (profiles/r01_parity.txt), so a kernel that lost 1000x accuracy, or corrupted one tile, would still
  pass them.  ``oracle.rounding_error_model`` predicts the standard deviation of the fp32 rounding error of
  the energy and of EVERY vertex' gradient by variance propagation; the guard asserts
  ``|dE| <= GUARD_E * std_E``, ``|dg|_2 <= GUARD_G * |std_g|_2`` and, per vertex,
  ``|dg_v| <= GUARD_V * (std_g_v + 1e-3 * rms(std_g))``.  Largest ratios measured on MI355X over this whole suite
  (78 cases, profiles/r02_parity.txt lists every one): energy 6.1, gradient 0.71, worst vertex 6.4 -- the guard
  factors leave 5-11x of head-room, the old tolerances left 1 000-10 000x.  The two full-size tests (configs 3 and 4)
  assert the same three guards through the plain-C twin of the model (``_assert_guards_c``).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

EPS = 2.0 ** -23
# regression guard factors: 7-12x the largest measured ratio err / predicted std (see profiles/r02_parity.txt)
GUARD_E, GUARD_G, GUARD_V = 40.0, 8.0, 30.0


@pytest.fixture(scope="module")
def ext():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from tssplat_amd import tet_spheres_ext
    return tet_spheres_ext


def _oracle():
    from oracle import tet_energy_oracle as O
    return O


def _eval_gpu(ext, ts, x_np, c1, c2, order, go=1.0):
    x = torch.from_numpy(x_np).cuda().requires_grad_(True)
    e = ext.forward(x, ts, c1, c2, order)
    g = ext.backward(torch.tensor(go), x, ts, c1, c2, order)
    return float(e), g.cpu().numpy().astype(np.float64)


def _assert_parity(ext, ts, scene_rest, scene_tets, x_np, c1, c2, order, go=1.0, label="", L=None):
    O = _oracle()
    cache = O.prepare(scene_rest, scene_tets, L=L)
    E, Es, Eb, g = O.energy_and_grad(x_np, cache, c1, c2, order, grad_output=go)
    tol_e, tol_g = O.factored_tolerances(x_np, cache, c1, c2, order)
    A, nMx = O.tolerance_scales(x_np, cache)
    band_e = 1e-5 * abs(E) + 8 * EPS * float(np.float32(c1)) * A + 1e-5 * float(np.float32(c2)) * Eb
    band_g = abs(go) * (1e-5 * np.linalg.norm(g / go) + 8 * EPS * float(np.float32(c1)) * nMx)
    e_gpu, g_gpu = _eval_gpu(ext, ts, x_np, c1, c2, order, go)
    err_e = abs(e_gpu - E)
    err_g = float(np.linalg.norm(g_gpu - g))
    std_e, std_gv = O.rounding_error_model(x_np, cache, c1, c2, order)
    std_g = abs(go) * float(np.sqrt(np.sum(std_gv ** 2)))
    err_v = np.linalg.norm(g_gpu - g, axis=1)
    floor_v = abs(go) * (std_gv + 1e-3 * float(np.sqrt(np.mean(std_gv ** 2))))
    ratio_v = float(np.max(err_v / np.maximum(floor_v, 1e-300))) if err_v.size else 0.0
    if ratio_v > 20 and os.environ.get("TSSPLAT_AMD_PARITY_REPORT"):
        k = int(np.argmax(err_v / np.maximum(floor_v, 1e-300)))
        with open(os.environ["TSSPLAT_AMD_PARITY_REPORT"], "a") as fh:
            fh.write(f"#   outlier vertex {k}: err {err_v[k]:.3e} std {abs(go) * std_gv[k]:.3e} |g_v| {np.linalg.norm(g[k]):.3e} gpu {g_gpu[k]} oracle {g[k]} "
                     f"rms std {abs(go) * float(np.sqrt(np.mean(std_gv ** 2))):.3e} max std {abs(go) * float(std_gv.max()):.3e}\n")
    line = (f"[{label}] E={E:.6e} gpu={e_gpu:.6e} err={err_e:.2e} tol={tol_e:.2e} band={band_e:.2e} | "
            f"|g|={np.linalg.norm(g):.4e} err={err_g:.2e} tol={abs(go) * tol_g:.2e} band={band_g:.2e} | "
            f"guard ratios err/std: E {err_e / max(std_e, 1e-300):.2f} g {err_g / max(std_g, 1e-300):.2f} vertex-max {ratio_v:.2f}")
    print(line)
    if os.environ.get("TSSPLAT_AMD_PARITY_REPORT"):      # profiles/r01_parity.txt is made this way
        with open(os.environ["TSSPLAT_AMD_PARITY_REPORT"], "a") as fh:
            fh.write(line + "\n")
    assert np.isfinite(e_gpu) and np.isfinite(g_gpu).all()
    assert err_e <= tol_e, f"{label}: energy error {err_e:.3e} > factored tolerance {tol_e:.3e}"
    assert err_g <= abs(go) * tol_g, f"{label}: gradient error {err_g:.3e} > factored tolerance {abs(go) * tol_g:.3e}"
    assert err_e <= band_e + tol_e
    assert err_g <= band_g + abs(go) * tol_g
    # regression guard (module docstring): a few standard deviations of the predicted fp32 rounding error
    assert err_e <= GUARD_E * std_e, f"{label}: energy error {err_e:.3e} > {GUARD_E} x predicted std {std_e:.3e}"
    assert err_g <= GUARD_G * std_g, f"{label}: gradient error {err_g:.3e} > {GUARD_G} x predicted std {std_g:.3e}"
    assert ratio_v <= GUARD_V, f"{label}: a vertex is off by {ratio_v:.1f} predicted standard deviations"
    # energy terms individually
    es_gpu, eb_gpu = ts.energy_terms()
    assert abs(es_gpu - Es) <= tol_e / max(float(np.float32(c1)), 1e-30) + 1e-12 + 1e-6 * abs(Es)   # (the last term: c1 == c2 == 0 makes tol_e 0)
    assert abs(eb_gpu - Eb) <= 2e-5 * Eb + 1e-12 + tol_e / max(float(np.float32(c2)), 1e-30)


def _assert_guards_c(label, rest, tets, x_np, c1, c2, order, e_gpu, g_gpu, nbr=None, terms=None):
    """The three regression guards of ``_assert_parity`` at sizes the numpy oracle cannot hold: float64 values from the
    plain-C oracle, predicted fp32 rounding error from its plain-C rounding model (oracle/c: tso_rounding_model, checked
    against the numpy model in tests/test_oracle.py).  Returns the oracle's (E, g)."""
    from oracle import c_oracle
    E, Es, Eb, g = c_oracle.energy_and_grad(rest, tets, x_np, c1, c2, order, nbr=nbr)
    std_e, std_gv = c_oracle.rounding_error_model(rest, tets, x_np, c1, c2, order, nbr=nbr)
    # The energy leaves the library twice: as the fp32 scalar (half an ulp of E on top of the evaluation's own error --
    # at E = 25.9 that is 9.5e-7, fifty times the predicted evaluation error) and as the two double-precision terms the
    # fp32 value is rounded from.  The guard is asserted on the terms, the fp32 value gets the half ulp on top.
    if terms is not None:
        e_terms = float(np.float32(c1)) * terms[0] + float(np.float32(c2)) * terms[1]
        assert abs(e_terms - e_gpu) <= 2.0 ** -24 * abs(e_terms) * 1.0001, (e_terms, e_gpu)
        err_e = abs(e_terms - E)
    else:
        err_e = max(abs(e_gpu - E) - 2.0 ** -24 * abs(E), 0.0)
    err_v = np.linalg.norm(g_gpu - g, axis=1)
    err_g = float(np.sqrt(np.sum(err_v ** 2)))
    std_g = float(np.sqrt(np.sum(std_gv ** 2)))
    floor_v = std_gv + 1e-3 * float(np.sqrt(np.mean(std_gv ** 2)))
    ratio_v = float(np.max(err_v / np.maximum(floor_v, 1e-300)))
    line = (f"[{label}] E={E:.6e} gpu={e_gpu:.6e} err={err_e:.2e} | |g|={np.linalg.norm(g):.4e} err={err_g:.2e} | "
            f"guard ratios err/std (C model): E {err_e / max(std_e, 1e-300):.2f} g {err_g / max(std_g, 1e-300):.2f} vertex-max {ratio_v:.2f}")
    print(line)
    if os.environ.get("TSSPLAT_AMD_PARITY_REPORT"):
        with open(os.environ["TSSPLAT_AMD_PARITY_REPORT"], "a") as fh:
            fh.write(line + "\n")
    assert err_e <= GUARD_E * std_e, f"{label}: energy error {err_e:.3e} > {GUARD_E} x predicted std {std_e:.3e}"
    assert err_g <= GUARD_G * std_g, f"{label}: gradient error {err_g:.3e} > {GUARD_G} x predicted std {std_g:.3e}"
    assert ratio_v <= GUARD_V, f"{label}: a vertex is off by {ratio_v:.1f} predicted standard deviations"
    return E, g


def _replicated_adjacency(sc, S):
    """Face adjacency of S copies of one sphere template from the adjacency of the first copy (the plain-C oracle's
    sort over 84 M faces takes longer than everything else in the full-size test)."""
    from oracle import c_oracle
    mt, nv = sc.n_tets // S, sc.n_vertices // S
    for s in (1, S // 2, S - 1):
        assert np.array_equal(sc.tets[s * mt:(s + 1) * mt] - s * nv, sc.tets[:mt])
    nbr0 = c_oracle.face_adjacency(sc.tets[:mt])
    return np.concatenate([np.where(nbr0 >= 0, nbr0 + s * mt, -1) for s in range(S)]).astype(np.int32)


@pytest.mark.parametrize("sigma", [0.0, 0.02, 0.3])
@pytest.mark.parametrize("order", [2, 4])
def test_config2_64_spheres(ext, sigma, order):
    """BASELINE config 2: 64 x kuhn_ball(8) = 196 608 tets."""
    from tssplat_amd import scenes
    sc = scenes.make_scene("kuhn8", 64)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    assert ts.plan_info()["n_tiles"] >= 64
    x = scenes.deform(sc, sigma)
    mult = 16.0 if order == 4 else 1.0
    _assert_parity(ext, ts, sc.rest, sc.tets, x, 2e-4 / 64 * mult, 2e-4 * mult, order, label=f"kuhn8x64 s={sigma} p={order}")


@pytest.mark.parametrize("kw", [dict(), dict(debug_flags=2), dict(lane_search_sweeps=-1), dict(lane_search_sweeps=4), dict(lds_budget_bytes=40960, max_threads=512),
                                dict(rebuild_dminv=True), dict(rebuild_dminv=True, max_threads=640, lds_budget_bytes=163840),
                                dict(max_threads=768, lds_budget_bytes=163840),      # one workgroup per CU
                                dict(max_threads=384, lds_budget_bytes=40960, target_owned=500)])
@pytest.mark.parametrize("sigma,order", [(0.02, 2), (0.3, 4)])
def test_multi_tile_hires(ext, kw, sigma, order):
    """kuhn_ball(19) spheres need ~16 tiles each: halo slots, staged shared vertices, finish kernel."""
    from tssplat_amd import scenes
    sc = scenes.make_scene("kuhn19", 3)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), **kw)
    info = ts.plan_info()
    assert info["n_tiles"] > 3 and info["shared_vertex_copies"] > 0
    x = scenes.deform(sc, sigma)
    _assert_parity(ext, ts, sc.rest, sc.tets, x, 7e-5, 2e-4, order, go=0.37, label=f"kuhn19x3 {kw} s={sigma} p={order}")


def test_real_mesh_aveg(ext, aveg):
    """The reference's own tet mesh (a.veg, TetWild quality, valence <= 56), replicated 3x."""
    from tssplat_amd import scenes
    rest, tets = aveg
    sc = scenes.replicate_spheres(rest.astype(np.float64), tets, 3, seed=3)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    for sigma, order in [(0.0, 2), (0.01, 2), (0.1, 4)]:
        x = scenes.deform(sc, sigma)
        _assert_parity(ext, ts, sc.rest, sc.tets, x, 6e-5, 2e-4, order, label=f"a.veg x3 s={sigma} p={order}")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_unstructured_delaunay(ext, seed):
    """Random Delaunay balls (sliver-filtered): irregular valence, tets with 1-4 neighbours, multi-tile."""
    from tssplat_amd import scenes
    sc = scenes.make_scene("delaunay3000", 6, seed=seed)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    assert ts.plan_info()["n_tiles"] > 6
    for sigma, order in ((0.02, 2), (0.3, 4)):
        x = scenes.deform(sc, sigma, seed=seed + 10)
        _assert_parity(ext, ts, sc.rest, sc.tets, x, 2e-4 / 6, 2e-4, order, label=f"delaunay seed={seed} s={sigma} o={order}")


def test_cone_hub_vertex(ext):
    """One vertex of valence 1280 (hub of a cone of triangles): its incidence list spans many chunks."""
    from tssplat_amd import scenes
    sc = scenes.make_scene("cone", 5)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    x = scenes.deform(sc, 0.2)
    _assert_parity(ext, ts, sc.rest, sc.tets, x, 1e-4, 2e-4, 2, label="cone x5")


def test_config1_literal_fixture_s1_cone(ext):
    """BASELINE.json config 1 names mesh_data/s.1.obj: the reference's template sphere (tests/golden/s1_sphere.npz, verbatim)
    coned to its centroid -- 2 996 tets around ONE vertex of valence 2 996 -- alone and replicated, order
    2 and 4, mildly and strongly deformed."""
    from tssplat_amd import scenes
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s1_sphere.npz"))
    cv, ct = scenes.cone_sphere(g["vertices"].astype(np.float64), g["faces"])
    assert ct.shape[0] == 2996 and np.bincount(ct.reshape(-1)).max() == 2996
    for n_spheres, sigma, order in ((1, 0.02, 2), (1, 0.3, 4), (6, 0.2, 2)):
        sc = scenes.replicate_spheres(cv, ct, n_spheres)            # radii U(0.05, 0.30), like every scene of SURVEY 8(d)
        ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
        x = scenes.deform(sc, sigma)
        _assert_parity(ext, ts, sc.rest, sc.tets, x, 2e-4 / n_spheres, 2e-4, order, label=f"s1_cone x{n_spheres} sigma {sigma} order {order}")


def test_known_answers_on_gpu(ext):
    from tssplat_amd import scenes
    sc = scenes.make_scene("kuhn8", 2)
    m = sc.n_tets
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    # rest state: E ~ 0 (fp32 roundoff of the factored form, NOT the O(1) garbage of the M form), grad ~ 0
    e, g = _eval_gpu(ext, ts, sc.rest, 1.0, 1.0, 2)
    assert abs(e) < 1e-6 and np.abs(g).max() < 1e-2
    # reflection x -> diag(1,1,-1) x: F = A everywhere, E_s = 0, E_b = m * 1^p, (det A = -1)
    A = np.diag([1.0, 1.0, -1.0])
    xa = (sc.rest.astype(np.float64) @ A.T + 0.25).astype(np.float32)
    for order in (2, 4):
        e, _ = _eval_gpu(ext, ts, xa, 0.0, 1.0, order)
        assert abs(e - m) <= 1e-4 * m
    # any other order switches the penalty off (tet_spheres_cuda.cu:57-63)
    e, g = _eval_gpu(ext, ts, xa, 0.0, 1.0, 3)
    assert e == 0.0 and np.abs(g).max() == 0.0
    # translation invariance
    x = scenes.deform(sc, 0.1)
    e0, g0 = _eval_gpu(ext, ts, x, 1e-4, 2e-4, 2)
    e1, g1 = _eval_gpu(ext, ts, (x + np.float32(0.5)), 1e-4, 2e-4, 2)
    assert abs(e0 - e1) <= 1e-4 * abs(e0)
    assert np.linalg.norm(g0 - g1) <= 1e-3 * np.linalg.norm(g0)


def test_forward_only_equals_fused_and_is_deterministic(ext):
    from tssplat_amd import scenes
    sc = scenes.make_scene("kuhn19", 2)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    x = torch.from_numpy(scenes.deform(sc, 0.3)).cuda()
    e_fwd = [float(ext.forward(x, ts, 1e-4, 2e-4, 2)) for _ in range(3)]
    xg = x.clone().requires_grad_(True)
    e_fused = [float(ext.forward(xg, ts, 1e-4, 2e-4, 2)) for _ in range(3)]
    assert len(set(e_fwd)) == 1 and len(set(e_fused)) == 1       # fixed-order reductions: bitwise repeatable
    assert abs(e_fwd[0] - e_fused[0]) <= 1e-6 * abs(e_fused[0])


def test_autograd_surface_and_cache(ext):
    """SmoothnessBarrierFunc contract (energies/smooth_barrier.py:9-31) and the fused-pass cache."""
    from tssplat_amd import scenes
    from tssplat_amd.energies import SmoothnessBarrierEnergy, SmoothnessBarrierFunc

    class Flags:
        smooth_eng_coeff = 2e-4 / 3
        barrier_coeff = 2e-4
        increase_order_iter = 1000

    sc = scenes.make_scene("kuhn8", 3)
    mod = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags)
    mod._ext = None        # the Python Function and the operator functions' fused cache (the C++ node keeps its gradient in its
                           # own context: test_cpp_autograd_nodes_equal_python_nodes)
    x = torch.nn.Parameter(torch.from_numpy(scenes.deform(sc, 0.3)).cuda())
    c1, c2 = mod.coeff_scheduler(1200)
    assert abs(c1 / Flags.smooth_eng_coeff - 16.0) < 1e-9
    e = mod(x, 1200, c1, c2)                                   # it > 1000 -> order 4
    assert e.dim() == 0 and e.dtype == torch.float32 and e.is_cuda
    assert mod.tet_sp._cache is not None                       # fused pass ran
    loss = 2.5 * e + 1.0
    loss.backward()
    assert mod.tet_sp._cache is None                           # consumed
    assert x.grad.shape == x.shape and x.grad.dtype == torch.float32 and x.grad.device == x.device
    O = _oracle()
    cache = O.prepare(sc.rest, sc.tets)
    E, _, _, g = O.energy_and_grad(x.detach().cpu().numpy(), cache, c1, c2, 4, grad_output=2.5)
    _, tol_g = O.factored_tolerances(x.detach().cpu().numpy(), cache, c1, c2, 4)
    assert np.linalg.norm(x.grad.cpu().numpy() - g) <= 2.5 * tol_g
    # cache miss path: x modified in place between forward and backward -> full backward kernel
    g_cached = x.grad.clone()
    x.grad = None
    e = SmoothnessBarrierFunc.apply(x, mod.tet_sp, c1, c2, 4)
    mod.tet_sp._cache = None
    (2.5 * e).backward()
    assert torch.allclose(x.grad, g_cached, rtol=1e-5, atol=1e-6 * float(g_cached.abs().max()))
    # CPU grad_output (the reference hands a CPU 0-dim tensor, tet_spheres_cuda.cu:194,257)
    g_cpu_go = ext_backward_cpu_go(mod, x, c1, c2)
    assert torch.allclose(g_cpu_go, g_cached, rtol=1e-5, atol=1e-6 * float(g_cached.abs().max()))
    # under torch.no_grad() (logging, validation) no gradient pass runs and nothing is cached
    assert mod.tet_sp._cache is None
    with torch.no_grad():
        e_ng = mod(x, 1200, c1, c2)
    assert mod.tet_sp._cache is None and abs(float(e_ng) - E) <= 1e-5 * abs(E)
    # ... and such a call between `loss = energy(x)` and `loss.backward()` leaves the kept gradient in place
    x.grad = None
    e = mod(x, 1200, c1, c2)
    kept = mod.tet_sp._cache
    with torch.no_grad():
        mod(x, 1200, c1, c2)
    assert kept is not None and mod.tet_sp._cache is kept
    (2.5 * e).backward()
    assert mod.tet_sp._cache is None
    assert torch.allclose(x.grad, g_cached, rtol=1e-5, atol=1e-6 * float(g_cached.abs().max()))
    # reference CPU-energy convention on request
    from tssplat_amd import tet_spheres_ext as _ext
    _ext.CPU_ENERGY = True
    try:
        e_cpu = mod(x, 1200, c1, c2)
        assert not e_cpu.is_cuda and e_cpu.dim() == 0
    finally:
        _ext.CPU_ENERGY = False


def ext_backward_cpu_go(mod, x, c1, c2):
    from tssplat_amd import tet_spheres_ext
    return tet_spheres_ext.backward(torch.tensor(2.5), x.detach(), mod.tet_sp, c1, c2, 4)


def test_veg_constructor_and_helpers(ext, aveg, tmp_path):
    from tssplat_amd import scenes
    rest, tets = aveg
    path = tmp_path / "a.veg"
    scenes.write_veg(path, rest[:, :], tets)
    ts = ext.TetSpheres(str(path))
    assert ts.n == rest.shape[0] and ts.nele == tets.shape[0]
    rx = ext.random_x(ts)
    assert rx.shape == (ts.n, 3) and not rx.is_cuda and rx.min() >= 0 and rx.max() <= 1
    x = scenes.deform(scenes.replicate_spheres(rest.astype(np.float64), tets, 1, seed=0), 0.0)
    # grad_limit: intended semantics (utils/optimizer.py:84-86)
    g = torch.randn(1000, 3, device="cuda")
    g[123, 1] = -50.0
    ref = g.clone() * (2.0 / 50.0)
    ext.grad_limit(g, 10.0, 2.0)
    assert torch.allclose(g, ref, rtol=1e-6, atol=0)
    g2 = torch.randn(100, 3, device="cuda").clamp(-3, 3)
    keep = g2.clone()
    ext.grad_limit(g2, 10.0, 2.0)
    assert torch.equal(g2, keep)


def test_loud_failures(ext):
    from tssplat_amd import scenes
    sc = scenes.make_scene("kuhn2", 1)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    with pytest.raises(RuntimeError):
        ext.forward(torch.from_numpy(sc.rest), ts, 1.0, 1.0, 2)            # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ext.forward(torch.zeros(5, 3, device="cuda"), ts, 1.0, 1.0, 2)     # wrong size
    empty = ext.TetSpheres(sc.rest, sc.tets.reshape(-1))                   # 2-D vertices -> empty object
    with pytest.raises(RuntimeError):
        ext.forward(torch.from_numpy(sc.rest).cuda(), empty, 1.0, 1.0, 2)


def test_config3_one_million_tets_properties(ext):
    """BASELINE config 3 size (256 x kuhn8 = 786 432 tets): C oracle parity + block additivity."""
    from tssplat_amd import scenes
    from oracle import c_oracle
    sc = scenes.make_scene("kuhn8", 256)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    x = scenes.deform(sc, 0.3)
    c1, c2 = 2e-4 / 256, 2e-4
    e, g = _eval_gpu(ext, ts, x, c1, c2, 2)
    # the same three guards as _assert_parity (energy, gradient norm, worst vertex against the predicted fp32 rounding error)
    E, go = _assert_guards_c("config3 kuhn8x256 s=0.3 p=2", sc.rest, sc.tets, x, c1, c2, 2, e, g, terms=ts.energy_terms())
    assert abs(e - E) <= 2e-5 * abs(E)
    assert np.linalg.norm(g - go) <= 5e-4 * np.linalg.norm(go)
    # additivity over spheres = the sharding invariant: energy of the first half + second half
    half_v, half_t = sc.n_vertices // 2, sc.n_tets // 2
    ta = ext.TetSpheres(sc.rest[:half_v].reshape(-1), sc.tets[:half_t].reshape(-1))
    tb = ext.TetSpheres(sc.rest[half_v:].reshape(-1), (sc.tets[half_t:] - half_v).reshape(-1))
    ea, ga = _eval_gpu(ext, ta, x[:half_v], c1, c2, 2)
    eb, gb = _eval_gpu(ext, tb, x[half_v:], c1, c2, 2)
    assert abs((ea + eb) - e) <= 1e-5 * abs(e)
    assert np.linalg.norm(np.concatenate([ga, gb]) - g) <= 1e-5 * np.linalg.norm(g)


def test_config4_full_size_properties(ext):
    """BASELINE config 4 / the bench workload (512 x kuhn19 = 21 070 848 tets): C-oracle parity at full
    size, plus size-independent properties: bit-identical repeats, zero net force and zero net torque per
    sphere (the energy is invariant under rigid motions), and a checksum of per-sphere checksums."""
    from tssplat_amd import scenes
    from oracle import c_oracle
    S = 512
    sc = scenes.make_scene("kuhn19", S)
    assert sc.n_tets == 21_070_848 and sc.n_vertices == 4_096_000
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    x = scenes.deform(sc, 0.02)
    c1, c2 = 2e-4 / S, 2e-4
    e, g = _eval_gpu(ext, ts, x, c1, c2, 2)
    e2, g2 = _eval_gpu(ext, ts, x, c1, c2, 2)
    terms = ts.energy_terms()
    assert e == e2 and np.array_equal(g, g2), "evaluation must be deterministic (fixed reduction order)"
    # the same three guards as _assert_parity, at full size: a single corrupted vertex among 4 096 000 fails the third
    E, go = _assert_guards_c("config4 kuhn19x512 s=0.02 p=2", sc.rest, sc.tets, x, c1, c2, 2, e, g, nbr=_replicated_adjacency(sc, S), terms=terms)
    assert abs(e - E) <= 2e-5 * abs(E)
    assert np.linalg.norm(g - go) <= 5e-4 * np.linalg.norm(go)
    # per-sphere slices: same bound sphere by sphere (a single wrong tile cannot hide in the global norm)
    nv = sc.n_vertices // S
    gs, gos = g.reshape(S, nv, 3), go.reshape(S, nv, 3)
    err = np.linalg.norm((gs - gos).reshape(S, -1), axis=1)
    ref = np.linalg.norm(gos.reshape(S, -1), axis=1)
    assert np.all(err <= 2e-3 * ref + 1e-12), f"worst sphere: {np.max(err / ref):.2e}"
    # rigid-motion invariance: net force and net torque of every sphere vanish
    xs = x.astype(np.float64).reshape(S, nv, 3)
    scale = np.abs(gs).sum(axis=1).max(axis=1)                           # per-sphere sum of |force|
    assert np.all(np.abs(gs.sum(axis=1)).max(axis=1) <= 1e-4 * scale)
    torque = np.cross(xs - xs.mean(axis=1, keepdims=True), gs).sum(axis=1)
    assert np.all(np.abs(torque).max(axis=1) <= 1e-4 * scale)
    # checksum of checksums against the oracle
    assert abs(gs.sum(axis=(1, 2)).sum() - gos.sum(axis=(1, 2)).sum()) <= 1e-4 * np.abs(gos).sum() / np.sqrt(go.size)


def test_aveg_full_size_properties(ext):
    """The reference's own mesh class at the headline's size: 952 copies of tssplat_ext/a.veg (TetWild
    quality, valence <= 56) = 21 058 240 tets.  The same guards as config 4: C-oracle parity at full size (energy guard,
    gradient L2 guard, per-vertex guard), bit-identical repeats, zero net force / torque per sphere, checksum of checksums."""
    from tssplat_amd import scenes
    S = 952
    sc = scenes.make_scene("aveg", S)
    assert sc.n_tets == 952 * 22120 and sc.n_vertices == 952 * 4500
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    x = scenes.deform(sc, 0.02)
    c1, c2 = 2e-4 / S, 2e-4
    e, g = _eval_gpu(ext, ts, x, c1, c2, 2)
    e2, g2 = _eval_gpu(ext, ts, x, c1, c2, 2)
    terms = ts.energy_terms()
    assert e == e2 and np.array_equal(g, g2), "evaluation must be deterministic (fixed reduction order)"
    E, go = _assert_guards_c("aveg x952 s=0.02 p=2", sc.rest, sc.tets, x, c1, c2, 2, e, g, nbr=_replicated_adjacency(sc, S), terms=terms)
    assert abs(e - E) <= 2e-5 * abs(E)
    assert np.linalg.norm(g - go) <= 5e-4 * np.linalg.norm(go)
    nv = sc.n_vertices // S
    gs, gos = g.reshape(S, nv, 3), go.reshape(S, nv, 3)
    err = np.linalg.norm((gs - gos).reshape(S, -1), axis=1)
    ref = np.linalg.norm(gos.reshape(S, -1), axis=1)
    assert np.all(err <= 2e-3 * ref + 1e-12), f"worst sphere: {np.max(err / ref):.2e}"
    xs = x.astype(np.float64).reshape(S, nv, 3)
    scale = np.abs(gs).sum(axis=1).max(axis=1)
    assert np.all(np.abs(gs.sum(axis=1)).max(axis=1) <= 1e-4 * scale)
    torque = np.cross(xs - xs.mean(axis=1, keepdims=True), gs).sum(axis=1)
    assert np.all(np.abs(torque).max(axis=1) <= 1e-4 * scale)
    assert abs(gs.sum(axis=(1, 2)).sum() - gos.sum(axis=(1, 2)).sum()) <= 1e-4 * np.abs(gos).sum() / np.sqrt(go.size)


def test_replicated_sphere_beyond_4_gib(ext):
    """64-bit addressing at scale: 1 700 copies of ONE deformed kuhn19 sphere (69.96 M tets, plan planes > 4 GiB, staging rows,
    finish lists and tile blobs addressed far beyond 2^32 bytes).  Every copy is tiled identically (the plan is a function of
    the rest geometry), so every copy's gradient must be BIT-identical to the single-sphere evaluation's, and the energy
    S times the single sphere's.  (profiles/r04_bench_kuhn19x8192.json runs 337 M tets / 29 GB the same way.)"""
    from tssplat_amd import scenes
    S = 1700
    sc = scenes.make_scene("kuhn19", 1)
    n1, m1 = sc.n_vertices, sc.n_tets
    x1 = scenes.deform(sc, 0.3)                       # many inverted tets: both energy terms live
    # (explicit max_threads = the default value: a lone sphere would otherwise get the small-batch tiling -- smaller tiles, another
    # summation order -- while the 1 700 copies get the large-batch one)
    ts1 = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), max_threads=768)
    c1, c2 = 2e-4, 2e-4
    e1, g1 = _eval_gpu(ext, ts1, x1, c1, c2, 4)
    rest = np.tile(sc.rest, (S, 1))
    tets = (sc.tets[None].astype(np.int64) + (np.arange(S, dtype=np.int64) * n1)[:, None, None]).reshape(-1, 4).astype(np.int32)
    ts = ext.TetSpheres(rest.reshape(-1), tets.reshape(-1))
    info = ts.plan_info()
    assert info["n_tets"] == S * m1 and info["n_components"] == S
    assert info["total_slots"] * 4 * info["n_planes"] > 2 ** 32 and info["device_bytes"] > 2 ** 32
    del rest, tets
    x = torch.from_numpy(np.tile(x1, (S, 1))).cuda()
    e = ext.forward(x, ts, c1, c2, 4)
    g = ext.backward(torch.tensor(1.0), x, ts, c1, c2, 4)
    assert abs(float(e) - S * e1) <= 2e-6 * abs(S * e1)
    gs = g.reshape(S, n1, 3)
    ref = torch.from_numpy(g1.astype(np.float32)).cuda()
    assert torch.equal(gs[0], ref), "the first copy differs from the single-sphere evaluation"
    same = (gs == ref[None]).all(dim=2).all(dim=1)
    assert bool(same.all()), f"copies {torch.nonzero(~same).flatten()[:8].tolist()} differ from the first"


def test_degenerate_inputs(ext):
    """Empty batches and vertices no tet references: energy 0, gradient 0 (never uninitialised)."""
    from tssplat_amd import scenes
    empty = ext.TetSpheres(np.zeros(0, np.float32), np.zeros(0, np.int32))
    assert empty.n == 0 and empty.nele == 0
    e = ext.forward(torch.zeros(0, 3, device="cuda"), empty, 1.0, 1.0, 2)
    assert float(e) == 0.0
    v, t = scenes.kuhn_ball(2)
    rest = np.concatenate([v, [[5.0, 5.0, 5.0], [6.0, 6.0, 6.0]]]).astype(np.float32)
    ts = ext.TetSpheres(rest.reshape(-1), t.reshape(-1))
    x = torch.from_numpy(rest + 0.05).cuda().requires_grad_(True)
    x.data[:27] += 0.1 * torch.randn(27, 3, device="cuda")
    e = ext.forward(x, ts, 1.0, 1.0, 2)
    g = ext.backward(torch.tensor(1.0), x, ts, 1.0, 1.0, 2)
    assert torch.isfinite(e) and torch.isfinite(g).all()
    assert torch.all(g[-2:] == 0)
    # poisoned output buffers must be fully overwritten by the non-cached backward as well
    g2 = ext.backward(torch.tensor(1.0), x.detach(), ts, 1.0, 1.0, 2)
    assert torch.allclose(g, g2, rtol=1e-6, atol=1e-7)


def test_random_tiling_options_on_gpu(ext):
    """Block sizes, LDS budgets (one or two workgroups per CU) and tiling options against the oracle on small mixed
    meshes -- a fixed pseudo-random sample of the option space."""
    from tssplat_amd import scenes
    rng = np.random.default_rng(2024)
    kinds = ["kuhn4", "kuhn8", "delaunay400", "delaunay1500", "cone"]
    ran = 0
    # (TSSPLAT_AMD_SOAK=n: n trials instead of 16 -- a one-off soak of the option space, lane layouts 3 and 4 included from trial 16 on)
    trials = int(os.environ.get("TSSPLAT_AMD_SOAK", "16"))
    for trial in range(trials):
        kind = kinds[trial % len(kinds)]
        kw = dict(lds_budget_bytes=int(rng.choice([0, 24000, 40960, 65536, 81920, 120000, 163840])),
                  max_threads=int(rng.choice([0, 128, 256, 512, 640, 768])),
                  rebuild_dminv=bool(rng.integers(4) == 0),
                  target_owned=int(rng.choice([0, 200, 900])), debug_flags=int(rng.integers(4)),
                  lane_search_sweeps=int(rng.choice([0, 0, -1, 1, 3])))
        if trial >= 16 and rng.integers(3) == 0:
            kw.update(slots_per_thread=int(rng.choice([3, 4])), rebuild_dminv=False)
            if kw["slots_per_thread"] == 3 and rng.integers(2):
                kw["max_threads"] = 1024
        sc = scenes.make_scene(kind, int(rng.integers(1, 4)), seed=trial)
        try:
            ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), **kw)
        except RuntimeError as e:
            assert "LDS budget" in str(e) or "tiled" in str(e) or "max_threads exceeds" in str(e), str(e)
            continue
        x = scenes.deform(sc, float(rng.choice([0.02, 0.3])), seed=trial + 50)
        _assert_parity(ext, ts, sc.rest, sc.tets, x, 2e-4 / sc.n_spheres, 2e-4, int(rng.choice([2, 4])),
                       go=float(rng.choice([1.0, 0.25])), label=f"random#{trial} {kind} {kw}")
        ran += 1
    assert ran >= min(10, trials)


# Every lane layout tile_kernel_for() hands out besides the default (kernels.hip): 3 slots per lane x 512 threads (two workgroups
# per CU), 3 x 1 024 and 4 x 768 (one workgroup per CU, up to 160 KiB).  VERDICT r5 / ADVICE r5: these six instantiations were
# only covered by the CPU plan replay.  Device-side differences against the default: plane loads of SPT dwords at a 12- / 16-byte lane
# stride, 1 024-thread blocks (16 waves in the block reduction), the reversed vertex blocks of waves 4-7 at other wave counts,
# 128 / 168-VGPR launch bounds.
_LANE_LAYOUTS = [dict(slots_per_thread=3, max_threads=512), dict(slots_per_thread=3, max_threads=512, lds_budget_bytes=54400),
                 dict(slots_per_thread=3, max_threads=1024, lds_budget_bytes=163840),
                 dict(slots_per_thread=4, max_threads=768, lds_budget_bytes=163840),
                 dict(slots_per_thread=4, max_threads=768), dict(slots_per_thread=3, max_threads=1024)]


@pytest.mark.parametrize("kw", _LANE_LAYOUTS, ids=lambda kw: "spt%d_%d_%d" % (kw["slots_per_thread"], kw["max_threads"], kw.get("lds_budget_bytes", 0)))
@pytest.mark.parametrize("kind,S", [("kuhn19", 3), ("delaunay3000", 3)])
def test_lane_layouts_multi_tile(ext, kw, kind, S):
    """slots_per_thread 3 and 4 on multi-tile scenes (halo slots, staged shared vertices, finish kernel): order 2 and 4, fused
    forward + backward AND the forward-only kernels, against the oracle."""
    from tssplat_amd import scenes
    sc = scenes.make_scene(kind, S, seed=5)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), **kw)
    info = ts.plan_info()
    assert info["slots_per_thread"] == kw["slots_per_thread"] and info["n_tiles"] >= S
    assert info["block_threads"] <= kw["max_threads"] and info["n_planes"] == 13
    for sigma, order, go in ((0.02, 2, 1.0), (0.3, 4, 0.37)):
        x = scenes.deform(sc, sigma, seed=11)
        _assert_parity(ext, ts, sc.rest, sc.tets, x, 2e-4 / S, 2e-4, order, go=go, label=f"{kind}x{S} {kw} s={sigma} p={order}")
        # the forward-only instantiation <false, ...> of the same layout: same energy as the fused pass (fixed-order sums)
        xt = torch.from_numpy(x).cuda()
        e_fwd = float(ext.forward(xt, ts, 2e-4 / S, 2e-4, order, fuse=False))
        e_fused = float(ext.forward(xt.clone().requires_grad_(True), ts, 2e-4 / S, 2e-4, order))
        assert abs(e_fwd - e_fused) <= 1e-6 * abs(e_fused), (e_fwd, e_fused)


@pytest.mark.parametrize("kw", [dict(slots_per_thread=3, max_threads=1024, lds_budget_bytes=163840),
                                dict(slots_per_thread=4, max_threads=768, lds_budget_bytes=163840)],
                         ids=["spt3_1024", "spt4_768"])
def test_lane_layouts_whole_sphere_tiles(ext, kw):
    """With 160 KiB a 3 072-tet sphere is ONE tile: no halo (slots = tets), no shared vertices, no staging rows -- the finish
    kernel only reduces the energy partials.  Also the hub-vertex fixture (a vertex split into copies of <= 64 slots INSIDE one
    tile goes through the staging rows even then)."""
    from tssplat_amd import scenes
    sc = scenes.make_scene("kuhn8", 8)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), **kw)
    info = ts.plan_info()
    assert info["n_tiles"] == 8 and info["total_slots"] == sc.n_tets and info["shared_vertex_copies"] == 0, info
    for sigma, order in ((0.0, 2), (0.02, 2), (0.3, 4)):
        x = scenes.deform(sc, sigma)
        _assert_parity(ext, ts, sc.rest, sc.tets, x, 2e-4 / 8, 2e-4, order, label=f"kuhn8x8 one tile per sphere {kw} s={sigma} p={order}")
    cone = scenes.make_scene("cone", 3)
    tc = ext.TetSpheres(cone.rest.reshape(-1), cone.tets.reshape(-1), **kw)
    _assert_parity(ext, tc, cone.rest, cone.tets, scenes.deform(cone, 0.2), 1e-4, 2e-4, 2, go=0.5, label=f"cone x3 {kw}")


@pytest.mark.parametrize("c1,c2", [(0.0, 2e-4), (2e-4, 0.0), (1e-30, 2e-4), (0.0, 0.0)], ids=["c1=0", "c2=0", "c1=1e-30", "both=0"])
@pytest.mark.parametrize("order", [2, 4])
def test_unfactored_coefficient_branch_gradient(ext, c1, c2, order):
    """Pass 3 works with c2 / c1 and applies c1 once per vertex -- except where the ratio is not a well-behaved fp32 number (c1 == 0:
    only the penalty is left; |c2 / c1| > 2^40; NaN for 0 / 0), where c1 and c2 are applied separately and the vertex sums are written
    unscaled (kernels.hip: `factored`, q_scale).  VERDICT r5: that branch's GRADIENT was never compared with the oracle.  Multi-tile
    scene with many inverted tets, both coefficient routes: launch arguments and the device-side coefficient buffer of a graph replay."""
    from tssplat_amd import scenes
    sc = scenes.make_scene("kuhn19", 2)
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1))
    assert ts.plan_info()["shared_vertex_copies"] > 0
    x = scenes.deform(sc, 0.3)
    O = _oracle()
    cache = O.prepare(sc.rest, sc.tets)
    E, Es, Eb, g = O.energy_and_grad(x, cache, c1, c2, order, grad_output=0.6)
    assert Eb > 0 and Es > 0
    _assert_parity(ext, ts, sc.rest, sc.tets, x, c1, c2, order, go=0.6, label=f"unfactored c1={c1} c2={c2} p={order}")
    # the same through the two other coefficient routes: a HIP-graph replay (coefficients = node arguments, ratio divided on the
    # host) and tsamd_evaluate_dev_coef (coefficients read from a device buffer: the kernel divides c2 / c1 itself)
    import ctypes as C
    from tssplat_amd import _capi
    from tssplat_amd.energies import SmoothnessBarrierEnergy
    from tssplat_amd.energies.graphed import GraphedSmoothnessBarrier

    class Flags:
        smooth_eng_coeff, barrier_coeff, increase_order_iter = c1, c2, 1000

    mod = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags)
    xt = torch.from_numpy(x).cuda()
    gr = GraphedSmoothnessBarrier(mod, xt, grad_scale=0.6)
    e_rep, g_rep = gr.evaluate(c1, c2, order)
    coef = torch.tensor([c1, c2], dtype=torch.float32, device="cuda")
    go = torch.tensor([0.6], dtype=torch.float32, device="cuda")
    e_dev = torch.zeros((), dtype=torch.float32, device="cuda")
    g_dev = torch.full_like(xt, float("nan"))
    _capi.check(_capi.load().tsamd_evaluate_dev_coef(mod.tet_sp._handle(), xt.data_ptr(), go.data_ptr(), coef.data_ptr(), order, None,
                                                     e_dev.data_ptr(), g_dev.data_ptr()))
    torch.cuda.synchronize()
    for tag, e_k, g_k in (("replay", e_rep, g_rep), ("dev_coef", e_dev, g_dev)):
        g_k = g_k.cpu().numpy().astype(np.float64)
        assert abs(float(e_k) - E) <= 1e-5 * abs(E) + 1e-12, tag
        assert np.linalg.norm(g_k - g) <= 2e-5 * np.linalg.norm(g) + 1e-12, tag
        if c1 == 0.0 and c2 == 0.0:
            assert float(e_k) == 0.0 and not g_k.any(), tag


def test_unreferenced_vertices_on_gpu(ext):
    """Vertices no tet references (the FIRST vertex of the array among them: its finish entry has zero staging rows at offset 0,
    ADVICE r5) get a zero gradient from the finish kernel; everything else matches the oracle."""
    from tssplat_amd import scenes
    sc = scenes.make_scene("kuhn8", 2)
    pad = np.array([[9.0, 9.0, 9.0]], np.float32)
    rest = np.concatenate([pad, sc.rest, pad, pad]).astype(np.float32)
    tets = (sc.tets + 1).astype(np.int32)
    ts = ext.TetSpheres(rest.reshape(-1), tets.reshape(-1))
    x = np.concatenate([pad + 1, scenes.deform(sc, 0.3), pad - 2, pad]).astype(np.float32)
    e, g = _eval_gpu(ext, ts, x, 1e-4, 2e-4, 2, go=0.5)
    O = _oracle()
    E, _, _, g64 = O.energy_and_grad(x, O.prepare(rest, tets), 1e-4, 2e-4, 2, grad_output=0.5)
    assert not g[0].any() and not g[-2:].any()
    assert abs(e - E) <= 1e-5 * abs(E) and np.linalg.norm(g - g64) <= 1e-5 * np.linalg.norm(g64)


@pytest.mark.parametrize("kind,S,kw", [("kuhn8", 8, {}), ("kuhn19", 2, {}), ("kuhn19", 2, dict(max_threads=768, lds_budget_bytes=81920)),
                                       ("kuhn19", 2, dict(max_threads=512, lds_budget_bytes=54400)),
                                       ("delaunay3000", 3, {})])
def test_explicit_operator_parity(ext, kind, S, kw):
    """tsamd_create_with_operator: the element operator L as data.  The row-scaled umbrella
    (non-symmetric: the gradient needs L^T), random weights, and the assumed uniform operator passed explicitly
    (must agree with the built-in path to rounding)."""
    from tssplat_amd import scenes
    O = _oracle()
    sc = scenes.make_scene(kind, S)
    nbr = O.face_adjacency(sc.tets)
    rng = np.random.default_rng(11)
    import scipy.sparse as sp
    m = sc.n_tets
    rows = np.repeat(np.arange(m), 4)
    cols = nbr.ravel()
    ok = cols >= 0
    Lr = (sp.diags(rng.uniform(1.0, 5.0, m)) + sp.csr_matrix((rng.uniform(-2.0, -0.2, ok.sum()), (rows[ok], cols[ok])), shape=(m, m))).tocsr()
    Ls = (sp.diags(Lr.diagonal()) + 0.5 * ((Lr - sp.diags(Lr.diagonal())) + (Lr - sp.diags(Lr.diagonal())).T)).tocsr()   # symmetric, random weights
    for name, L in (("scaled", O.element_laplacian_scaled(nbr)), ("random", Lr), ("symmetric", Ls), ("uniform", O.element_laplacian(nbr))):
        ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), operator=L, **kw)
        assert ts.plan_info()["n_planes"] == (18 if name in ("uniform", "symmetric") else 22)    # a symmetric operator stores its weights once
        for sigma, order in ((0.02, 2), (0.3, 4)):
            x = scenes.deform(sc, sigma)
            _assert_parity(ext, ts, sc.rest, sc.tets, x, 2e-4 / S, 2e-4, order, go=0.5, L=L,
                           label=f"operator={name} {kind}x{S} {kw} s={sigma} p={order}")
    # explicit uniform == built-in
    ts_d = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), **kw)
    x = scenes.deform(sc, 0.1)
    e_d, g_d = _eval_gpu(ext, ts_d, x, 2e-4 / S, 2e-4, 2)
    e_x, g_x = _eval_gpu(ext, ts, x, 2e-4 / S, 2e-4, 2)
    assert abs(e_d - e_x) <= 2e-6 * abs(e_d) and np.linalg.norm(g_d - g_x) <= 2e-6 * np.linalg.norm(g_d)


def test_reference_spelling_of_the_autograd_function(ext):
    """The Function exactly as /root/reference/energies/smooth_barrier.py:9-31 spells it -- `forward` WITHOUT ctx +
    `setup_context`, saved tensor `x_cur`, constants on ctx, `int(order)` on the way back -- executed on the GPU
    over this repo's `tet_spheres_ext`, in the reference's return convention (CPU 0-dim energy, so the
    grad_output that comes back is a CPU 0-dim tensor as well, tet_spheres_cuda.cu:194,257)."""
    from tssplat_amd import scenes
    from tet_spheres import tet_spheres_ext                     # the reference's import path (smooth_barrier.py:6)

    class SmoothnessBarrierFunc(torch.autograd.Function):
        @staticmethod
        def forward(x_cur, tet_sp, c1, c2, order):
            return tet_spheres_ext.forward(x_cur, tet_sp, c1, c2, order)

        @staticmethod
        def setup_context(ctx, inputs, output):
            x_cur, tet_sp, c1, c2, order = inputs
            ctx.save_for_backward(x_cur)
            ctx.constants = (tet_sp, c1, c2, order)

        @staticmethod
        def backward(ctx, grad_output):
            if grad_output is None:
                return None, None, None, None, None
            x_cur, = ctx.saved_tensors
            tet_sp, c1, c2, order = ctx.constants
            grad_final = tet_spheres_ext.backward(grad_output, x_cur, tet_sp, c1, c2, int(order))
            return grad_final, None, None, None, None

    sc = scenes.make_scene("kuhn8", 5)
    v_flat = sc.rest.flatten().astype(np.float32)                # smooth_barrier.py:38-40
    f_flat = sc.tets.flatten().astype(np.int32)
    tet_sp = tet_spheres_ext.TetSpheres(v_flat, f_flat)
    x_np = scenes.deform(sc, 0.3)
    O = _oracle()
    cache = O.prepare(sc.rest, sc.tets)
    c1, c2 = 2e-4 / 5, 2e-4
    prev = tet_spheres_ext.CPU_ENERGY
    tet_spheres_ext.CPU_ENERGY = True
    try:
        for order in (2, 4):
            x = torch.nn.Parameter(torch.from_numpy(x_np).cuda())
            e = SmoothnessBarrierFunc.apply(x, tet_sp, c1, c2, order)
            assert e.dim() == 0 and e.dtype == torch.float32 and not e.is_cuda          # .cu:194
            img_loss = (x * x).sum() * 0.0 + 1.0                                        # a CUDA scalar, as trainer.py:115 adds
            loss = img_loss * 100 + 1.7 * e
            loss.backward()
            E, _, _, g = O.energy_and_grad(x_np, cache, c1, c2, order, grad_output=1.7)
            std_e, std_gv = O.rounding_error_model(x_np, cache, c1, c2, order)
            assert abs(float(e) - E) <= GUARD_E * std_e
            assert np.linalg.norm(x.grad.cpu().numpy() - g) <= GUARD_G * 1.7 * float(np.sqrt(np.sum(std_gv ** 2)))
            assert x.grad.shape == x.shape and x.grad.device == x.device
    finally:
        tet_spheres_ext.CPU_ENERGY = prev


def test_graph_replay_equals_eager(ext):
    """GraphedSmoothnessBarrier (HIP-graph replay, coefficients read on the device) against the eager autograd route
    over the reference's schedule: changing coefficients, the order switch at increase_order_iter, in-place updates
    of the parameter between replays.  Same kernels on the same inputs: the energy is bitwise equal."""
    from tssplat_amd import scenes
    from tssplat_amd.energies import SmoothnessBarrierEnergy, GraphedSmoothnessBarrier

    class Flags:
        smooth_eng_coeff = 2e-4 / 6
        barrier_coeff = 2e-4
        increase_order_iter = 1000

    sc = scenes.make_scene("kuhn8", 6)
    mod = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags)
    x = torch.nn.Parameter(torch.from_numpy(scenes.deform(sc, 0.3)).cuda())
    graphed = GraphedSmoothnessBarrier(mod, x, grad_scale=0.75)
    for it in (0, 1, 600, 600, 1000, 1001, 1500, 3):
        with torch.no_grad():
            x.add_(0.001 * torch.randn_like(x))                      # same storage, new values
        e_g, g_g = graphed.step(it)
        e_g, g_g = e_g.clone(), g_g.clone()
        x.grad = None
        c1, c2 = mod.coeff_scheduler(it)
        e = mod(x, it, c1, c2)
        (0.75 * e).backward()
        assert float(e_g) == float(e.detach()), (it, float(e_g), float(e.detach()))
        # (the gradient is not bitwise equal: the replay applies grad_output inside the kernel, as one factor c1 * 0.75,
        # the eager route scales by c1 in the kernel and by 0.75 in tsamd_scale)
        assert torch.allclose(g_g, x.grad, rtol=3e-7, atol=0), it
    assert sorted(graphed._graphs) == [2, 4]


def test_graph_replay_through_autograd(ext):
    """``SmoothnessBarrierEnergy(graph=True)``: code shaped like the reference trainer (``loss = a + w * energy(x, it, c1,
    c2)``, ``loss.backward()``, /root/reference/trainer.py:94-130) gets the HIP-graph replay through an autograd node
    (VERDICT r2 item 2).  Against the eager module over the reference's schedule, with a second loss term on the same
    parameter, in-place parameter updates between steps, ``torch.no_grad()`` evaluations and a stale-backward check."""
    from tssplat_amd import scenes
    from tssplat_amd.energies import SmoothnessBarrierEnergy

    class Flags:
        smooth_eng_coeff = 2e-4 / 6
        barrier_coeff = 2e-4
        increase_order_iter = 1000

    sc = scenes.make_scene("kuhn8", 6)
    eager = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags)
    graphed = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags, graph=True)
    x0 = torch.from_numpy(scenes.deform(sc, 0.3)).cuda()
    xe, xg = torch.nn.Parameter(x0.clone()), torch.nn.Parameter(x0.clone())
    w = torch.randn_like(x0)
    for it in (0, 1, 600, 600, 1000, 1001, 1500, 3):
        with torch.no_grad():
            d = 0.001 * torch.randn_like(x0)
            xe.add_(d)
            xg.add_(d)
        c1, c2 = eager.coeff_scheduler(it)
        out = []
        for mod, x in ((eager, xe), (graphed, xg)):
            x.grad = None
            e = mod(x, it, c1, c2)
            loss = (x * w).sum() + 0.75 * e                           # a second term on the same parameter
            loss.backward()
            out.append((float(e.detach()), x.grad.clone()))
        assert out[0][0] == out[1][0], (it, out[0][0], out[1][0])     # same kernels, same inputs: bitwise-equal energy
        assert torch.allclose(out[0][1], out[1][1], rtol=3e-7, atol=1e-6 * float(out[0][1].abs().max())), it
    assert sorted(graphed._graphed._graphs) == [2, 4]
    assert xg.grad.data_ptr() != graphed._graphed.grad.data_ptr()     # x.grad never aliases the static gradient buffer
    # energy-only evaluations take the eager energy kernel and leave the replay state alone
    with torch.no_grad():
        e_ng = graphed(xg, 3, c1, c2)
    assert abs(float(e_ng) - out[1][0]) <= 1e-6 * abs(out[1][0])
    # a backward whose static gradient was overwritten by a newer evaluation must not return the wrong gradient silently
    e_old = graphed(xg, 3, c1, c2)
    graphed(xg, 3, c1, c2)
    with pytest.raises(RuntimeError, match="newer evaluation"):
        e_old.backward()
    # a new parameter tensor (other storage) re-captures
    x2 = torch.nn.Parameter(x0.clone())
    e2 = graphed(x2, 3, c1, c2)
    e2.backward()
    e3 = eager(xe, 3, c1, c2)
    assert graphed._graphed.x.data_ptr() == x2.data_ptr() and x2.grad is not None and e3 is not None


def test_compiled_c_consumer_on_device(tmp_path):
    """tests/c/abi_device.c: a compiled C99 program -- no Python, torch or ctypes between it and the library -- drives
    the device entry points of include/tssplat_amd.h and checks them against the C oracle."""
    import subprocess
    from oracle import c_oracle
    from tssplat_amd import _capi
    _capi.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib, ora = _capi.lib_path(), c_oracle.build()
    hip = "/opt/rocm/lib"
    exe = tmp_path / "abi_device"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I", os.path.join(root, "include"),
           "-I", "/opt/rocm/include", os.path.join(root, "tests", "c", "abi_device.c"), "-o", str(exe),
           "-L", os.path.dirname(lib), "-ltssplat_amd", "-L", os.path.dirname(ora), "-ltet_energy_oracle",
           "-L", hip, "-lamdhip64", "-lm", f"-Wl,-rpath,{os.path.dirname(lib)}", f"-Wl,-rpath,{os.path.dirname(ora)}",
           f"-Wl,-rpath,{hip}"]
    built = subprocess.run(cmd, capture_output=True, text=True)
    assert built.returncode == 0, built.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, f"exit {out.returncode}: {out.stdout[-2000:]} {out.stderr[-2000:]}"
    assert "abi device ok" in out.stdout


def test_fused_train_loop_equals_eager(ext):
    """``FusedEnergyAdamLoop``: n iterations of ``energy(x, it, *coeff_scheduler(it)).backward(); optimizer.step()``
    (/root/reference/trainer.py:130-133 for a loss that is the energy alone) replayed from ONE graph launch give the same
    parameters, moments, energies and optimiser counters as the eager loop -- bit for bit: the same kernels on the same inputs.
    Crosses the order switch (increase_order_iter) and a grad_limit stage boundary."""
    from tssplat_amd import scenes
    from tssplat_amd.energies import FusedEnergyAdamLoop, SmoothnessBarrierEnergy
    from tssplat_amd.utils.optimizer import AdamUniform

    class Flags:
        smooth_eng_coeff = 2e-4
        barrier_coeff = 2e-4
        increase_order_iter = 10

    sc = scenes.make_scene("kuhn8", 6)
    x0 = torch.from_numpy(scenes.deform(sc, 0.25)).cuda()
    kw = dict(lr=0.05, grad_limit=True, grad_limit_values=[0.01, 0.004], grad_limit_iters=[13])
    # eager
    mod_e = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags)
    xe = torch.nn.Parameter(x0.clone())
    opt_e = AdamUniform([xe], **kw)
    energies_e = []
    for it in range(24):
        c1, c2 = mod_e.coeff_scheduler(it)
        opt_e.zero_grad()
        e = mod_e(xe, it, c1, c2)
        e.backward()
        opt_e.step()
        energies_e.append(float(e.detach()))
    # fused: three launches of eight iterations
    mod_f = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags)
    xf = torch.nn.Parameter(x0.clone())
    opt_f = AdamUniform([xf], **kw)
    loop = FusedEnergyAdamLoop(mod_f, xf, opt_f, n_iters=8)
    energies_f = []
    for first in (0, 8, 16):
        energies_f += loop.run(first).cpu().tolist()
    assert energies_f == energies_e
    assert torch.equal(xf.data, xe.data)
    assert torch.equal(opt_f.state[xf]["g1"], opt_e.state[xe]["g1"]) and torch.equal(opt_f.state[xf]["g2"], opt_e.state[xe]["g2"])
    assert (opt_f.state[xf]["step"], opt_f.cc, opt_f.grad_limit_ptr) == (opt_e.state[xe]["step"], opt_e.cc, opt_e.grad_limit_ptr) == (24, 24, 1)
    assert float((xf.data - x0).abs().max()) > 1e-3                          # the loop went somewhere (24 limited steps)
    # the optimiser's moments are baked into the graph: replacing them must be refused, not silently ignored
    opt_f.reset()
    with pytest.raises(RuntimeError, match="moment tensors were replaced"):
        loop.run(24)
    with pytest.raises(RuntimeError, match="exactly the parameter"):
        FusedEnergyAdamLoop(mod_f, xf, AdamUniform([xf, torch.nn.Parameter(x0.clone())]), n_iters=2)


def test_cpp_autograd_nodes_equal_python_nodes(ext):
    """``SmoothnessBarrierEnergy`` runs its autograd node in C++ (csrc/torch_autograd.cpp) when the in-tree
    extension is there.  Same library calls as the Python Functions, so: bitwise-equal energies and gradients, eager and
    replayed, with a second loss term, a CPU grad_output, the order switch and the stale-backward check."""
    from tssplat_amd import _capi, scenes
    from tssplat_amd.energies import SmoothnessBarrierEnergy

    class Flags:
        smooth_eng_coeff = 2e-4 / 6
        barrier_coeff = 2e-4
        increase_order_iter = 1000

    assert _capi.autograd_ext() is not None, "the C++ autograd extension did not build / import on this box"
    sc = scenes.make_scene("kuhn8", 6)
    x0 = torch.from_numpy(scenes.deform(sc, 0.3)).cuda()
    w = torch.randn_like(x0)
    for graph in (False, True):
        cpp = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags, graph=graph)
        py = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags, graph=graph)
        py._ext = None                                              # force the Python Functions
        xc, xp = torch.nn.Parameter(x0.clone()), torch.nn.Parameter(x0.clone())
        for it in (0, 7, 600, 1001, 1500):
            for mod, x in ((cpp, xc), (py, xp)):
                x.grad = None
                c1, c2 = mod.coeff_scheduler(it)
                loss = (w * x).sum() * 1e-3 + 0.5 * mod(x, it, c1, c2)
                loss.backward()
                with torch.no_grad():
                    x.add_(-1e-3 * x.grad.clamp(-0.05, 0.05))       # in place: the replay keeps its storage
            e_c, e_p = float(cpp(xc, it, *cpp.coeff_scheduler(it)).detach()), float(py(xp, it, *py.coeff_scheduler(it)).detach())
            assert np.isfinite(e_c) and e_c == e_p, (graph, it, e_c, e_p)
            assert torch.equal(xc.grad, xp.grad), (graph, it)
            assert torch.equal(xc, xp)
        # grad_output on the CPU (a plain python-side constant): both routes move it to the device
        xc.grad = None
        cpp(xc, 3, 1e-4, 2e-4).backward(torch.tensor(2.0))
        g2 = xc.grad.clone()
        xc.grad = None
        cpp(xc, 3, 1e-4, 2e-4).backward()
        assert torch.allclose(g2, 2.0 * xc.grad, rtol=1e-6, atol=0)
    # stale backward of a replayed evaluation: refused by the C++ node exactly like by the Python one
    cpp = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags, graph=True)
    x = torch.nn.Parameter(x0.clone())
    e1 = cpp(x, 0, 1e-4, 2e-4)
    e2 = cpp(x, 1, 1e-4, 2e-4)
    with pytest.raises(RuntimeError, match="newer evaluation"):
        e1.backward()
    e2.backward()


def test_graph_replay_back_to_back_without_sync(ext):
    """``tsamd_graph_launch`` updates the coefficient arguments of the graph's kernel nodes
    (hipGraphExecKernelNodeSetParams) while earlier launches of the same exec may still be queued -- the real training
    path never synchronises between steps.  400 launches issued back to back on a scene whose kernels take several times
    longer than the host needs per launch (the queue is dozens of launches deep), a different (c1, c2) every launch, the order
    switch in the middle; every launch's energy and a gradient checksum are copied out ON THE STREAM and compared, after one
    final sync, with evaluations done one at a time.  A runtime that patched the arguments in place would hand early launches
    the coefficients of later ones."""
    from tssplat_amd import scenes
    from tssplat_amd.energies import SmoothnessBarrierEnergy, GraphedSmoothnessBarrier

    class Flags:
        smooth_eng_coeff = 2e-4 / 24
        barrier_coeff = 2e-4
        increase_order_iter = 1000

    sc = scenes.make_scene("kuhn19", 24)                         # ~1 M tets: ~25 us of kernels per launch
    mod = SmoothnessBarrierEnergy(sc.rest, sc.tets, Flags)
    x = torch.from_numpy(scenes.deform(sc, 0.1)).cuda()
    graphed = GraphedSmoothnessBarrier(mod, x)
    iters = [3 * i for i in range(400)]                          # 0 .. 1197: crosses it = 1000 (order 2 -> 4), coefficients change every step
    e_out = torch.zeros(len(iters), device="cuda")
    g_out = torch.zeros(len(iters), device="cuda")
    w = torch.randn_like(x)
    graphed.step(0)
    torch.cuda.synchronize()
    for k, it in enumerate(iters):                               # no synchronisation inside this loop
        e, g = graphed.step(it)
        e_out[k].copy_(e)
        g_out[k].copy_((g * w).sum())
    torch.cuda.synchronize()
    e_ref = torch.zeros_like(e_out)
    g_ref = torch.zeros_like(g_out)
    for k, it in enumerate(iters):
        e, g = graphed.step(it)
        torch.cuda.synchronize()                                 # one at a time
        e_ref[k] = e
        g_ref[k] = (g * w).sum()
    assert torch.equal(e_out, e_ref), int((e_out != e_ref).sum())
    assert torch.equal(g_out, g_ref), int((g_out != g_ref).sum())
    assert len(set(e_ref.tolist())) > 300                        # the schedule really moved the result from launch to launch


def test_sharded_module_with_its_own_exchange_on_gpu():
    """ShardedSmoothnessBarrierEnergy(graph=True, exchange="overlap") forward + backward() on the device, through a real (single
    rank) RCCL process group: the helper thread, the side stream and the lazily read job-wide value -- `bench.py --launch module
    --force-collective` asserts that the value of the last step's JobWideEnergy, read after the loop, equals the sum of the rank
    energies and that one collective per step was issued."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--scene", "kuhn8", "--spheres", "16", "--launch", "module", "--force-collective",
                        "--steps", "40", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-3000:]
    rec = json.loads(lines[-1])
    assert rec["config"]["launch"].startswith("ShardedSmoothnessBarrierEnergy(graph=True")
    assert "helper thread" in rec["config"]["energy_exchange"] and "nccl" in rec["config"]["energy_exchange"]
    assert rec["value"] > 0 and np.isfinite(rec["energy"])
    # one collective per 16 steps: same check of the last value against the rank sum
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--scene", "kuhn8", "--spheres", "16", "--launch", "module", "--force-collective",
                        "--module-every", "16", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-3000:]
    assert "every=16" in json.loads(lines[-1])["config"]["launch"]


def test_sharded_module_value_and_gradient_on_gpu(ext):
    """The same module in-process (no process group: the exchange is the identity): value and gradient against the oracle, the
    replayed and the eager local evaluator, reading the value late."""
    from tssplat_amd import scenes
    from tssplat_amd.sharding import JobWideEnergy, ShardedSmoothnessBarrierEnergy
    sc = scenes.make_scene("kuhn8", 6)

    class Flags:
        smooth_eng_coeff, barrier_coeff, increase_order_iter = 2e-4 / 6, 2e-4, 1000

    vo = sc.sphere_vertex_offsets
    to = np.arange(7) * (sc.n_tets // 6)
    O = _oracle()
    cache = O.prepare(sc.rest, sc.tets)
    for graph in (False, True):
        mod = ShardedSmoothnessBarrierEnergy(sc.rest, sc.tets, Flags, vo, to, graph=graph)
        x = torch.nn.Parameter(torch.from_numpy(scenes.deform(sc, 0.1)).cuda())
        kept = []
        for it in (0, 1, 2):
            x.grad = None
            c1, c2 = mod.coeff_scheduler(it)
            e = mod(x, it, c1, c2)
            assert isinstance(e, JobWideEnergy)
            e.backward()
            kept.append((e, c1, c2, x.grad.detach().cpu().numpy().astype(np.float64)))
        for e, c1, c2, g in kept:                              # values read after the loop: they come from the exchange's ring
            E, _, _, g64 = O.energy_and_grad(x.detach().cpu().numpy(), cache, c1, c2, 2)
            assert abs(float(e) - E) <= 1e-5 * abs(E), (graph, float(e), E)
            assert np.linalg.norm(g - g64) <= 2e-5 * np.linalg.norm(g64)
