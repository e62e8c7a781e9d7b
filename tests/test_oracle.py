"""CPU tests of the oracle itself: pinned against the reference's own `compute_G_matrix` golden
vectors, analytic known answers, finite differences, and cross-checked between its three
restatements (numpy float64, plain C float64, torch fp32 reference formulation)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import c_oracle, tet_energy_oracle as O
from tssplat_amd import scenes


@pytest.fixture(scope="module")
def small():
    sc = scenes.make_scene("kuhn4", 2)
    return sc, O.prepare(sc.rest, sc.tets)


def test_golden_G_from_reference(golden_dir):
    """tests/golden/g_matrix_golden.npz was produced by the reference's compute_G_matrix
    (geometry/mesh_utils.py:38-69); the oracle's G must reproduce it."""
    z = np.load(os.path.join(golden_dir, "g_matrix_golden.npz"))
    for key in ("aveg", "kuhn2"):
        G = O.gradient_operator_dense(z[f"{key}_rest"], z[f"{key}_tets"])
        ref = z[f"{key}_G"]
        assert G.shape == ref.shape
        assert np.abs(G - ref).max() <= 1e-11 * np.abs(ref).max()
    # and the sparse global operator agrees with the per-tet blocks: F = G x
    rest, tets = z["kuhn2_rest"], z["kuhn2_tets"]
    Gs = O.gradient_operator_sparse(rest, tets)
    assert Gs.shape == (9 * tets.shape[0], 3 * rest.shape[0]) and Gs.nnz == 36 * tets.shape[0]
    x = np.random.default_rng(0).standard_normal(rest.shape)
    _, Dminv = O.rest_operators(rest, tets)
    F = O.deformation_gradient(x.astype(np.float32), tets, Dminv)
    assert np.allclose(Gs @ x.astype(np.float32).astype(np.float64).reshape(-1), F.reshape(-1), atol=1e-10)


def test_rest_state_and_affine_maps(small):
    sc, cache = small
    m = sc.n_tets
    _, Dminv = O.rest_operators(sc.rest, sc.tets)
    F = O.deformation_gradient(sc.rest, sc.tets, Dminv)
    assert np.abs(F - np.eye(3)).max() < 1e-12                      # F(rest) = I
    E, Es, Eb, g = O.energy_and_grad(sc.rest, cache, 1.0, 1.0, 2)
    assert Es < 1e-9 and Eb == 0.0 and np.abs(g).max() < 1e-2      # fp32-rounded Dm^-1 leaves ~1e-7 noise in F
    A = np.array([[1.1, 0.2, 0.0], [-0.1, 0.9, 0.3], [0.05, 0.0, 1.2]])
    xa = sc.rest.astype(np.float64) @ A.T + np.array([0.3, -0.2, 0.1])
    Fa = O.deformation_gradient(xa, sc.tets, Dminv)
    assert np.abs(Fa - A).max() < 1e-12                             # F(Ax+b) = A for every tet
    # L annihilates constants: smoothness vanishes on affine maps, penalty = m * max(-det A, 0)^p
    R = np.diag([1.0, 1.0, -1.0])
    xr = (sc.rest.astype(np.float64) @ R.T).astype(np.float32)
    for order in (2, 4):
        E, Es, Eb, _ = O.energy_and_grad(xr, cache, 1.0, 1.0, order)
        assert Es < 1e-8 and abs(Eb - m) < 1e-4 * m
    E, Es, Eb, g = O.energy_and_grad(xr, cache, 1.0, 1.0, 3)        # order not in {2,4}: penalty off (.cu:57-63)
    assert Eb == 0.0


def test_invariances_and_additivity(small):
    sc, cache = small
    x = scenes.deform(sc, 0.2).astype(np.float64)
    E0, Es0, Eb0, g0 = O.energy_and_grad(x.astype(np.float32), cache, 1e-3, 2e-3, 2)
    th = 0.7
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    xr = (x @ Rz.T + 0.5).astype(np.float32)
    E1, Es1, Eb1, g1 = O.energy_and_grad(xr, cache, 1e-3, 2e-3, 2)
    assert abs(E1 - E0) < 1e-4 * abs(E0)                            # rigid motions leave E unchanged (F -> R F)
    assert np.linalg.norm(g1 - g0 @ Rz.T) < 1e-3 * np.linalg.norm(g0)
    # block additivity over spheres: the sharding invariant
    nv, nt = sc.n_vertices // 2, sc.n_tets // 2
    ca = O.prepare(sc.rest[:nv], sc.tets[:nt])
    cb = O.prepare(sc.rest[nv:], sc.tets[nt:] - nv)
    Ea = O.energy_and_grad(x[:nv].astype(np.float32), ca, 1e-3, 2e-3, 2)
    Eb_ = O.energy_and_grad(x[nv:].astype(np.float32), cb, 1e-3, 2e-3, 2)
    assert abs(Ea[0] + Eb_[0] - E0) < 1e-12 * abs(E0)
    assert np.abs(np.concatenate([Ea[3], Eb_[3]]) - g0).max() < 1e-12 * np.abs(g0).max()


@pytest.mark.parametrize("order", [2, 4])
def test_finite_differences(small, order):
    sc, cache = small
    x = scenes.deform(sc, 0.3)
    _, _, _, g = O.energy_and_grad(x, cache, 1e-3, 2e-3, order)
    idx = np.random.default_rng(1).integers(0, g.size, 12)
    fd = O.finite_difference_grad(x, cache, 1e-3, 2e-3, order, idx, h=1e-6)
    assert np.abs(fd - g.reshape(-1)[idx]).max() <= 1e-6 * max(1.0, np.abs(fd).max())


def test_matrix_form_equals_factored_form(small):
    """x^T (G^T L^T L G) x / 2 == 1/2 |L G x|^2 and M x == G^T L^T L G x (float64)."""
    sc, cache = small
    M, G = O.biharmonic_matrix(sc.rest, sc.tets)
    assert (abs(M - M.T) > 1e-9 * abs(M).max()).nnz == 0
    x = scenes.deform(sc, 0.1)
    xf = x.astype(np.float64).reshape(-1)
    # unrounded operators on both sides
    c_exact = O.prepare(sc.rest, sc.tets, round_fp32=False)
    _, Es, _, g = O.energy_and_grad(x, c_exact, 1.0, 0.0, 2)
    assert abs(0.5 * xf @ (M @ xf) - Es) <= 1e-9 * Es
    assert np.abs((M @ xf).reshape(-1, 3) - g).max() <= 1e-8 * np.abs(g).max()


def test_c_oracle_matches_numpy_oracle():
    c_oracle.build()
    for kind, S, sigma, order in [("kuhn4", 3, 0.3, 2), ("kuhn4", 3, 0.3, 4), ("cone", 2, 0.1, 2), ("kuhn3", 1, 0.0, 2)]:
        sc = scenes.make_scene(kind, S)
        cache = O.prepare(sc.rest, sc.tets)
        assert np.array_equal(c_oracle.face_adjacency(sc.tets), cache.nbr)
        x = scenes.deform(sc, sigma)
        E, Es, Eb, g = O.energy_and_grad(x, cache, 3e-4, 2e-4, order, grad_output=0.7)
        Ec, Esc, Ebc, gc = c_oracle.energy_and_grad(sc.rest, sc.tets, x, 3e-4, 2e-4, order, grad_output=0.7)
        assert abs(E - Ec) <= 1e-12 * max(abs(E), 1e-30) + 1e-20
        assert abs(Es - Esc) <= 1e-12 * max(Es, 1e-30) + 1e-20 and abs(Eb - Ebc) <= 1e-12 * max(Eb, 1e-30)
        assert np.abs(g - gc).max() <= 1e-11 * np.abs(g).max() + 1e-15   # at rest g is pure cancellation noise


def test_torch_reference_formulation_within_its_band(aveg):
    """The "vanilla PyTorch" fp32 M-formulation (oracle/torch_energies.py) agrees with the float64
    oracle inside SURVEY.md 8(c)'s band -- and needs that band: near rest it returns O(1) noise."""
    torch = pytest.importorskip("torch")
    from oracle import torch_energies as TE
    rest, tets = aveg
    rest, tets = rest[:], tets[:]
    # a 3000-tet sub-mesh keeps the sparse products fast
    keep = np.arange(3000)
    used, inv = np.unique(tets[keep], return_inverse=True)
    rest_s, tets_s = rest[used], inv.reshape(-1, 4).astype(np.int32)
    ts = TE.TorchTetSpheres(rest_s, tets_s)
    cache = O.prepare(rest_s, tets_s)
    eps = 2.0 ** -23
    rng = np.random.default_rng(5)
    for sigma in (0.0, 1e-2, 5e-2):
        x = (rest_s + sigma * rng.standard_normal(rest_s.shape)).astype(np.float32)
        c1, c2 = 2e-4, 2e-4
        E, Es, Eb, g = O.energy_and_grad(x, cache, c1, c2, 2)
        A, nMx = O.tolerance_scales(x, cache)
        e_t = float(TE.compute_energy(torch.from_numpy(x), ts, c1, c2, 2))
        g_t = TE.compute_energy_backward(1.0, torch.from_numpy(x), ts, c1, c2, 2).numpy().astype(np.float64)
        assert abs(e_t - E) <= 1e-5 * abs(E) + 8 * eps * c1 * A + 1e-5 * c2 * Eb
        assert np.linalg.norm(g_t - g) <= 1e-5 * np.linalg.norm(g) + 8 * eps * c1 * nMx


def test_non_manifold_is_rejected():
    tets = np.array([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 5]], dtype=np.int32)   # one face, three tets
    with pytest.raises(ValueError):
        O.face_adjacency(tets)
    with pytest.raises(ValueError):
        c_oracle.face_adjacency(tets)


@settings(max_examples=25, deadline=None)
@given(k=st.integers(1, 3), seed=st.integers(0, 10_000), sigma=st.floats(0.0, 0.5), order=st.sampled_from([2, 4]))
def test_property_gradient_is_derivative(k, seed, sigma, order):
    """Random small complexes: directional derivative of E matches g . dx."""
    v, t = scenes.kuhn_ball(k)
    rng = np.random.default_rng(seed)
    rest = (v * rng.uniform(0.2, 2.0) + rng.uniform(-1, 1, 3)).astype(np.float32)
    cache = O.prepare(rest, t)
    x = (rest + sigma * 0.3 * rng.standard_normal(rest.shape)).astype(np.float32)
    dx = rng.standard_normal(rest.shape)
    dx /= np.linalg.norm(dx)
    _, _, _, g = O.energy_and_grad(x, cache, 1e-2, 1e-2, order)
    idx = np.arange(x.size)
    # central difference along dx using the float64 energy
    h = 1e-6
    xp = x.astype(np.float64) + h * dx
    xm = x.astype(np.float64) - h * dx

    def E64(xx):
        T = cache.tets
        p = xx[T]
        Ds = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)
        F = Ds @ cache.Dminv
        H = cache.L @ F.reshape(-1, 9)
        pen, _ = O._penalty(O.det3(F), order)
        return float(np.float32(1e-2)) * 0.5 * np.sum(H * H) + float(np.float32(1e-2)) * np.sum(pen)

    dd = (E64(xp) - E64(xm)) / (2 * h)
    assert abs(dd - float(np.sum(g * dx))) <= 1e-5 * max(1.0, abs(dd), np.linalg.norm(g))
    assert idx.size == x.size


def test_pin_tool_identifies_the_operator():
    """tools/pin_L_with_pypgo.py is the one-command route from "L unpinned" to "pinned" on a machine with libpgo.
    Its comparison logic is exercised here on stand-in "libpgo" matrices built by the oracle (transposed vec(F)
    row order included): each candidate operator must be recognised as itself."""
    import importlib.util
    import scipy.sparse as sp
    from tssplat_amd import scenes
    spec = importlib.util.spec_from_file_location("pin_tool", os.path.join(os.path.dirname(__file__), "..", "tools", "pin_L_with_pypgo.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    v, t = scenes.kuhn_ball(3)
    rest = v.astype(np.float32)
    n, m = rest.shape[0], t.shape[0]
    G = O.gradient_operator_sparse(rest, t, n)
    perm = np.arange(9).reshape(3, 3).T.ravel()
    Pm = sp.kron(sp.identity(m), sp.csr_matrix((np.ones(9), (np.arange(9), perm)), shape=(9, 9)), format="csr")
    nbr = O.face_adjacency(t)
    for name, L in (("uniform", O.element_laplacian(nbr)), ("scaled", O.element_laplacian_scaled(nbr)),
                    ("vertex-neighbours", tool.vertex_neighbour_laplacian(t))):
        LG = sp.kron(L, sp.identity(9), format="csr") @ G
        winner, err, dG = tool.compare((LG.T @ LG).tocsr(), (Pm @ G).tocsr(), rest, t, verbose=False)
        assert winner == name and err <= 1e-12 and dG <= 1e-12


def test_libpgo_pin_if_present(golden_dir):
    """Once tools/pin_L_with_pypgo.py has been run on a machine with libpgo, its output pins L: the oracle's
    M = G^T L^T L G for the recorded operator must reproduce libpgo's GTLTLG on the reference's a.veg."""
    import scipy.sparse as sp
    path = os.path.join(golden_dir, "libpgo_operator_pin.npz")
    if not os.path.exists(path):
        pytest.skip("L is PARITY UNPINNED: libpgo has not been available to any builder yet (tools/pin_L_with_pypgo.py)")
    z = np.load(path)
    mesh = np.load(os.path.join(golden_dir, "aveg_mesh.npz"))
    rest, tets = mesh["rest"], mesh["tets"]
    n, m = int(z["n"]), int(z["m"])
    assert (n, m) == (rest.shape[0], tets.shape[0])
    M_pgo = sp.coo_matrix((z["M_val"], (z["M_row"], z["M_col"])), shape=(3 * n, 3 * n)).tocsr()
    assert str(z["winner"]) == "uniform", "libpgo's operator is not the one this library assumes by default"
    M, _ = O.biharmonic_matrix(rest, tets, n)
    assert abs(M - M_pgo).max() <= 1e-9 * abs(M_pgo).max()


@pytest.mark.parametrize("kind,S,sigma,order", [("kuhn8", 4, 0.3, 2), ("kuhn8", 3, 0.02, 4), ("kuhn6", 2, 0.0, 2),
                                               ("delaunay700", 2, 0.3, 4), ("cone", 1, 0.3, 2)])
def test_c_rounding_model_equals_numpy_model(kind, S, sigma, order):
    """oracle/c tso_rounding_model is the yardstick of the full-size GPU tests: it must be the numpy model
    (variances are stored in fp32 there: 1e-7 relative)."""
    from oracle import c_oracle
    from tssplat_amd import scenes
    sc = scenes.make_scene(kind, S)
    x = scenes.deform(sc, sigma)
    c1, c2 = 2e-4 / S, 2e-4
    cache = O.prepare(sc.rest, sc.tets)
    se, sg = O.rounding_error_model(x, cache, c1, c2, order)
    se2, sg2 = c_oracle.rounding_error_model(sc.rest, sc.tets, x, c1, c2, order)
    assert abs(se - se2) <= 1e-6 * se
    assert np.all(np.abs(sg - sg2) <= 1e-6 * sg + 1e-300)


# --------------------------------------------------------------------------- #
# oracle/_ref: the reference's own det / ddetA_dA / penalty kernels (tet_spheres_cuda.cu:9-102), compiled for the host
# --------------------------------------------------------------------------- #

def _check_against_reference_kernels(F, det32, det64, cof32, cof64, fwd, bwd):
    """The oracle's restatement of .cu:9-102 against outputs of the reference code itself.  The reference stores a
    3x3 matrix column-major (`elt`, .cu:9-19), the oracle row-major: det is transpose-invariant and cof(A^T) = cof(A)^T,
    so the same nine numbers go in and the same nine come out in either convention."""
    F64 = F.astype(np.float64).reshape(-1, 3, 3)
    J = O.det3(F64)
    C = O.cofactor3(F64).reshape(-1, 9)
    scale_j = np.abs(F64).max(axis=(1, 2)) ** 3 + 1e-300
    scale_c = np.abs(F64).max(axis=(1, 2))[:, None] ** 2 + 1e-300
    assert np.all(np.abs(J - det64) <= 4e-16 * scale_j)             # same six products, double: rounding of the sums only
    assert np.all(np.abs(C - cof64) <= 4e-16 * scale_c)
    assert np.all(np.abs(J - det32) <= 4e-7 * scale_j)              # the fp32 instantiation the kernels use
    assert np.all(np.abs(C - cof32) <= 4e-7 * scale_c)
    for order in (2, 3, 4):
        pen, dpen = O._penalty(J, order)
        ref_pen, ref_dP = fwd[order].astype(np.float64), bwd[order].astype(np.float64)
        if order == 3:                                              # neither 2 nor 4: the reference yields 0 (.cu:57-63, :86-91)
            assert np.all(ref_pen == 0) and np.all(ref_dP == 0) and np.all(pen == 0) and np.all(dpen == 0)
            continue
        # fp32 kernels against the float64 restatement: first-order propagation of the fp32 error of det (4e-7 scale_j,
        # asserted above) and of the cofactor entries (4e-7 scale_c) through J^order and pen'(J) cof F
        Jm = np.maximum(-J, 0.0) + 4e-7 * scale_j
        dJ, dC = 4e-7 * scale_j, 4e-7 * scale_c
        assert np.all(np.abs(pen - ref_pen) <= 2 * order * Jm ** (order - 1) * dJ + 1e-6 * pen)
        dP = dpen[:, None] * C                                      # d pen / dF = pen'(J) cof F   (.cu:93-95)
        tol_dP = 2 * (order * (order - 1) * Jm ** (order - 2) * dJ)[:, None] * (np.abs(C) + dC) + 2 * (order * Jm ** (order - 1))[:, None] * dC \
            + 1e-6 * np.abs(dP)
        assert np.all(np.abs(dP - ref_dP) <= tol_dP)
        assert np.all((ref_dP != 0).any(axis=1) <= (det32 < 0))     # zero unless inverted (.cu:97-101)


def test_det_cofactor_penalty_against_reference_golden():
    """Committed outputs of the reference's own kernels (tests/golden/make_ref_det_golden.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_det_golden.npz"))
    assert int((g["det_f64"] < 0).sum()) > 50                       # the penalty branch is exercised
    _check_against_reference_kernels(g["F"], g["det_f32"], g["det_f64"], g["cof_f32"], g["cof_f64"],
                                     {k: g[f"fwd{k}"] for k in (2, 3, 4)}, {k: g[f"bwd{k}"] for k in (2, 3, 4)})


def test_det_cofactor_penalty_against_oracle_ref_live():
    """oracle/_ref/libref_det.so itself (built from /root/reference where it lies; travels to the GPU box), on fresh
    random matrices -- and it reproduces the committed golden bit for bit."""
    from oracle import ref_det
    if not ref_det.available():
        pytest.skip("oracle/_ref is not built and /root/reference is not present")
    rng = np.random.default_rng(99)
    F = (np.eye(3).reshape(1, 9) + 0.8 * rng.standard_normal((500, 9))).astype(np.float32)
    _check_against_reference_kernels(F, ref_det.det(F), ref_det.det(F.astype(np.float64)), ref_det.ddetA_dA(F),
                                     ref_det.ddetA_dA(F.astype(np.float64)),
                                     {k: ref_det.forward_det(F, k) for k in (2, 3, 4)},
                                     {k: ref_det.backward_det(F, k) for k in (2, 3, 4)})
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_det_golden.npz"))
    assert np.array_equal(ref_det.det(g["F"]), g["det_f32"]) and np.array_equal(ref_det.backward_det(g["F"], 4), g["bwd4"])
