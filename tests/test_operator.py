"""The element operator L as DATA (SURVEY.md 8(b) optional create input).

The reference takes L from libpgo (`pgo_create_tet_biharmonic_gradient_matrix(geo, 1, 0)`,
/root/reference/tssplat_ext/tet_spheres/tet_spheres.cpp:148), whose source is not in the reference:
the default of this library is the ASSUMED uniform face-adjacency umbrella.  These tests cover the
path that substitutes any other operator of the same sparsity: host-side plan replay in float64 here,
HIP parity in tests/test_gpu_parity.py::test_explicit_operator_*.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tet_energy_oracle as O
from tssplat_amd import scenes
import tile_emulator as TE


@pytest.fixture(scope="module")
def ext():
    from tssplat_amd import tet_spheres_ext
    return tet_spheres_ext


def random_operator(nbr, rng, symmetric):
    """Random weights on the face-adjacency pattern (+ diagonal); optionally symmetric."""
    m = nbr.shape[0]
    rows = np.repeat(np.arange(m), 4)
    cols = nbr.ravel()
    ok = cols >= 0
    A = sp.csr_matrix((rng.uniform(-2.0, -0.2, ok.sum()), (rows[ok], cols[ok])), shape=(m, m))
    if symmetric:
        A = 0.5 * (A + A.T)
    return (sp.diags(rng.uniform(1.0, 5.0, m)) + A).tocsr()


def _replay(ext, sc, L, kw, cases=((0.02, 2, 1.0), (0.3, 4, 0.5)), symmetric=False):
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True, operator=L, **kw)
    assert ts.plan_info()["n_planes"] == (18 if symmetric else 22)      # a symmetric operator stores its weights once
    cache = O.prepare(sc.rest, sc.tets, L=L)
    for sigma, order, go in cases:
        x = scenes.deform(sc, sigma)
        E, Es, Eb, g = O.energy_and_grad(x, cache, 5e-5, 2e-4, order, grad_output=go)
        E2, Es2, Eb2, g2 = TE.emulate(ts, x, 5e-5, 2e-4, order, grad_output=go)
        assert abs(Es - Es2) <= 1e-12 * Es and abs(Eb - Eb2) <= 1e-12 * max(Eb, 1e-300)
        assert np.abs(g - g2).max() <= 1e-11 * np.abs(g).max()
    return ts


@pytest.mark.parametrize("kind,S,kw", [
    ("kuhn8", 2, {}),
    ("kuhn12", 1, {}),                                              # bisected: halo slots carry column weights
    ("kuhn12", 1, dict(max_threads=640, lds_budget_bytes=100000)),
    ("kuhn12", 1, dict(debug_flags=2)),                           # no conflict-aware re-ordering of the neighbours
    ("delaunay700", 2, dict(lds_budget_bytes=40000)),
])
def test_explicit_operator_replays_to_oracle(ext, kind, S, kw):
    sc = scenes.make_scene(kind, S)
    nbr = O.face_adjacency(sc.tets)
    rng = np.random.default_rng(5)
    _replay(ext, sc, O.element_laplacian_scaled(nbr), kw)           # row-scaled umbrella: non-symmetric
    _replay(ext, sc, random_operator(nbr, rng, symmetric=False), kw)
    _replay(ext, sc, random_operator(nbr, rng, symmetric=True), kw, symmetric=True)


def test_explicit_uniform_operator_equals_default(ext):
    """The assumed operator passed as data gives the same result as the built-in path, and the 13 common
    planes of the two plans are identical."""
    sc = scenes.make_scene("kuhn12", 1)
    L = O.element_laplacian(O.face_adjacency(sc.tets))
    # (explicit-operator plans use the same tiling as the built-in operator: their extra planes live in registers)
    ts_d = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True)
    ts_x = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True, operator=L)
    assert ts_d.plan_info()["n_planes"] == 13 and ts_x.plan_info()["n_planes"] == 18    # (the umbrella is symmetric)
    assert ts_x.plan_info()["block_threads"] == ts_d.plan_info()["block_threads"] and ts_x.plan_info()["lds_bytes"] == ts_d.plan_info()["lds_bytes"]
    for a, b in zip(TE.plan_tiles(ts_d), TE.plan_tiles(ts_x)):
        assert np.array_equal(a["planes"], b["planes"][:13]) and np.array_equal(a["row_start"], b["row_start"]) and np.array_equal(a["vdst"], b["vdst"])
    x = scenes.deform(sc, 0.1)
    r_d = TE.emulate(ts_d, x, 5e-5, 2e-4, 2)
    r_x = TE.emulate(ts_x, x, 5e-5, 2e-4, 2)
    # (not bitwise: the built-in path evaluates 4 * own - sum with a missing face reading own, the explicit one deg * own - sum)
    assert abs(r_d[0] - r_x[0]) <= 1e-13 * abs(r_d[0]) and np.abs(r_d[3] - r_x[3]).max() <= 1e-12 * np.abs(r_d[3]).max()


def test_operator_validation(ext):
    sc = scenes.make_scene("kuhn3", 1)
    v, t = sc.rest.reshape(-1), sc.tets.reshape(-1)
    m = sc.n_tets
    nbr = O.face_adjacency(sc.tets)
    L = O.element_laplacian(nbr).tolil()
    # an entry between tets that share no face
    far = next(j for j in range(m) if j != 0 and j not in nbr[0])
    bad = L.copy()
    bad[0, far] = 0.25
    with pytest.raises(RuntimeError, match="neither on the diagonal nor a face adjacency"):
        ext.TetSpheres(v, t, host_only=True, operator=bad.tocsr())
    # explicit zeros outside the pattern are tolerated, duplicates are summed
    csr = L.tocsr()
    rp = np.concatenate([csr.indptr, [csr.indptr[-1] + 2]]).astype(np.int64)[:-1].copy()
    ci = np.concatenate([csr.indices, []]).astype(np.int32)
    va = csr.data.copy()
    # split the diagonal of the last row in two halves and add a harmless zero
    last = slice(csr.indptr[-2], csr.indptr[-1])
    cols_last, vals_last = ci[last].copy(), va[last].copy()
    d = np.where(cols_last == m - 1)[0][0]
    ci2 = np.concatenate([ci, [m - 1, far if far != m - 1 else 0]]).astype(np.int32)
    va2 = np.concatenate([va, [vals_last[d] / 2, 0.0]])
    va2[csr.indptr[-2] + d] = vals_last[d] / 2
    rp2 = csr.indptr.astype(np.int64).copy()
    rp2[-1] += 2
    ts = ext.TetSpheres(v, t, host_only=True, operator=(rp2, ci2, va2))
    x = scenes.deform(sc, 0.1)
    cache = O.prepare(sc.rest, sc.tets)
    E, Es, Eb, g = O.energy_and_grad(x, cache, 5e-5, 2e-4, 2)
    E2, Es2, Eb2, g2 = TE.emulate(ts, x, 5e-5, 2e-4, 2)
    assert abs(Es - Es2) <= 1e-12 * Es and np.abs(g - g2).max() <= 1e-11 * np.abs(g).max()
    # wrong shapes
    with pytest.raises(ValueError):
        ext.TetSpheres(v, t, host_only=True, operator=sp.identity(m + 1, format="csr"))
    with pytest.raises(TypeError):
        ext.TetSpheres("x.veg", host_only=True, operator=csr)


def test_operator_csr_is_validated(ext):
    """a CSR that starts anywhere but 0, non-finite weights and operator + rebuild_dminv are errors, not
    silently accepted inputs."""
    sc = scenes.make_scene("kuhn3", 1)
    v, t = sc.rest.reshape(-1), sc.tets.reshape(-1)
    csr = O.element_laplacian(O.face_adjacency(sc.tets)).tocsr()
    rp, ci, va = csr.indptr.astype(np.int64), csr.indices.astype(np.int32), csr.data.astype(np.float64)
    bad_rp = rp.copy()
    bad_rp[0] = -3
    with pytest.raises(RuntimeError, match=r"rowptr\[0\] must be 0"):
        ext.TetSpheres(v, t, host_only=True, operator=(bad_rp, ci, va))
    for poison in (np.nan, np.inf):
        bad_va = va.copy()
        bad_va[5] = poison
        with pytest.raises(RuntimeError, match="non-finite value"):
            ext.TetSpheres(v, t, host_only=True, operator=(rp, ci, bad_va))
    with pytest.raises(RuntimeError, match="not combined with rebuild_dminv"):
        ext.TetSpheres(v, t, host_only=True, operator=csr, rebuild_dminv=True)
    # the tile kernels are compiled for at most 768 threads and two slots per lane
    with pytest.raises(RuntimeError, match="max_threads exceeds"):
        ext.TetSpheres(v, t, host_only=True, max_threads=1024)
    with pytest.raises(RuntimeError, match="slots_per_thread must be 0, 2, 3 or 4"):
        ext.TetSpheres(v, t, host_only=True, slots_per_thread=5)
    with pytest.raises(RuntimeError, match="built-in operator"):
        ext.TetSpheres(v, t, host_only=True, slots_per_thread=4, operator=csr)
