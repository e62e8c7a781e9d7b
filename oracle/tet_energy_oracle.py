"""CPU float64 ORACLE for the tet-sphere geometry energy.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path (``tssplat_amd``) never
does: it runs hand-written HIP kernels or fails loudly.

What it restates (all citations into /root/reference):

* energy composition ``E = c1 * 0.5 * x^T (G^T L^T L G) x + c2 * sum_e max(-det F_e, 0)^p``
  -- tssplat_ext/tet_spheres/tet_spheres_cuda.cu:129-157 (smoothness, the ``*0.5``
  at :157), :167-185 (penalty), :191 (``sm*c1 + bar*c2``);
* gradient ``g = gradH * (c1 * M x + c2 * G^T dEb/dF)`` -- .cu:214-258;
* 3x3 determinant / cofactor -- .cu:21-46 (`det`, `ddetA_dA`);
* penalty forward/backward per tet -- .cu:48-66 and :68-102 (order 2 or 4,
  any other order contributes 0);
* the gradient operator ``G`` (F = Ds * Dm^-1, row-major vec, dof 3*a+i)
  -- geometry/mesh_utils.py:38-69 (`compute_G_matrix`), which libpgo's
  `pgo_create_tet_gradient_matrix` (tet_spheres.cpp:149) also builds;
* fp32 -> double promotion of the rest positions before the operators are
  built, double -> fp32 rounding of the operator values afterwards
  -- tet_spheres.cpp:252-255 and :43-45.

PARITY PINNING.  ``G`` is pinned: `tests/golden/g_matrix_golden.npz` was
produced by importing the reference's own `compute_G_matrix` (script
`tests/golden/make_golden.py`), and this module reproduces it to 1e-12.
``L`` -- libpgo's `pgo_create_tet_biharmonic_gradient_matrix(geo, faceNeighbor=1,
scale=0)` (tet_spheres.cpp:148) -- is **PARITY UNPINNED**: libpgo
(github.com/bohanwang/libpgo, consumed through `pgo_c.h`, no version pinned by
the reference) is not vendored, and the reference holds no test, golden value
or source for it.  We restate it as the uniform face-adjacency graph Laplacian
over tets, applied to each of the 9 components of F
(``(L F)_e = deg(e) F_e - sum_{e' ~ e} F_e'``), which is what "faceNeighbor=1,
scale=0" denotes; adjacency and weights are plain data so the true operator
can be substituted.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = [
    "rest_operators",
    "face_adjacency",
    "element_laplacian",
    "element_laplacian_scaled",
    "deformation_gradient",
    "det3",
    "cofactor3",
    "gradient_operator_dense",
    "gradient_operator_sparse",
    "biharmonic_matrix",
    "energy",
    "energy_and_grad",
    "finite_difference_grad",
    "tolerance_scales",
    "factored_tolerances",
    "prepare",
]


# --------------------------------------------------------------------------- #
# rest-state operators
# --------------------------------------------------------------------------- #

def rest_operators(rest: np.ndarray, tets: np.ndarray, round_fp32: bool = False):
    """``Dm`` and ``Dm^-1`` per tet in float64 from (fp32-valued) rest positions.

    Follows tet_spheres.cpp:252-255: positions are float32 promoted to double
    before the operators are built.  ``round_fp32=True`` additionally rounds
    ``Dm^-1`` to float32 (tet_spheres.cpp:43-45) and returns it as float64.
    """
    X = np.asarray(rest, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    T = np.asarray(tets).reshape(-1, 4).astype(np.int64)
    p = X[T]                                     # [m,4,3]
    Dm = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)  # columns
    Dminv = np.linalg.inv(Dm)
    if round_fp32:
        Dminv = Dminv.astype(np.float32).astype(np.float64)
    return Dm, Dminv


def face_adjacency(tets: np.ndarray) -> np.ndarray:
    """``nbr[e, k]`` = tet sharing the face of ``e`` opposite local vertex ``k``, else -1.

    Raises ``ValueError`` if a face is shared by more than two tets.
    """
    T = np.asarray(tets).reshape(-1, 4).astype(np.int64)
    m = T.shape[0]
    opp = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
    faces = np.sort(T[:, opp], axis=2).reshape(-1, 3)      # [4m,3], row = 4*e + k
    order = np.lexsort((faces[:, 2], faces[:, 1], faces[:, 0]))
    fs = faces[order]
    same = np.all(fs[1:] == fs[:-1], axis=1)
    if same.size > 1 and np.any(same[1:] & same[:-1]):
        raise ValueError("non-manifold tet mesh: a face is shared by more than two tets")
    nbr = np.full(4 * m, -1, dtype=np.int64)
    a = order[:-1][same]
    b = order[1:][same]
    nbr[a] = b // 4
    nbr[b] = a // 4
    return nbr.reshape(m, 4)


def element_laplacian(nbr: np.ndarray) -> sp.csr_matrix:
    """Uniform face-adjacency graph Laplacian over tets, ``L = D - A`` (m x m)."""
    m = nbr.shape[0]
    rows = np.repeat(np.arange(m), 4)
    cols = nbr.ravel()
    ok = cols >= 0
    A = sp.csr_matrix((np.ones(ok.sum()), (rows[ok], cols[ok])), shape=(m, m))
    deg = np.asarray(A.sum(axis=1)).ravel()
    return (sp.diags(deg) - A).tocsr()


# --------------------------------------------------------------------------- #
# per-tet 3x3 algebra (tet_spheres_cuda.cu:21-46)
# --------------------------------------------------------------------------- #

def deformation_gradient(x: np.ndarray, tets: np.ndarray, Dminv: np.ndarray) -> np.ndarray:
    """``F_e = Ds_e * Dm_e^-1`` with ``Ds = [x1-x0, x2-x0, x3-x0]`` as columns."""
    p = np.asarray(x, dtype=np.float64).reshape(-1, 3)[np.asarray(tets).reshape(-1, 4)]
    Ds = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)
    return Ds @ Dminv


def det3(F: np.ndarray) -> np.ndarray:
    """Six-term expansion, same grouping as `det` at .cu:21-30."""
    return (-F[:, 0, 2] * F[:, 1, 1] * F[:, 2, 0] + F[:, 0, 1] * F[:, 1, 2] * F[:, 2, 0]
            + F[:, 0, 2] * F[:, 1, 0] * F[:, 2, 1] - F[:, 0, 0] * F[:, 1, 2] * F[:, 2, 1]
            - F[:, 0, 1] * F[:, 1, 0] * F[:, 2, 2] + F[:, 0, 0] * F[:, 1, 1] * F[:, 2, 2])


def cofactor3(F: np.ndarray) -> np.ndarray:
    """``d det(F) / dF`` (the cofactor matrix), `ddetA_dA` at .cu:32-46."""
    C = np.empty_like(F)
    C[:, 0, 0] = F[:, 1, 1] * F[:, 2, 2] - F[:, 1, 2] * F[:, 2, 1]
    C[:, 0, 1] = F[:, 1, 2] * F[:, 2, 0] - F[:, 1, 0] * F[:, 2, 2]
    C[:, 0, 2] = F[:, 1, 0] * F[:, 2, 1] - F[:, 1, 1] * F[:, 2, 0]
    C[:, 1, 0] = F[:, 0, 2] * F[:, 2, 1] - F[:, 0, 1] * F[:, 2, 2]
    C[:, 1, 1] = F[:, 0, 0] * F[:, 2, 2] - F[:, 0, 2] * F[:, 2, 0]
    C[:, 1, 2] = F[:, 0, 1] * F[:, 2, 0] - F[:, 0, 0] * F[:, 2, 1]
    C[:, 2, 0] = F[:, 0, 1] * F[:, 1, 2] - F[:, 0, 2] * F[:, 1, 1]
    C[:, 2, 1] = F[:, 0, 2] * F[:, 1, 0] - F[:, 0, 0] * F[:, 1, 2]
    C[:, 2, 2] = F[:, 0, 0] * F[:, 1, 1] - F[:, 0, 1] * F[:, 1, 0]
    return C


def _cofactor_abs(Fa: np.ndarray) -> np.ndarray:
    """Entrywise upper bound of |cofactor| for |F| = Fa (all products added)."""
    C = np.empty_like(Fa)
    for i in range(3):
        for j in range(3):
            r = [a for a in range(3) if a != i]
            c = [b for b in range(3) if b != j]
            C[:, i, j] = Fa[:, r[0], c[0]] * Fa[:, r[1], c[1]] + Fa[:, r[0], c[1]] * Fa[:, r[1], c[0]]
    return C


def _penalty(J: np.ndarray, order: int):
    """Per-tet penalty value and d(penalty)/dJ for J = det F (.cu:48-102)."""
    Jm = np.maximum(-J, 0.0)
    if order == 2:
        return Jm * Jm, np.where(J < 0, -2.0 * Jm, 0.0)
    if order == 4:
        return Jm ** 4, np.where(J < 0, -4.0 * Jm ** 3, 0.0)
    return np.zeros_like(J), np.zeros_like(J)


# --------------------------------------------------------------------------- #
# explicit matrices (the reference formulation)
# --------------------------------------------------------------------------- #

def gradient_operator_dense(rest: np.ndarray, tets: np.ndarray) -> np.ndarray:
    """Per-tet ``G_e`` as float64 ``[m, 9, 12]``; row ``3i+j`` <-> ``F[i,j]``, column
    ``3a+i`` <-> coordinate ``i`` of local vertex ``a`` (mesh_utils.py:38-69)."""
    _, Dminv = rest_operators(rest, tets)
    m = Dminv.shape[0]
    g = np.empty((m, 4, 3))
    g[:, 1:, :] = Dminv                       # g_{k+1}[j] = Dminv[k, j]
    g[:, 0, :] = -Dminv.sum(axis=1)
    G = np.zeros((m, 9, 12))
    for a in range(4):
        for i in range(3):
            for j in range(3):
                G[:, 3 * i + j, 3 * a + i] = g[:, a, j]
    return G


def gradient_operator_sparse(rest, tets, n_vertices: int | None = None) -> sp.csr_matrix:
    """Global ``G`` (9m x 3n), 36 nonzeros per tet (tet_spheres.cpp:149, :156)."""
    T = np.asarray(tets).reshape(-1, 4).astype(np.int64)
    m = T.shape[0]
    n = int(n_vertices if n_vertices is not None else np.asarray(rest).reshape(-1, 3).shape[0])
    Ge = gradient_operator_dense(rest, tets)                   # [m,9,12]
    rows = (9 * np.arange(m)[:, None, None] + np.arange(9)[None, :, None]) + np.zeros((1, 1, 12), dtype=np.int64)
    cols = (3 * T[:, :, None] + np.arange(3)[None, None, :]).reshape(m, 1, 12) + np.zeros((1, 9, 1), dtype=np.int64)
    nz = Ge != 0
    # keep the structural 36 per tet even where a value happens to be 0
    struct = np.zeros((9, 12), dtype=bool)
    for a in range(4):
        for i in range(3):
            for j in range(3):
                struct[3 * i + j, 3 * a + i] = True
    keep = np.broadcast_to(struct, Ge.shape) | nz
    return sp.csr_matrix((Ge[keep], (rows[keep], cols[keep])), shape=(9 * m, 3 * n))


def biharmonic_matrix(rest, tets, n_vertices: int | None = None):
    """``(M, G)`` with ``M = G^T (L (x) I9)^T (L (x) I9) G`` (tet_spheres.cpp:148-149)."""
    G = gradient_operator_sparse(rest, tets, n_vertices)
    L = element_laplacian(face_adjacency(tets))
    L9 = sp.kron(L, sp.identity(9), format="csr")
    LG = (L9 @ G).tocsr()
    M = (LG.T @ LG).tocsr()
    return M, G


# --------------------------------------------------------------------------- #
# energy and analytic gradient (factored form, float64)
# --------------------------------------------------------------------------- #

class _Cache:
    """Rest-state data for one mesh so repeated evaluations are cheap."""

    def __init__(self, rest, tets, round_fp32=True, nbr=None, L=None):
        self.tets = np.asarray(tets).reshape(-1, 4).astype(np.int64)
        self.n = int(np.asarray(rest).reshape(-1, 3).shape[0])
        _, self.Dminv = rest_operators(rest, self.tets, round_fp32=round_fp32)
        self.nbr = face_adjacency(self.tets) if nbr is None else np.asarray(nbr)
        if L is None:
            self.L = element_laplacian(self.nbr)
        else:
            # an explicit element operator (what tsamd_create_with_operator takes); its values are rounded
            # to fp32 like every operator value of the reference (tet_spheres.cpp:43-45)
            L = sp.csr_matrix(L, dtype=np.float64).copy()     # own arrays: scipy sorts indices in place later on
            L.sum_duplicates()
            if round_fp32:
                L.data = L.data.astype(np.float32).astype(np.float64)
            self.L = L


def prepare(rest, tets, round_fp32: bool = True, nbr=None, L=None) -> _Cache:
    """Build the rest-state cache.  ``round_fp32=True`` mirrors the reference's
    double->fp32 rounding of the operator values (tet_spheres.cpp:43-45).  ``L`` replaces the assumed
    uniform element Laplacian by an explicit (m x m, possibly non-symmetric) operator."""
    return _Cache(rest, tets, round_fp32=round_fp32, nbr=nbr, L=L)


def element_laplacian_scaled(nbr: np.ndarray) -> sp.csr_matrix:
    """The row-scaled variant ``D^-1 (D - A)`` (diagonal 1, off-diagonals ``-1/deg``): what a ``scale=1``
    argument of libpgo's `pgo_create_tet_biharmonic_gradient_matrix` would plausibly denote.  Non-symmetric;
    used by the tests of the explicit-operator path."""
    L = element_laplacian(nbr)
    d = L.diagonal()
    d[d == 0] = 1.0
    return sp.diags(1.0 / d) @ L


def energy_and_grad(x, cache: _Cache, c1: float, c2: float, order: int,
                    grad_output: float = 1.0, want_grad: bool = True):
    """Return ``(E, E_s, E_b, grad)`` in float64; ``grad`` is ``[n,3]`` or None.

    ``c1``/``c2`` are narrowed to float32 first, as pybind narrows the python
    doubles to C ``float`` at tet_spheres.cpp:208-216.
    """
    c1 = float(np.float32(c1))
    c2 = float(np.float32(c2))
    T = cache.tets
    m = T.shape[0]
    F = deformation_gradient(np.asarray(x, dtype=np.float32), T, cache.Dminv)   # [m,3,3]
    H = cache.L @ F.reshape(m, 9)                                               # (L F)_e
    Es = 0.5 * float(np.sum(H * H))
    J = det3(F)
    pen, dpen = _penalty(J, int(order))
    Eb = float(np.sum(pen))
    E = c1 * Es + c2 * Eb
    if not want_grad:
        return E, Es, Eb, None
    Q = (cache.L.T @ H).reshape(m, 3, 3)                                        # L^T (L F)
    P = c1 * Q + c2 * dpen[:, None, None] * cofactor3(F)                        # dE/dF
    dDs = P @ np.transpose(cache.Dminv, (0, 2, 1))                              # dE/dDs, columns -> x1..x3
    g = np.zeros((cache.n, 3))
    for k in range(3):
        np.add.at(g, T[:, k + 1], dDs[:, :, k])
    np.add.at(g, T[:, 0], -dDs.sum(axis=2))
    return E, Es, Eb, float(grad_output) * g


def energy(x, cache: _Cache, c1: float, c2: float, order: int) -> float:
    return energy_and_grad(x, cache, c1, c2, order, want_grad=False)[0]


def finite_difference_grad(x, cache, c1, c2, order, idx, h=1e-6):
    """Central differences of the float64 energy at flat dof indices ``idx``
    (x is treated as float64 here: this checks the oracle's own calculus)."""
    x0 = np.asarray(x, dtype=np.float64).reshape(-1).copy()
    T = cache.tets
    m = T.shape[0]

    def E(xf):
        p = xf.reshape(-1, 3)[T]
        Ds = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)
        F = Ds @ cache.Dminv
        H = cache.L @ F.reshape(m, 9)
        pen, _ = _penalty(det3(F), int(order))
        return float(np.float32(c1)) * 0.5 * np.sum(H * H) + float(np.float32(c2)) * np.sum(pen)

    out = np.empty(len(idx))
    for t, i in enumerate(idx):
        xp, xm = x0.copy(), x0.copy()
        xp[i] += h
        xm[i] -= h
        out[t] = (E(xp) - E(xm)) / (2 * h)
    return out


def _abs_chain(Fa, cache: _Cache):
    """Push a nonnegative per-tet field through ``|G|^T |L|^T |L|``."""
    T = cache.tets
    m = T.shape[0]
    Da = np.abs(cache.Dminv)
    g = np.empty((m, 4, 3))
    g[:, 1:, :] = Da
    g[:, 0, :] = Da.sum(axis=1)
    La = abs(cache.L)
    Ha = La @ Fa.reshape(m, 9)
    Qa = (La.T @ Ha).reshape(m, 3, 3)
    ga = np.zeros((cache.n, 3))
    contrib = np.einsum("mij,maj->mai", Qa, g)
    for a in range(4):
        np.add.at(ga, T[:, a], contrib[:, a])
    return Ha, ga, g


def tolerance_scales(x, cache: _Cache):
    """Absolute scales of SURVEY.md 8(c) for the *reference formulation*:
    ``A = 0.5 |x|^T |M| |x|`` and ``|| |M| |x| ||_2``, evaluated through the
    factors as ``|G|^T |L|^T |L| |G| |x|`` (an upper bound on the true
    ``|M||x|``).  The fp32 ``M``-based path can only be held to
    ``8*eps*c1*A`` absolute -- see SURVEY.md F11."""
    T = cache.tets
    xa = np.abs(np.asarray(x, dtype=np.float64).reshape(-1, 3))
    Da = np.abs(cache.Dminv)
    g = np.empty((T.shape[0], 4, 3))
    g[:, 1:, :] = Da
    g[:, 0, :] = Da.sum(axis=1)
    Fa = np.einsum("mai,maj->mij", xa[T], g)          # |G||x|
    _, ga, _ = _abs_chain(Fa, cache)
    A = 0.5 * float(np.sum(xa * ga))
    return A, float(np.linalg.norm(ga))


def factored_tolerances(x, cache: _Cache, c1, c2, order, rtol=1e-5, gamma_ulps=16.0):
    """Tolerances for a *factored* fp32 evaluation (``L (G x)`` with ``G x``
    formed from edge vectors) against this float64 oracle.

    fp32 model: ``F`` carries an entrywise error ``<= gamma * (|Ds| |Dm^-1|)``,
    ``gamma = gamma_ulps * 2^-24``; it propagates linearly through ``L``,
    ``L^T`` and ``G^T``.  Returns ``(tol_E, tol_g)``:

    ``tol_E = rtol*(c1*Es + c2*Eb) + c1*(gamma*|H|.|Ha| + gamma^2*|Ha|^2/2)
              + c2*gamma*sum_e p*|J_e|^(p-1)*perm(|F_e|)``
    ``tol_g = rtol*(c1*|g_s| + c2*|g_b|) + c1*gamma*| |G|^T|L|^T|L| Fa |_2
              + c2*8*gamma*| |G|^T ((|dpen| + |d2pen|*perm) cof_abs) |_2``

    Near the rest state this is orders of magnitude tighter than SURVEY.md
    8(c)'s band for the reference formulation (``1e-5 rel + 8 eps c1 A``, where
    the fp32 ``M`` loses everything to cancellation); the GPU tests assert both.
    """
    c1 = float(np.float32(c1))
    c2 = float(np.float32(c2))
    gamma = gamma_ulps * 2.0 ** -24
    T = cache.tets
    m = T.shape[0]
    p = np.asarray(x, dtype=np.float32).astype(np.float64).reshape(-1, 3)[T]
    Ds = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)
    F = Ds @ cache.Dminv
    Fa = np.abs(Ds) @ np.abs(cache.Dminv)
    H = cache.L @ F.reshape(m, 9)
    Ha, ga, _ = _abs_chain(Fa, cache)
    Es = 0.5 * float(np.sum(H * H))
    pen, dpen = _penalty(det3(F), int(order))
    Eb = float(np.sum(pen))
    nH, nHa = float(np.linalg.norm(H)), float(np.linalg.norm(Ha))
    tol_E = rtol * (c1 * Es + c2 * Eb) + c1 * (gamma * nH * nHa + 0.5 * gamma * gamma * nHa * nHa)
    # gradient pieces, separately normed so cancellation between them cannot hide error
    Q = (cache.L.T @ H).reshape(m, 3, 3)
    Pb = dpen[:, None, None] * cofactor3(F)
    DT = np.transpose(cache.Dminv, (0, 2, 1))
    gs = np.zeros((cache.n, 3))
    gb = np.zeros((cache.n, 3))
    for P, out in ((Q, gs), (Pb, gb)):
        d = P @ DT
        for k in range(3):
            np.add.at(out, T[:, k + 1], d[:, :, k])
        np.add.at(out, T[:, 0], -d.sum(axis=2))
    # barrier term: det F carries an absolute error ~ gamma * perm(|F|); it moves
    # dpen by p(p-1)|J|^(p-2) * that (which also covers a flipped J<0 branch at
    # |J| below the error), and the cofactor by ~ gamma * cof_abs.
    Fab = np.abs(F)
    perm = (Fab[:, 0, 2] * Fab[:, 1, 1] * Fab[:, 2, 0] + Fab[:, 0, 1] * Fab[:, 1, 2] * Fab[:, 2, 0]
            + Fab[:, 0, 2] * Fab[:, 1, 0] * Fab[:, 2, 1] + Fab[:, 0, 0] * Fab[:, 1, 2] * Fab[:, 2, 1]
            + Fab[:, 0, 1] * Fab[:, 1, 0] * Fab[:, 2, 2] + Fab[:, 0, 0] * Fab[:, 1, 1] * Fab[:, 2, 2])
    cofa = _cofactor_abs(Fab)
    J = det3(F)
    if int(order) == 2:
        ddpen = 2.0 * np.ones_like(J)
    elif int(order) == 4:
        ddpen = 12.0 * J * J
    else:
        ddpen = np.zeros_like(J)
    Pba = (np.abs(dpen) + ddpen * perm)[:, None, None] * cofa
    gba = np.zeros((cache.n, 3))
    d = Pba @ np.abs(DT)
    for k in range(3):
        np.add.at(gba, T[:, k + 1], d[:, :, k])
    np.add.at(gba, T[:, 0], d.sum(axis=2))
    tol_E += c2 * gamma * float(np.sum(np.abs(dpen) * perm))
    tol_g = (rtol * (c1 * float(np.linalg.norm(gs)) + c2 * float(np.linalg.norm(gb)))
             + c1 * gamma * float(np.linalg.norm(ga)) + c2 * 8 * gamma * float(np.linalg.norm(gba)))
    return tol_E, tol_g


def rounding_error_model(x, cache: _Cache, c1, c2, order):
    """Statistical (variance-propagation) model of the fp32 rounding error of a FACTORED evaluation.

    ``factored_tolerances`` above is a worst-case bound (absolute values summed); measured errors of the HIP
    path sit 3-4 orders below it, so it cannot serve as a regression guard.  Here every fp32 operation is
    given an independent relative error of standard deviation ``u = 2^-24`` and the variances are pushed
    through ``F = Ds Dm^-1``, ``H = L F``, ``Q = L^T H``, ``P = c1 Q + c2 pen' cof F``, ``d = P Dm^-T`` and the
    per-vertex sums.  Returns ``(std_E, std_g)`` with ``std_g`` of shape ``[n]``: the predicted standard
    deviation of the error of each vertex' gradient (Euclidean norm over x, y, z).  The tests assert the
    measured errors against a fixed multiple of these (calibrated on MI355X, profiles/r02_parity.txt).
    """
    u2 = (2.0 ** -24) ** 2
    c1 = float(np.float32(c1))
    c2 = float(np.float32(c2))
    T = cache.tets
    m = T.shape[0]
    p = np.asarray(x, dtype=np.float32).astype(np.float64).reshape(-1, 3)[T]
    Ds = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)
    Di = cache.Dminv
    F = Ds @ Di
    vF = u2 * np.einsum("mik,mkj->mij", Ds * Ds, Di * Di)             # sum_k (Ds_ik Dminv_kj)^2
    L = cache.L.tocsr()
    L2 = L.multiply(L).tocsr()
    Ff = F.reshape(m, 9)
    H = L @ Ff
    vH = L2 @ vF.reshape(m, 9) + u2 * (L2 @ (Ff * Ff))
    Q = L.T @ H
    vQ = L2.T @ vH + u2 * (L2.T @ (H * H))
    J = det3(F)
    pen, dpen = _penalty(J, int(order))
    Cf = cofactor3(F)
    F2 = F * F
    # det: six triple products, each with its own rounding, plus the sensitivity to F (d det / dF = cof)
    vJ = np.sum(Cf * Cf * vF, axis=(1, 2)) + u2 * (
        F2[:, 0, 2] * F2[:, 1, 1] * F2[:, 2, 0] + F2[:, 0, 1] * F2[:, 1, 2] * F2[:, 2, 0] + F2[:, 0, 2] * F2[:, 1, 0] * F2[:, 2, 1]
        + F2[:, 0, 0] * F2[:, 1, 2] * F2[:, 2, 1] + F2[:, 0, 1] * F2[:, 1, 0] * F2[:, 2, 2] + F2[:, 0, 0] * F2[:, 1, 1] * F2[:, 2, 2])
    if int(order) == 2:
        ddpen = np.where(J < 0, 2.0, 0.0)
    elif int(order) == 4:
        ddpen = np.where(J < 0, 12.0 * J * J, 0.0)
    else:
        ddpen = np.zeros_like(J)
    # cofactor entries C_ij = F_a F_d - F_b F_c
    vC = np.empty_like(F)
    for i in range(3):
        for j in range(3):
            r = [a for a in range(3) if a != i]
            c = [b for b in range(3) if b != j]
            a_, d_, b_, c_ = (r[0], c[0]), (r[1], c[1]), (r[0], c[1]), (r[1], c[0])
            Fa, Fd, Fb, Fc = F[:, a_[0], a_[1]], F[:, d_[0], d_[1]], F[:, b_[0], b_[1]], F[:, c_[0], c_[1]]
            vC[:, i, j] = (Fd ** 2 * vF[:, a_[0], a_[1]] + Fa ** 2 * vF[:, d_[0], d_[1]] + Fc ** 2 * vF[:, b_[0], b_[1]]
                           + Fb ** 2 * vF[:, c_[0], c_[1]] + u2 * ((Fa * Fd) ** 2 + (Fb * Fc) ** 2))
    vPb = (ddpen ** 2 * vJ)[:, None, None] * Cf ** 2 + (dpen ** 2)[:, None, None] * vC + u2 * (dpen[:, None, None] * Cf) ** 2
    Pm = c1 * Q.reshape(m, 3, 3) + c2 * dpen[:, None, None] * Cf
    vP = c1 * c1 * vQ.reshape(m, 3, 3) + c2 * c2 * vPb + u2 * Pm ** 2
    d = Pm @ np.transpose(Di, (0, 2, 1))
    vd = np.einsum("mij,mkj->mik", vP, Di * Di) + u2 * np.einsum("mij,mkj->mik", Pm * Pm, Di * Di)
    vg = np.zeros(cache.n)
    f0 = -d.sum(axis=2)
    for k in range(3):
        np.add.at(vg, T[:, k + 1], vd[:, :, k].sum(axis=1) + u2 * (d[:, :, k] ** 2).sum(axis=1))
    np.add.at(vg, T[:, 0], vd.sum(axis=(1, 2)) + u2 * (f0 ** 2).sum(axis=1) + u2 * (d ** 2).sum(axis=(1, 2)))
    Es = 0.5 * float(np.sum(H * H))
    Eb = float(np.sum(pen))
    # the two sums are accumulated in fp32 per tile (~1000 tets: lanes, then a wave tree) and in double across tiles:
    # relative error ~ 2u of each tile's partial sum
    part = min(1.0, 1024.0 / max(m, 1))
    vE = c1 * c1 * (float(np.sum(H * H * vH)) + 4 * u2 * Es * Es * part + u2 * 0.25 * float(np.sum(H ** 4))) \
        + c2 * c2 * (float(np.sum(dpen ** 2 * vJ)) + u2 * float(np.sum(pen ** 2)) + 4 * u2 * Eb * Eb * part)
    # second-order term: E_s = 1/2 |H + dH|^2 carries the (always positive) bias 1/2 sum var(H), which dominates at
    # and near the rest state where H itself vanishes
    bias = c1 * 0.5 * float(np.sum(vH))
    return float(np.sqrt(vE)) + bias, np.sqrt(vg)
