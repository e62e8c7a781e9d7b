"""The "vanilla PyTorch" CPU energy path -- ORACLE-SIDE, TEST/BASELINE ONLY.

The reference announces a PyTorch version of the extension but does not ship
it: /root/reference/energies/smooth_barrier.py:7 imports (commented out)
``compute_energy, compute_G_matrix, compute_L_matrix, get_torch_sparse_mat``
from a missing ``energies/torch_energies.py``, and README.md:111 lists it as an
open item.  This module is our restatement of that path under those four
names.  It keeps the *reference formulation* -- explicit fp32 sparse
``M = G^T L^T L G`` and ``G``, five sparse products per forward+backward,
exactly the sequence of tet_spheres_cuda.cu:118-263 -- so that

* ``bench.py``'s ``cpu_baseline`` leg can time "the reference's CPU energy path"
  on the GPU box's host cores, and
* the tests can show that our factored HIP kernels sit inside the reference
  formulation's own fp32 error band.

Never imported by the product path.  ``L`` is parity-unpinned (see
oracle/tet_energy_oracle.py).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from . import tet_energy_oracle as _o

__all__ = ["compute_G_matrix", "compute_L_matrix", "get_torch_sparse_mat",
           "compute_energy", "compute_energy_backward", "TorchTetSpheres"]


def compute_G_matrix(rest: np.ndarray, tets: np.ndarray) -> sp.csr_matrix:
    """Global 9m x 3n gradient operator in float64 (tet_spheres.cpp:149)."""
    return _o.gradient_operator_sparse(rest, tets)


def compute_L_matrix(tets: np.ndarray) -> sp.csr_matrix:
    """9m x 9m element Laplacian ``L (x) I9`` (tet_spheres.cpp:148, args 1,0)."""
    L = _o.element_laplacian(_o.face_adjacency(tets))
    return sp.kron(L, sp.identity(9), format="csr")


def get_torch_sparse_mat(mat: sp.spmatrix, layout: str = "csr", dtype=torch.float32) -> torch.Tensor:
    """scipy sparse (float64) -> torch sparse fp32 (tet_spheres.cpp:43-45 rounding).

    ``layout="coo"`` is the storage the reference hands to cuSPARSE
    (tet_spheres.cpp:60); ``"csr"`` is the fast CPU layout.
    """
    if layout == "csr":
        m = mat.tocsr()
        return torch.sparse_csr_tensor(
            torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
            torch.from_numpy(m.data).to(dtype), size=m.shape)
    m = mat.tocoo()
    idx = torch.from_numpy(np.stack([m.row, m.col]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(m.data).to(dtype), size=m.shape).coalesce()


def _det(F: torch.Tensor) -> torch.Tensor:
    # six-term expansion, tet_spheres_cuda.cu:21-30
    return (-F[:, 0, 2] * F[:, 1, 1] * F[:, 2, 0] + F[:, 0, 1] * F[:, 1, 2] * F[:, 2, 0]
            + F[:, 0, 2] * F[:, 1, 0] * F[:, 2, 1] - F[:, 0, 0] * F[:, 1, 2] * F[:, 2, 1]
            - F[:, 0, 1] * F[:, 1, 0] * F[:, 2, 2] + F[:, 0, 0] * F[:, 1, 1] * F[:, 2, 2])


def _cof(F: torch.Tensor) -> torch.Tensor:
    # tet_spheres_cuda.cu:32-46
    r0 = torch.stack([F[:, 1, 1] * F[:, 2, 2] - F[:, 1, 2] * F[:, 2, 1],
                      F[:, 1, 2] * F[:, 2, 0] - F[:, 1, 0] * F[:, 2, 2],
                      F[:, 1, 0] * F[:, 2, 1] - F[:, 1, 1] * F[:, 2, 0]], dim=1)
    r1 = torch.stack([F[:, 0, 2] * F[:, 2, 1] - F[:, 0, 1] * F[:, 2, 2],
                      F[:, 0, 0] * F[:, 2, 2] - F[:, 0, 2] * F[:, 2, 0],
                      F[:, 0, 1] * F[:, 2, 0] - F[:, 0, 0] * F[:, 2, 1]], dim=1)
    r2 = torch.stack([F[:, 0, 1] * F[:, 1, 2] - F[:, 0, 2] * F[:, 1, 1],
                      F[:, 0, 2] * F[:, 1, 0] - F[:, 0, 0] * F[:, 1, 2],
                      F[:, 0, 0] * F[:, 1, 1] - F[:, 0, 1] * F[:, 1, 0]], dim=1)
    return torch.stack([r0, r1, r2], dim=1)


class TorchTetSpheres:
    """CPU stand-in for the reference's `TetSpheres` state (tet_spheres.h:9-42):
    fp32 sparse ``GTLTLG`` and ``G`` built once from the rest mesh."""

    def __init__(self, rest: np.ndarray, tets: np.ndarray, layout: str = "csr"):
        rest = np.asarray(rest, dtype=np.float32).reshape(-1, 3)
        tets = np.asarray(tets, dtype=np.int32).reshape(-1, 4)
        self.n = rest.shape[0]
        self.nele = tets.shape[0]
        M, G = _o.biharmonic_matrix(rest, tets)
        self.GTLTLG = get_torch_sparse_mat(M, layout)
        self.G = get_torch_sparse_mat(G, layout)
        self.GT = get_torch_sparse_mat(G.T.tocsr(), layout)


def compute_energy(x: torch.Tensor, ts: TorchTetSpheres, c1: float, c2: float, order: int) -> torch.Tensor:
    """Forward, step for step as tet_spheres_cuda.cu:118-195."""
    xf = x.reshape(-1, 1).to(torch.float32)
    Mx = torch.sparse.mm(ts.GTLTLG, xf) if ts.GTLTLG.layout != torch.sparse_csr else ts.GTLTLG @ xf
    sm = 0.5 * torch.dot(Mx[:, 0], xf[:, 0])                              # :131-157
    F = (ts.G @ xf).reshape(ts.nele, 3, 3)                                # :167
    J = torch.clamp(-_det(F), min=0.0)                                    # :55-56
    Jp = J * J if order == 2 else (J * J * J * J if order == 4 else torch.zeros_like(J))
    bar = Jp.sum()                                                        # :185
    return sm * np.float32(c1) + bar * np.float32(c2)                     # :191


def compute_energy_backward(grad_out, x: torch.Tensor, ts: TorchTetSpheres, c1: float, c2: float,
                            order: int) -> torch.Tensor:
    """Backward, step for step as tet_spheres_cuda.cu:197-263."""
    xf = x.reshape(-1, 1).to(torch.float32)
    g = (ts.GTLTLG @ xf) * np.float32(c1)                                 # :215-216
    F = (ts.G @ xf).reshape(ts.nele, 3, 3)                                # :221
    J = _det(F)
    neg = J < 0
    Jm = -J
    if order == 2:
        Jp = 2.0 * Jm
    elif order == 4:
        Jp = 4.0 * Jm * Jm * Jm
    else:
        Jp = torch.zeros_like(J)
    dF = torch.where(neg[:, None, None], _cof(F) * (-Jp)[:, None, None], torch.zeros_like(F))  # :75-101
    g = g + np.float32(c2) * (ts.GT @ dF.reshape(-1, 1))                  # :246-248
    return (g * float(grad_out)).reshape(x.shape)                         # :257-258
