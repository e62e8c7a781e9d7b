"""CPU oracle (TEST INFRASTRUCTURE, never on the product path) for the surface glue of SURVEY.md 8(f) row 2.

numpy float64 restatements of

* ``get_surface_vf``                 /root/reference/geometry/mesh_utils.py:5-35
* ``tet_v[surface_vid]``             /root/reference/geometry/tetmesh_geometry.py:33
* ``_compute_vertex_normal``         /root/reference/geometry/tetmesh_geometry.py:39-66

plus the adjoints torch autograd derives for them.  Pinned by tests/golden/surface_golden.npz, which
tests/golden/make_golden.py produces by running the REFERENCE functions themselves (float64, CPU) in the
authoring container: surface extraction bit-exact, normals and their autograd gradients to 1e-12.
"""
from __future__ import annotations

import numpy as np

# orientation of the face opposite local vertex k (mesh_utils.py:8-13)
_PATTERNS = np.array([[1, 2, 3], [0, 3, 2], [0, 1, 3], [0, 2, 1]])


def surface_vf(tets: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """(surface vertex ids ascending, triangles in compacted ids) -- mesh_utils.py:5-35."""
    tets = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    org = np.concatenate([tets[:, p] for p in _PATTERNS], axis=0)              # :7-14, k-major like np.vstack
    key = np.sort(org, axis=1)                                                 # :17
    uniq, first, counts = np.unique(key, axis=0, return_index=True, return_counts=True)   # :19-21
    once = counts == 1                                                         # :23
    vid = np.unique(uniq[once])                                                # :26
    faces = np.searchsorted(vid, org[first][once])                             # :28-33 (dict lookup == rank)
    return vid.astype(np.int32), faces.astype(np.int32)


def surface_positions(tet_v: np.ndarray, surface_vid: np.ndarray) -> np.ndarray:
    return np.asarray(tet_v, dtype=np.float64)[np.asarray(surface_vid, dtype=np.int64)]     # tetmesh_geometry.py:33


def surface_positions_backward(grad_v_pos: np.ndarray, surface_vid: np.ndarray, n_tet_vertices: int) -> np.ndarray:
    g = np.zeros((n_tet_vertices, 3))
    np.add.at(g, np.asarray(surface_vid, dtype=np.int64), np.asarray(grad_v_pos, dtype=np.float64))
    return g


def _raw_normals(v_pos, faces):
    v = np.asarray(v_pos, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])            # :40-48
    n = np.zeros_like(v)
    for c in range(3):                                                         # :51-54
        np.add.at(n, f[:, c], fn)
    return n


def vertex_normals(v_pos: np.ndarray, faces: np.ndarray) -> np.ndarray:
    n = _raw_normals(v_pos, faces)
    ok = (n * n).sum(axis=1, keepdims=True) > 1e-20                            # :57-60
    n = np.where(ok, n, np.array([0.0, 0.0, 1.0]))
    return n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-12)     # F.normalize, :61


def vertex_normals_backward(v_pos: np.ndarray, faces: np.ndarray, grad_nrm: np.ndarray) -> np.ndarray:
    """(d v_nrm / d v_pos)^T grad_nrm, derived by hand (checked against torch autograd of the reference)."""
    v = np.asarray(v_pos, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    g = np.asarray(grad_nrm, dtype=np.float64)
    n = _raw_normals(v, f)
    nn = (n * n).sum(axis=1, keepdims=True)
    ok = nn > 1e-20
    ln = np.maximum(np.sqrt(np.where(ok, nn, 1.0)), 1e-12)
    nh = n / ln
    h = np.where(ok, (g - nh * (nh * g).sum(axis=1, keepdims=True)) / ln, 0.0)
    G = h[f[:, 0]] + h[f[:, 1]] + h[f[:, 2]]
    e1, e2 = v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
    d1, d2 = np.cross(e2, G), np.cross(G, e1)
    out = np.zeros_like(v)
    np.add.at(out, f[:, 1], d1)
    np.add.at(out, f[:, 2], d2)
    np.add.at(out, f[:, 0], -(d1 + d2))
    return out
