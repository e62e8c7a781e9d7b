"""Synthetic tet-sphere scenes and tet-mesh file I/O.

Pure numpy, no GPU.  These generators define the workloads that SURVEY.md
section 8(d) names; `bench.py`, the tests and the oracle all draw from here so
that every leg sees bit-identical inputs for a given seed.

Layout contract (mirrors the reference's multi-sphere concatenation,
/root/reference/geometry/tetmesh_geometry.py:310-331): spheres are stacked
one after another, sphere ``s`` owns a contiguous vertex range, tet indices
are offset by the running vertex count, positions are float32 ``[n, 3]`` and
tets are int32 ``[m, 4]`` with 0-based indices.
"""
from __future__ import annotations

import itertools
import os
from dataclasses import dataclass

import numpy as np

__all__ = [
    "TetScene",
    "kuhn_ball",
    "cone_sphere",
    "icosphere_surface",
    "replicate_spheres",
    "make_scene",
    "deform",
    "read_veg",
    "write_veg",
]


@dataclass
class TetScene:
    """A batch of tet-spheres in the reference's flat layout."""

    rest: np.ndarray            # float32 [n, 3] rest positions
    tets: np.ndarray            # int32   [m, 4]
    sphere_vertex_offsets: np.ndarray  # int64 [S+1]
    sphere_tet_offsets: np.ndarray     # int64 [S+1]
    radii: np.ndarray           # float64 [S] scale applied to each sphere

    @property
    def n_vertices(self) -> int:
        return int(self.rest.shape[0])

    @property
    def n_tets(self) -> int:
        return int(self.tets.shape[0])

    @property
    def n_spheres(self) -> int:
        return int(self.radii.shape[0])

    def slice_spheres(self, lo: int, hi: int) -> "TetScene":
        """Spheres ``[lo, hi)`` as a scene of their own (vertex ids re-based): what one rank owns when the batch
        is sharded by whole spheres."""
        v0, v1 = int(self.sphere_vertex_offsets[lo]), int(self.sphere_vertex_offsets[hi])
        t0, t1 = int(self.sphere_tet_offsets[lo]), int(self.sphere_tet_offsets[hi])
        return TetScene(rest=self.rest[v0:v1], tets=(self.tets[t0:t1] - v0).astype(np.int32),
                        sphere_vertex_offsets=self.sphere_vertex_offsets[lo:hi + 1] - v0,
                        sphere_tet_offsets=self.sphere_tet_offsets[lo:hi + 1] - t0, radii=self.radii[lo:hi])


def _orient_positive(verts: np.ndarray, tets: np.ndarray) -> np.ndarray:
    """Swap local vertices 1<->2 of negatively oriented tets so det(Dm) > 0."""
    p = verts[tets].astype(np.float64)
    dm = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)
    neg = np.linalg.det(dm) < 0
    out = tets.copy()
    out[neg, 1], out[neg, 2] = tets[neg, 2], tets[neg, 1]
    return out


def kuhn_ball(k: int, blend: float = 0.5) -> tuple[np.ndarray, np.ndarray]:
    """Rounded-cube tet-sphere: (k+1)^3 lattice, 6 Kuhn tets per cell.

    SURVEY.md 8(d): vertex id ``(i*(k+1)+j)*(k+1)+l`` on [-1,1]^3; every cube
    is cut into the 6 Freudenthal simplices (one per axis permutation); the
    lattice is then pushed towards the unit ball with
    ``p <- p * ((1-blend) + blend * |p|_inf / |p|_2)``.
    Returns float64 ``[n,3]`` vertices and int32 ``[m,4]`` positively oriented
    tets.  k=8 -> n=729, m=3072; k=19 -> n=8000, m=41154.
    """
    if k < 1:
        raise ValueError("k must be >= 1")
    g = np.linspace(-1.0, 1.0, k + 1)
    ii, jj, ll = np.meshgrid(g, g, g, indexing="ij")
    verts = np.stack([ii.ravel(), jj.ravel(), ll.ravel()], axis=1)
    n2 = np.linalg.norm(verts, axis=1)
    ninf = np.abs(verts).max(axis=1)
    scale = np.where(n2 > 0, (1.0 - blend) + blend * ninf / np.maximum(n2, 1e-300), 1.0)
    verts = verts * scale[:, None]

    def vid(i, j, l):
        return (i * (k + 1) + j) * (k + 1) + l

    ci, cj, cl = np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij")
    ci, cj, cl = ci.ravel(), cj.ravel(), cl.ravel()
    tets = []
    for perm in itertools.permutations(range(3)):
        cur = [ci.copy(), cj.copy(), cl.copy()]
        corners = [vid(*cur)]
        for ax in perm:
            cur[ax] = cur[ax] + 1
            corners.append(vid(*cur))
        tets.append(np.stack(corners, axis=1))
    # interleave so the 6 tets of a cell are adjacent in memory
    tets = np.stack(tets, axis=1).reshape(-1, 4).astype(np.int32)
    tets = _orient_positive(verts, tets)
    return verts, tets


def icosphere_surface(subdiv: int = 3) -> tuple[np.ndarray, np.ndarray]:
    """Unit icosphere surface (float64 verts, int32 outward triangles)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
         [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]],
        dtype=np.float64)
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9],
         [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2],
         [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
         [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(subdiv):
        cache: dict[tuple[int, int], int] = {}
        vl = list(v)
        nf = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                p = vl[a] + vl[b]
                vl.append(p / np.linalg.norm(p))
                cache[key] = len(vl) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v = np.array(vl)
        f = np.array(nf, dtype=np.int64)
    return v, f.astype(np.int32)


def cone_sphere(surface_v: np.ndarray, surface_f: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Cone every surface triangle to the centroid: one hub vertex of huge valence.

    The contention stress fixture of SURVEY.md 8(d) (``s1_cone`` when fed the
    reference's template sphere, an icosphere otherwise).
    """
    c = surface_v.mean(axis=0, keepdims=True)
    verts = np.concatenate([surface_v, c], axis=0)
    hub = surface_v.shape[0]
    tets = np.concatenate(
        [np.full((surface_f.shape[0], 1), hub, dtype=np.int32), surface_f.astype(np.int32)], axis=1)
    tets = _orient_positive(verts, tets)
    return verts, tets


def delaunay_ball(n_points: int, seed: int = 0, min_quality: float = 0.08) -> tuple[np.ndarray, np.ndarray]:
    """Unstructured fixture: Delaunay tetrahedralisation of random points in the unit ball, slivers removed.

    Unlike the Kuhn lattice this has irregular valence (3 .. ~40 tets per vertex), interior holes where
    slivers were dropped (tets with fewer than four neighbours anywhere in the mesh) and no index
    locality.  Quality = 6 sqrt(2) V / l_rms^3 (1 for a regular tet); dropping tets keeps the mesh
    face-manifold.  Needs scipy (tests / tools only).
    """
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((n_points, 3))
    p *= (rng.uniform(0, 1, size=(n_points, 1)) ** (1 / 3)) / np.linalg.norm(p, axis=1, keepdims=True)
    tets = Delaunay(p).simplices.astype(np.int32)
    tets = _orient_positive(p, tets)
    q = p[tets]
    vol = np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0]) / 6.0
    edges = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    l2 = sum(((q[:, a] - q[:, b]) ** 2).sum(axis=1) for a, b in edges) / 6.0
    quality = 6.0 * np.sqrt(2.0) * vol / l2 ** 1.5
    return p, np.ascontiguousarray(tets[quality >= min_quality])


def replicate_spheres(verts: np.ndarray, tets: np.ndarray, n_spheres: int,
                      seed: int = 0, sphere_range: tuple[int, int] | None = None) -> TetScene:
    """Stack ``n_spheres`` scaled/translated copies of one template tet mesh.

    Sphere ``s``: radius ``r_s ~ U(0.05, 0.30)``, centre ``c_s ~ U(-0.7, 0.7)^3``
    from ``numpy.random.default_rng(seed)``.  ``sphere_range=(lo, hi)`` builds only the spheres
    ``[lo, hi)`` of that batch -- same radii and centres as in the full scene, vertex ids re-based -- i.e. exactly
    ``replicate_spheres(...).slice_spheres(lo, hi)`` without materialising the rest (what one rank of a sharded job owns).
    """
    rng = np.random.default_rng(seed)
    radii = rng.uniform(0.05, 0.30, size=n_spheres)
    centres = rng.uniform(-0.7, 0.7, size=(n_spheres, 3))
    lo, hi = (0, n_spheres) if sphere_range is None else sphere_range
    if not 0 <= lo <= hi <= n_spheres:
        raise ValueError(f"sphere_range {sphere_range} outside [0, {n_spheres}]")
    k = hi - lo
    nv, nt = verts.shape[0], tets.shape[0]
    rest = np.empty((k * nv, 3), dtype=np.float32)
    allt = np.empty((k * nt, 4), dtype=np.int32)
    v32 = verts.astype(np.float64)
    for j, s in enumerate(range(lo, hi)):
        rest[j * nv:(j + 1) * nv] = (v32 * radii[s] + centres[s]).astype(np.float32)
        allt[j * nt:(j + 1) * nt] = tets + np.int32(j * nv)
    return TetScene(
        rest=rest,
        tets=allt,
        sphere_vertex_offsets=np.arange(k + 1, dtype=np.int64) * nv,
        sphere_tet_offsets=np.arange(k + 1, dtype=np.int64) * nt,
        radii=radii[lo:hi],
    )


def template_mesh(kind: str, seed: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """The per-sphere template of a named workload: ``kuhnK``, ``cone``, ``delaunayN`` (N random points), ``aveg`` (the
    reference's TetWild-quality fixture /root/reference/tssplat_ext/a.veg, 4 500 vertices / 22 120 tets, kept as
    tests/golden/aveg_mesh.npz, centred and scaled into the unit ball like the other templates)."""
    if kind.startswith("kuhn"):
        return kuhn_ball(int(kind[4:]))
    if kind == "cone":
        return cone_sphere(*icosphere_surface(3))
    if kind.startswith("delaunay"):
        return delaunay_ball(int(kind[8:]), seed=seed)
    if kind == "aveg":
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "aveg_mesh.npz")
        g = np.load(path)
        v = g["rest"].astype(np.float64).reshape(-1, 3)
        v = v - 0.5 * (v.min(axis=0) + v.max(axis=0))
        return v / np.linalg.norm(v, axis=1).max(), _orient_positive(v, g["tets"].astype(np.int32).reshape(-1, 4))
    raise ValueError(f"unknown scene kind {kind!r}")


def make_scene(kind: str, n_spheres: int, seed: int = 0, sphere_range: tuple[int, int] | None = None) -> TetScene:
    """Named workloads: ``kuhn8`` (3 072 tets/sphere), ``kuhn19`` (41 154), ``kuhnK`` for any K, ``cone`` (icosphere coned
    to its centre), ``delaunayN`` (N random points), ``aveg`` (the reference's a.veg).  ``sphere_range``: see
    :func:`replicate_spheres`."""
    v, t = template_mesh(kind, seed)
    return replicate_spheres(v, t, n_spheres, seed=seed, sphere_range=sphere_range)


def deform(scene: TetScene, sigma: float, seed: int = 1, vertex_range: tuple[int, int, int] | None = None) -> np.ndarray:
    """``x = X_rest + sigma * r_s * N(0,1)`` in float32.  ``vertex_range=(v0, v1, n_total)``: ``scene`` is the
    slice ``[v0, v1)`` of a scene of ``n_total`` vertices -- return that slice of the FULL scene's deformed positions (the noise
    stream is drawn for the whole batch, so that a rank's positions do not depend on how the batch is sharded)."""
    rng = np.random.default_rng(seed)
    per_vertex_r = np.repeat(scene.radii, np.diff(scene.sphere_vertex_offsets))
    if vertex_range is None:
        noise = rng.standard_normal(scene.rest.shape)
    else:
        v0, v1, n_total = vertex_range
        if v0:
            rng.standard_normal((v0, 3))             # (advance the stream; block-wise to bound memory)
        noise = rng.standard_normal((v1 - v0, 3))
    return (scene.rest.astype(np.float64) + sigma * per_vertex_r[:, None] * noise).astype(np.float32)


# --------------------------------------------------------------------------- #
# Vega .veg tet meshes (the format behind TetSpheres(filename),
# /root/reference/tssplat_ext/tet_spheres/tet_spheres.cpp:108-117).
# --------------------------------------------------------------------------- #

def read_veg(path: str | os.PathLike) -> tuple[np.ndarray, np.ndarray]:
    """Read ``*VERTICES`` / ``*ELEMENTS TET`` from a Vega file.

    Indices on disk are 1-based (or 0-based if the file says so by using id 0);
    returns float64 ``[n,3]`` and 0-based int32 ``[m,4]``.
    """
    verts: list[list[float]] = []
    tets: list[list[int]] = []
    vid: list[int] = []
    mode = None
    header_left = 0
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            if line.startswith("*"):
                key = line.upper()
                if key.startswith("*VERTICES"):
                    mode, header_left = "v", 1
                elif key.startswith("*ELEMENTS"):
                    mode, header_left = "e", 2
                else:
                    mode = None
                continue
            if mode is None:
                continue
            if header_left:
                header_left -= 1
                if mode == "e" and header_left == 1 and line.upper() != "TET":
                    raise ValueError(f"{path}: only TET elements are supported, got {line!r}")
                continue
            tok = line.replace(",", " ").split()
            if mode == "v":
                vid.append(int(tok[0]))
                verts.append([float(tok[1]), float(tok[2]), float(tok[3])])
            else:
                tets.append([int(tok[1]), int(tok[2]), int(tok[3]), int(tok[4])])
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    t = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    base = min(vid) if vid else 1
    return v, (t - base).astype(np.int32)


def write_veg(path: str | os.PathLike, verts: np.ndarray, tets: np.ndarray) -> None:
    """Write a 1-based Vega tet mesh readable by :func:`read_veg`."""
    with open(path, "w") as fh:
        fh.write("# Vega mesh file.\n")
        fh.write(f"# {verts.shape[0]} vertices, {tets.shape[0]} elements\n\n*VERTICES\n")
        fh.write(f"{verts.shape[0]} 3 0 0\n")
        for i, p in enumerate(np.asarray(verts, dtype=np.float64)):
            fh.write(f"{i + 1} {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n")
        fh.write("\n*ELEMENTS\nTET\n")
        fh.write(f"{tets.shape[0]} 4 0\n")
        for i, t in enumerate(np.asarray(tets, dtype=np.int64)):
            fh.write(f"{i + 1} {t[0] + 1} {t[1] + 1} {t[2] + 1} {t[3] + 1}\n")


# ---- cameras for the renderer benches (tools/bench_raster.py, tools/bench_pipeline.py) ----
def orbit_mvps(n_views: int, distance: float = 3.0, fov_deg: float = 40.0, near: float = 0.5, far: float = 8.0,
               elevation_deg: float = 20.0) -> np.ndarray:
    """``n_views`` model-view-projection matrices (float32 ``[n, 4, 4]``, OpenGL clip space) on a circle around the origin --
    the kind of batch /root/reference/data/*.py hands to ``MeshRasterizer.forward(mvp, ...)``.  (The renderer oracle keeps its
    own copy: product code never imports ``oracle/``.)"""
    f = 1.0 / np.tan(np.radians(fov_deg) / 2.0)
    proj = np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, (far + near) / (near - far), 2 * far * near / (near - far)], [0, 0, -1, 0]])
    el = np.radians(elevation_deg)
    out = []
    for k in range(n_views):
        az = 2 * np.pi * k / n_views
        eye = distance * np.array([np.cos(el) * np.sin(az), np.sin(el), np.cos(el) * np.cos(az)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, [0.0, 1.0, 0.0])
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        view = np.eye(4)
        view[0, :3], view[1, :3], view[2, :3] = right, up, -fwd
        view[:3, 3] = -view[:3, :3] @ eye
        out.append(proj @ view)
    return np.stack(out).astype(np.float32)


def dataset_mvps(n_views: int, radius: float = 4.0, fov_deg: float = 39.3077, near: float = 0.001, far: float = 10.0) -> np.ndarray:
    """The camera set of the reference's image data (/root/reference/data/render_dataset.py:15-57, :96-148): ``n_views`` eyes on a
    golden-ratio spiral over the sphere of ``radius`` around the origin, looking at the origin (up = z, or y when the view
    direction comes within 22.5 degrees of z), projection = the script's ``perspective()`` (note its flipped y row).  Returns
    ``projection @ view`` per view, float32 ``[n, 4, 4]`` -- the ``mvp`` batch trainer.py hands to ``MeshRasterizer.forward``."""
    golden = (1 + 5 ** 0.5) / 2
    i = np.arange(n_views)
    theta = 2 * np.pi * i / golden
    phi = np.arccos(1 - 2 * i / n_views)
    eyes = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1) * radius
    t = np.tan(np.radians(fov_deg) * 0.5)
    proj = np.zeros((4, 4))
    proj[0, 0], proj[1, 1] = 1 / t, -1 / t
    proj[2, 2], proj[3, 2], proj[2, 3] = -(far + near) / (far - near), -1, -(2 * far * near) / (far - near)
    out = []
    for eye in eyes:
        d = eye / np.linalg.norm(eye)
        up = np.array([0.0, 0.0, 1.0])
        if abs(np.dot(up, d)) > np.cos(np.pi / 8):
            up = np.array([0.0, 1.0, 0.0])
        look = -d
        right = np.cross(look, up)
        right /= np.linalg.norm(right)
        up = np.cross(right, look)
        up /= np.linalg.norm(up)
        view = np.eye(4)
        view[0, :3], view[1, :3], view[2, :3] = right, up, -look
        view[0, 3], view[1, 3], view[2, 3] = -np.dot(right, eye), -np.dot(up, eye), np.dot(look, eye)
        out.append(proj @ view)
    return np.stack(out).astype(np.float32)


def transform_pos(mvp: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """``[v, 1] @ mvp^T`` per view, float32 (mesh_rasterizer.py:57-78, non-ortho branch)."""
    posw = np.concatenate([np.asarray(pos, dtype=np.float32), np.ones((pos.shape[0], 1), dtype=np.float32)], axis=1)
    return np.matmul(posw[None], np.transpose(np.asarray(mvp, dtype=np.float32), (0, 2, 1))).astype(np.float32)
