"""Minimal `pypgo` stand-in: the tet-mesh container and Vega I/O the reference needs.

The reference imports libpgo's Python module at the top of trainer.py:1, energies/smooth_barrier.py:1,
geometry/tetmesh_geometry.py:2 and geometry/tetrahedron_mesh.py:1, but on the energy path it only uses
it as a tet-mesh container with .veg I/O (geometry/tetrahedron_mesh.py:14-24,70-91):

    create_tetmesh_from_file, create_tetmesh, get_tetmesh_vertex_positions,
    get_tetmesh_element_indices, update_tetmesh_vertices, save_tetmesh_to_file

Those six are implemented here in numpy (same argument order; positions ``[n,3]`` float64, elements
``[m,4]`` int32, files 1-based Vega with the material/region footer libpgo writes, cf.
/root/reference/tssplat_ext/a.veg).  Everything else libpgo offers -- isotropic remeshing, the
tet-sphere edge surface, matrices -- is NOT provided: those calls raise NotImplementedError naming the
reference call site, so a missing piece is loud, never silent.  If the real pypgo is installed it
shadows this package only when it comes first on sys.path.
"""
from __future__ import annotations

import numpy as np

from tssplat_amd import scenes as _scenes

__all__ = ["TetMesh", "create_tetmesh_from_file", "create_tetmesh", "get_tetmesh_vertex_positions",
           "get_tetmesh_element_indices", "update_tetmesh_vertices", "save_tetmesh_to_file"]


class TetMesh:
    """Vertices, tets and the (E, nu, density) triple libpgo carries along (tetrahedron_mesh.py:21-24)."""

    def __init__(self, vertices, elements, E=1e9, nu=0.45, density=1000.0):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.elements = np.ascontiguousarray(elements, dtype=np.int32).reshape(-1, 4)
        if self.elements.size and (self.elements.min() < 0 or self.elements.max() >= self.vertices.shape[0]):
            raise ValueError("tet index out of range")
        self.E, self.nu, self.density = float(E), float(nu), float(density)


def create_tetmesh_from_file(filename) -> TetMesh:
    """tetrahedron_mesh.py:15 -- read a Vega .veg file (1-based on disk)."""
    v, t = _scenes.read_veg(filename)
    E, nu, density = 1e9, 0.45, 1000.0
    with open(filename) as fh:
        for line in fh:
            if line.strip().upper().startswith("ENU"):
                tok = [s.strip() for s in line.split(",")]
                density, E, nu = float(tok[1]), float(tok[2]), float(tok[3])
    return TetMesh(v, t, E, nu, density)


def create_tetmesh(vertices, elements, E=1e9, nu=0.45, density=1000.0) -> TetMesh:
    """tetrahedron_mesh.py:22-23 -- flat float32 positions, flat int32 0-based tets."""
    return TetMesh(np.asarray(vertices, dtype=np.float64).reshape(-1, 3), np.asarray(elements).reshape(-1, 4),
                   E, nu, density)


def get_tetmesh_vertex_positions(tetmesh: TetMesh) -> np.ndarray:
    """tetrahedron_mesh.py:16 -- ``[n,3]`` float64 copy."""
    return tetmesh.vertices.copy()


def get_tetmesh_element_indices(tetmesh: TetMesh) -> np.ndarray:
    """tetrahedron_mesh.py:17 -- ``[m,4]`` int32 copy, 0-based."""
    return tetmesh.elements.copy()


def update_tetmesh_vertices(tetmesh: TetMesh, vertices) -> TetMesh:
    """tetrahedron_mesh.py:72 -- a NEW mesh with the same connectivity and material."""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    if v.shape != tetmesh.vertices.shape:
        raise ValueError("vertex array has the wrong shape")
    return TetMesh(v, tetmesh.elements, tetmesh.E, tetmesh.nu, tetmesh.density)


def save_tetmesh_to_file(tetmesh: TetMesh, filename) -> None:
    """tetrahedron_mesh.py:84-85 -- Vega .veg with libpgo's material/region footer."""
    _scenes.write_veg(filename, tetmesh.vertices, tetmesh.elements)
    with open(filename, "a") as fh:
        fh.write(f"\n*MATERIAL defaultMaterial\nENU, {tetmesh.density:g}, {tetmesh.E:g}, {tetmesh.nu:g}\n\n"
                 "*REGION\nallElements, defaultMaterial\n")


def __getattr__(name):
    sites = {
        "create_trimeshgeo": "geometry/tetmesh_geometry.py:288",
        "mesh_isotropic_remeshing": "geometry/tetmesh_geometry.py:291",
        "trimeshgeo_get_vertices": "geometry/tetmesh_geometry.py:293",
        "trimeshgeo_get_triangles": "geometry/tetmesh_geometry.py:294",
        "create_tetsphere_edge_surface": "geometry/tetmesh_fish.py:76",
    }
    if name in sites:
        def missing(*_a, **_k):
            raise NotImplementedError(
                f"pypgo.{name} (used at {sites[name]}) is mesh preparation outside the energy hot path and is not "
                "provided by this stand-in; install libpgo's pypgo for it")
        return missing
    raise AttributeError(f"module 'pypgo' has no attribute {name!r}")
