#!/usr/bin/env python3
"""Fixtures from the reference's own meshes (run in the authoring container, where /root/reference exists; the GPU box only
sees the committed .npz files):

* ``s1_sphere.npz``  -- /root/reference/mesh_data/s.1.obj verbatim: the template sphere every tet-sphere of the reference starts
  from (config/gso.yaml: template_surface_sphere_path; geometry/tetmesh_geometry.py:280-284), 1 500 vertices / 2 996 triangles.
  Coned to its centroid it is the literal "single tet-sphere (~3k tets, mesh_data/s.1.obj)" of BASELINE.json config 1
  (s1_cone, one vertex of valence 2 996).
* ``mario_mesh.npz`` -- /root/reference/mesh_data/mario_example/model.obj (a Google Scanned Objects mesh, CC-BY 4.0, 10 060
  vertices / 20 116 triangles; the one object the reference ships), centred, scaled into the unit ball and decimated by
  vertex clustering on a 48^3 grid to keep the fixture small.  The target object of the config-5 run (tools/train_object.py).

    python tests/golden/make_mesh_goldens.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mesh_data"


def read_obj(path):
    v, f = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                v.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(t.split("/")[0]) - 1 for t in line.split()[1:]]
                for k in range(1, len(idx) - 1):          # fan-triangulate polygons
                    f.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(v, dtype=np.float64), np.asarray(f, dtype=np.int64)


def cluster_decimate(v, f, cells):
    """Vertex clustering: vertices of one grid cell collapse to their mean; triangles that lose a corner are dropped."""
    lo, hi = v.min(0), v.max(0)
    key = np.minimum(((v - lo) / (hi - lo).max() * cells).astype(np.int64), cells - 1)
    flat = (key[:, 0] * cells + key[:, 1]) * cells + key[:, 2]
    uniq, inv = np.unique(flat, return_inverse=True)
    nv = np.zeros((uniq.size, 3))
    np.add.at(nv, inv, v)
    nv /= np.bincount(inv)[:, None]
    nf = inv[f]
    keep = (nf[:, 0] != nf[:, 1]) & (nf[:, 1] != nf[:, 2]) & (nf[:, 0] != nf[:, 2])
    nf = nf[keep]
    # drop duplicate triangles (same vertex set)
    _, first = np.unique(np.sort(nf, axis=1), axis=0, return_index=True)
    return nv, nf[np.sort(first)]


def main():
    v, f = read_obj(os.path.join(REF, "s.1.obj"))
    assert v.shape == (1500, 3) and f.shape == (2996, 3)
    np.savez_compressed(os.path.join(HERE, "s1_sphere.npz"), vertices=v.astype(np.float32), faces=f.astype(np.int32))
    v, f = read_obj(os.path.join(REF, "mario_example", "model.obj"))
    assert v.shape[0] == 10060 and f.shape[0] == 20116
    c = 0.5 * (v.min(0) + v.max(0))
    v = (v - c) / np.linalg.norm(v - c, axis=1).max()          # unit ball around the bounding-box centre
    dv, df = cluster_decimate(v, f, 48)
    np.savez_compressed(os.path.join(HERE, "mario_mesh.npz"), vertices=dv.astype(np.float32), faces=df.astype(np.int32),
                        source=np.array("mesh_data/mario_example/model.obj (GSO, CC-BY 4.0): centred, unit ball, 48^3 vertex clustering"))
    print("s1_sphere:", 1500, 2996, "| mario:", dv.shape[0], "vertices", df.shape[0], "triangles")


if __name__ == "__main__":
    main()
