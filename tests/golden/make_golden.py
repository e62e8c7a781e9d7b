"""Generate the committed golden fixtures.  Run in the AUTHORING container only:

    python tests/golden/make_golden.py

It imports the one runnable fragment of the reference,
/root/reference/geometry/mesh_utils.py::compute_G_matrix (:38-69), by file
path (its package __init__ pulls pypgo, which is not installed), and reads the
reference's only real tet mesh, /root/reference/tssplat_ext/a.veg.  Neither
exists on the GPU box, hence the outputs are committed:

* g_matrix_golden.npz -- per-tet 9x12 gradient operators from the REFERENCE
  function on (a) 192 tets of a.veg (with the vertices they touch, float32
  positions promoted to float64 exactly as tet_spheres.cpp:252-255 does) and
  (b) a full kuhn_ball(2).  Pins the oracle's `G` (oracle/tet_energy_oracle.py).
* adam_uniform_golden.npz -- six steps of the REFERENCE optimiser class (utils/optimizer.py:4-89) in
  float64, with and without grad_limit: pins oracle/adam_uniform_oracle.py.
* surface_golden.npz -- the reference's surface extraction, surface gather and vertex normals (+ autograd
  gradient), see surface_golden() below: pins oracle/surface_oracle.py.
* aveg_mesh.npz -- a.veg converted to the extension's input layout
  (float32 [n,3], int32 [m,4], 0-based): the "real TetWild-quality mesh"
  fixture of SURVEY.md 8(d).
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True        # (nothing is written into /root/reference, not even a bytecode cache)

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tssplat_amd import scenes  # noqa: E402

REF = "/root/reference"


def main():
    spec = importlib.util.spec_from_file_location("ref_mesh_utils", f"{REF}/geometry/mesh_utils.py")
    mu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mu)

    v, t = scenes.read_veg(f"{REF}/tssplat_ext/a.veg")
    v32 = v.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "aveg_mesh.npz"), rest=v32, tets=t.astype(np.int32))

    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(t.shape[0], size=192, replace=False))
    sub_t = t[pick]
    used, inv = np.unique(sub_t, return_inverse=True)
    sub_v = v32[used]
    sub_t = inv.reshape(-1, 4).astype(np.int32)
    G_a = mu.compute_G_matrix(sub_v.astype(np.float64), sub_t.astype(np.int64))

    kv, kt = scenes.kuhn_ball(2)
    kv32 = kv.astype(np.float32)
    G_k = mu.compute_G_matrix(kv32.astype(np.float64), kt.astype(np.int64))

    np.savez_compressed(
        os.path.join(HERE, "g_matrix_golden.npz"),
        aveg_rest=sub_v, aveg_tets=sub_t, aveg_tet_ids=pick.astype(np.int32), aveg_G=G_a,
        kuhn2_rest=kv32, kuhn2_tets=kt.astype(np.int32), kuhn2_G=G_k,
    )
    # ---- AdamUniform golden trajectory from the reference class itself (utils/optimizer.py:4-89) ----
    import torch
    spec = importlib.util.spec_from_file_location("ref_optimizer", f"{REF}/utils/optimizer.py")
    ro = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ro)
    out = {}
    for name, kw in (("plain", dict(lr=0.2)), ("limited", dict(lr=0.2, grad_limit=True, grad_limit_values=[0.05, 0.01],
                                                               grad_limit_iters=[3]))):
        g = torch.Generator().manual_seed(11)
        p = torch.nn.Parameter(torch.randn(257, 3, generator=g, dtype=torch.float64))
        opt = ro.AdamUniform([p], **kw)
        grads, traj = [], []
        for it in range(6):
            gr = torch.randn(257, 3, generator=g, dtype=torch.float64) * (10.0 ** (it - 2))
            p.grad = gr.clone()
            opt.step()
            grads.append(gr.numpy().copy())
            traj.append(p.detach().numpy().copy())
        out[f"{name}_p0"] = (traj[0] * 0 + 0)  # placeholder replaced below
        out[f"{name}_grads"] = np.stack(grads)
        out[f"{name}_traj"] = np.stack(traj)
    g = torch.Generator().manual_seed(11)
    out["p0"] = torch.randn(257, 3, generator=g, dtype=torch.float64).numpy()
    for k in ("plain_p0", "limited_p0"):
        out.pop(k)
    np.savez_compressed(os.path.join(HERE, "adam_uniform_golden.npz"), **out)
    surface_golden(mu)
    print("wrote", os.listdir(HERE))


def surface_golden(mu):
    """surface_golden.npz -- the REFERENCE get_surface_vf (geometry/mesh_utils.py:5-35) on a.veg and a Kuhn
    ball, and the REFERENCE TetMeshGeometryForwardData (geometry/tetmesh_geometry.py:27-66: surface gather +
    _compute_vertex_normal) with its torch-autograd gradient, float64 on the CPU.  The module only imports with
    its absent dependencies stubbed (trimesh, xatlas, omegaconf: never called here); pypgo / tet_spheres
    resolve to this repository's stand-ins, which the class under test does not touch either."""
    import types
    import warnings
    import torch
    for name, attrs in {"trimesh": {}, "xatlas": {}, "omegaconf": {"OmegaConf": object, "open_dict": object, "DictConfig": dict}}.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    sys.path.insert(0, REF)
    pkg = types.ModuleType("geometry")
    pkg.__path__ = [f"{REF}/geometry"]
    sys.modules["geometry"] = pkg
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        tg = importlib.import_module("geometry.tetmesh_geometry")
    warnings.filterwarnings("ignore", message="Using torch.cross without specifying the dim")

    out = {}
    v, t = scenes.read_veg(f"{REF}/tssplat_ext/a.veg")
    kv, kt = scenes.kuhn_ball(4)
    for tag, verts, tets in (("aveg", v, t), ("kuhn4", kv, kt)):
        vid, faces = mu.get_surface_vf(tets.astype(np.int64))
        out[f"{tag}_vid"], out[f"{tag}_faces"] = vid.astype(np.int32), faces.astype(np.int32)
        rng = np.random.default_rng(5)
        x = verts.astype(np.float32).astype(np.float64) + 0.03 * rng.standard_normal(verts.shape)
        if tag == "kuhn4":   # a degenerate fan: collapse every face around one surface vertex -> zero normal -> (0,0,1)
            sv = int(vid[7])
            ring = np.unique(vid[faces[(faces == 7).any(axis=1)]])
            x[ring] = x[sv]
        tet_v = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        data = tg.TetMeshGeometryForwardData(tet_v, torch.from_numpy(tets.astype(np.int64)), torch.from_numpy(vid.astype(np.int64)),
                                             torch.from_numpy(faces.astype(np.int64)))
        nrm = data._compute_vertex_normal()
        w_n = torch.tensor(rng.standard_normal(tuple(nrm.shape)))
        w_p = torch.tensor(rng.standard_normal(tuple(nrm.shape)))
        loss = (nrm * w_n).sum() + (data.v_pos * w_p).sum()
        loss.backward()
        out[f"{tag}_x"] = x
        out[f"{tag}_v_pos"] = data.v_pos.detach().numpy()
        out[f"{tag}_nrm"] = nrm.detach().numpy()
        out[f"{tag}_w_n"], out[f"{tag}_w_p"] = w_n.numpy(), w_p.numpy()
        out[f"{tag}_grad_tet_v"] = tet_v.grad.numpy()
    out["kuhn4_tets"] = kt.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "surface_golden.npz"), **out)


if __name__ == "__main__":
    main()
