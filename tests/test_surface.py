"""Surface glue: oracle pinned to the reference's own outputs (CPU), host extraction
through the C ABI (CPU, no device), HIP kernels against the oracle (GPU)."""
import os

import numpy as np
import pytest

from oracle import surface_oracle as SO


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "surface_golden.npz"))


def test_oracle_surface_extraction_is_the_reference_bit_for_bit(gold, aveg):
    _, tets = aveg
    vid, faces = SO.surface_vf(tets)
    assert np.array_equal(vid, gold["aveg_vid"]) and np.array_equal(faces, gold["aveg_faces"])
    vid, faces = SO.surface_vf(gold["kuhn4_tets"])
    assert np.array_equal(vid, gold["kuhn4_vid"]) and np.array_equal(faces, gold["kuhn4_faces"])
    assert faces.shape[0] == 6 * 2 * 4 * 4      # 4 x 4 squares per cube side, two triangles each


@pytest.mark.parametrize("tag", ["aveg", "kuhn4"])
def test_oracle_normals_and_gradient_match_reference_autograd(gold, tag):
    x, vid, faces = gold[f"{tag}_x"], gold[f"{tag}_vid"], gold[f"{tag}_faces"]
    v_pos = SO.surface_positions(x, vid)
    assert np.array_equal(v_pos, gold[f"{tag}_v_pos"])
    nrm = SO.vertex_normals(v_pos, faces)
    assert np.abs(nrm - gold[f"{tag}_nrm"]).max() <= 1e-12
    if tag == "kuhn4":      # the collapsed fan: the reference substitutes (0, 0, 1)
        assert np.array_equal(gold["kuhn4_nrm"][7], [0.0, 0.0, 1.0]) and np.array_equal(nrm[7], [0.0, 0.0, 1.0])
    g_vp = SO.vertex_normals_backward(v_pos, faces, gold[f"{tag}_w_n"]) + gold[f"{tag}_w_p"]
    g = SO.surface_positions_backward(g_vp, vid, x.shape[0])
    ref = gold[f"{tag}_grad_tet_v"]
    assert np.abs(g - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_host_extraction_through_the_c_abi(gold, aveg):
    from tssplat_amd import geometry
    _, tets = aveg
    vid, faces = geometry.get_surface_vf(tets)
    assert vid.dtype == np.int32 and faces.dtype == np.int32
    assert np.array_equal(vid, gold["aveg_vid"]) and np.array_equal(faces, gold["aveg_faces"])
    vid, faces = geometry.get_surface_vf(gold["kuhn4_tets"])
    assert np.array_equal(vid, gold["kuhn4_vid"]) and np.array_equal(faces, gold["kuhn4_faces"])
    # closed surface: every edge is used once in each direction
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]).astype(np.int64)
    fwd = np.sort(e[:, 0] * 10**6 + e[:, 1])
    bwd = np.sort(e[:, 1] * 10**6 + e[:, 0])
    assert np.array_equal(fwd, bwd)
    # errors are loud
    bad = gold["kuhn4_tets"].copy()
    bad[0, 0] = -1
    with pytest.raises(RuntimeError, match="out of range"):
        geometry.get_surface_vf(bad)
    nm = np.array([[0, 1, 2, 3], [0, 2, 1, 4], [0, 1, 2, 5]], dtype=np.int32)
    with pytest.raises(RuntimeError, match="non-manifold"):
        geometry.get_surface_vf(nm)
    v0, f0 = geometry.get_surface_vf(np.zeros((0, 4), np.int32))
    assert v0.size == 0 and f0.shape == (0, 3)


# --------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["aveg", "kuhn4"])
def test_gpu_forward_data_matches_oracle(gold, tag):
    import torch
    from tssplat_amd import geometry
    x, vid, faces = gold[f"{tag}_x"].astype(np.float32), gold[f"{tag}_vid"], gold[f"{tag}_faces"]
    tet_v = torch.from_numpy(x).cuda().requires_grad_(True)
    vid_t, f_t = torch.from_numpy(vid).cuda(), torch.from_numpy(faces).cuda()
    data = geometry.TetMeshGeometryForwardData(tet_v, None, vid_t, f_t)
    assert torch.equal(data.v_pos, tet_v.detach()[vid_t.long()])                    # the gather is exact
    nrm = data._compute_vertex_normal()
    v64 = SO.surface_positions(x, vid)
    ref_n = SO.vertex_normals(v64, faces)
    # fp32 vs float64 on identical fp32 inputs: cancellation in the cross products scales with the fan's area sum
    assert np.abs(nrm.detach().cpu().numpy() - ref_n).max() <= 2e-5
    w_n, w_p = gold[f"{tag}_w_n"], gold[f"{tag}_w_p"]
    loss = (nrm * torch.from_numpy(w_n).float().cuda()).sum() + (data.v_pos * torch.from_numpy(w_p).float().cuda()).sum()
    loss.backward()
    g_vp = SO.vertex_normals_backward(v64, faces, w_n.astype(np.float32)) + w_p.astype(np.float32)
    ref_g = SO.surface_positions_backward(g_vp, vid, x.shape[0])
    got = tet_v.grad.cpu().numpy()
    assert np.abs(got - ref_g).max() <= 2e-4 * np.abs(ref_g).max()
    assert np.all(got[np.setdiff1d(np.arange(x.shape[0]), vid)] == 0)               # interior vertices: exactly zero
    # second construction reuses the cached topology handle and is bitwise repeatable
    data2 = geometry.TetMeshGeometryForwardData(tet_v, None, vid_t, f_t)
    assert data2._ops is data._ops and torch.equal(data2._compute_vertex_normal(), nrm)


@pytest.mark.gpu
def test_gpu_surface_against_reference_torch_chain_and_permute():
    """Same inputs through the reference's op chain (index, cross, scatter_add_, where, normalize) in torch on
    the GPU: the fused kernels must agree with it to fp32 rounding, forward and backward."""
    import torch
    import torch.nn.functional as F
    from tssplat_amd import geometry, scenes
    v, t = scenes.kuhn_ball(12)
    vid, faces = geometry.get_surface_vf(t)
    x = torch.from_numpy(v.astype(np.float32)).cuda()
    x = x + 0.02 * torch.randn_like(x)
    vid_t, f_t = torch.from_numpy(vid).cuda(), torch.from_numpy(faces).cuda()

    def chain(tet_v):   # tetmesh_geometry.py:33,39-66
        vp = tet_v[vid_t.long()]
        i0, i1, i2 = f_t[:, 0].long(), f_t[:, 1].long(), f_t[:, 2].long()
        fn = torch.cross(vp[i1] - vp[i0], vp[i2] - vp[i0], dim=1)
        n = torch.zeros_like(vp)
        for i in (i0, i1, i2):
            n = n.scatter_add(0, i[:, None].repeat(1, 3), fn)
        n = torch.where((n * n).sum(-1, keepdim=True) > 1e-20, n, torch.tensor([0.0, 0.0, 1.0], device=n.device))
        return vp, F.normalize(n, dim=1)

    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    w = torch.randn(len(vid), 3, device="cuda")
    vp_a, n_a = chain(a)
    (n_a * w).sum().backward()
    d = geometry.TetMeshGeometryForwardData(b, None, vid_t, f_t)
    n_b = d._compute_vertex_normal()
    (n_b * w).sum().backward()
    assert torch.equal(vp_a, d.v_pos)
    assert (n_a - n_b).abs().max() <= 1e-5
    assert (a.grad - b.grad).abs().max() <= 1e-4 * a.grad.abs().max()
    # permute_surface_v (tetmesh_geometry.py:176-182): only surface vertices move, by at most dev / 2
    y = x.clone()
    geometry.permute_surface_v(y, vid_t, 0.01)
    moved = (y - x).abs().max(dim=1).values
    interior = torch.ones(len(x), dtype=torch.bool, device="cuda")
    interior[vid_t.long()] = False
    assert torch.all(moved[interior] == 0) and torch.all(moved <= 0.005 + 1e-7) and moved[vid_t.long()].max() > 0.003
    # loud failures: wrong device / dtype / shape
    with pytest.raises(RuntimeError, match="float32"):
        d._ops.vertex_normals(d.v_pos.double())
    with pytest.raises(RuntimeError, match="must live on"):
        d._ops.positions(x.cpu())
