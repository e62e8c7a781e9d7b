"""Renderer slice: ``tssplat_amd.dr.rasterize`` / ``interpolate`` / ``antialias`` against
oracle/raster_oracle.py.

CPU tests: the oracle's own invariants (watertight shared edges, orthographic interpolation reproduces the pixel grid,
nearest depth, lower id on ties, dropped triangles) and the C ABI's argument checks.  GPU tests (``-m gpu``): triangle
ids BIT-EXACT against the oracle, (u, v, z/w) and interpolated attributes within fp32 tolerance, interpolate backward
against the oracle's float64 scatter.  Parity with nvdiffrast itself is UNPINNED (the library is not in this image)."""
import ctypes as C

import numpy as np
import pytest

from oracle import raster_oracle as R
from tssplat_amd import _capi, scenes


def _quad(z=0.0, w=1.0, lo=-1.0, hi=1.0):
    pos = np.array([[lo, lo, z, 1], [hi, lo, z, 1], [hi, hi, z, 1], [lo, hi, z, 1]], dtype=np.float32)
    pos[:, :3] *= w
    pos[:, 3] = w
    return pos, np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)


def test_full_screen_quad_is_covered_exactly_once():
    pos, tri = _quad()
    for H, W in ((8, 8), (7, 13)):
        key = R.rasterize_ids(pos, tri, H, W)
        assert (key != R.NO_FRAGMENT).all()
        ids = (key & np.uint64(0xFFFFFFFF)).astype(np.int64)
        assert set(np.unique(ids)) == {0, 1}
        # the diagonal pixels (centre exactly on the shared edge when H == W) go to exactly one triangle: the two
        # per-triangle rasterisations are disjoint and together cover everything
        k0 = R.rasterize_ids(pos, tri[:1], H, W) != R.NO_FRAGMENT
        k1 = R.rasterize_ids(pos, tri[1:], H, W) != R.NO_FRAGMENT
        assert not (k0 & k1).any() and (k0 | k1).all()


def test_watertight_fan_at_arbitrary_positions():
    rng = np.random.default_rng(5)
    H = W = 32
    for _ in range(20):
        # a fan of triangles around an interior point: shared edges at random orientations, pixel centres on edges included
        c = rng.uniform(-0.3, 0.3, 2)
        ring = [c + rng.uniform(0.4, 0.7) * np.array([np.cos(a), np.sin(a)])
                for a in 2 * np.pi * np.arange(7) / 7 + rng.uniform(-0.3, 0.3, 7)]          # (every wedge well under 180 degrees)
        pts = np.array([c] + ring)
        pts = np.round(pts * 16) / 16                              # many coordinates land exactly on pixel centres / edges
        pos = np.concatenate([pts, np.zeros((8, 1)), np.ones((8, 1))], axis=1).astype(np.float32)
        tri = np.array([[0, 1 + k, 1 + (k + 1) % 7] for k in range(7)], dtype=np.int32)
        cover = sum((R.rasterize_ids(pos, tri[k:k + 1], H, W) != R.NO_FRAGMENT).astype(int) for k in range(7))
        assert cover.max() <= 1                                    # never twice
        union = R.rasterize_ids(pos, tri, H, W) != R.NO_FRAGMENT
        assert np.array_equal(union, cover == 1)


def test_orthographic_interpolation_reproduces_pixel_centres_and_depth_order():
    H, W = 16, 24
    pos, tri = _quad(z=0.25)
    rast = R.rasterize(pos, tri, (H, W))
    attr = pos[None, :, :2].astype(np.float64)
    out = R.interpolate(attr, rast, tri)
    fx = (np.arange(W) + 0.5) / W * 2 - 1
    fy = (np.arange(H) + 0.5) / H * 2 - 1
    assert np.allclose(out[0, ..., 0], fx[None, :], atol=1e-12) and np.allclose(out[0, ..., 1], fy[:, None], atol=1e-12)
    assert np.allclose(rast[0, ..., 2], 0.25)
    # a nearer, smaller quad in front wins where it covers; an identical copy at the SAME depth loses to the lower id
    pos2, tri2 = _quad(z=-0.5, lo=-0.5, hi=0.5)
    both_pos = np.concatenate([pos, pos2, pos])
    both_tri = np.concatenate([tri, tri2 + 4, tri + 8])
    r2 = R.rasterize(both_pos, both_tri, (H, W))
    ids = r2[0, ..., 3].astype(int) - 1
    inner = (np.abs(fx)[None, :] < 0.5) & (np.abs(fy)[:, None] < 0.5)
    assert np.isin(ids[inner], [2, 3]).all() and np.isin(ids[~inner], [0, 1]).all()
    assert np.allclose(r2[0, ..., 2][inner], -0.5)


def test_dropped_triangles_and_background():
    H = W = 8
    pos, tri = _quad()
    behind = pos.copy()
    behind[:, 3] = -1.0                                             # every vertex behind the eye: nothing to clip, nothing drawn
    assert (R.rasterize(behind, tri[:1], (H, W)) == 0).all()
    far = pos.copy()
    far[:, 2] = 2.0                                                 # z/w = 2: beyond the far plane, every fragment dropped
    assert (R.rasterize(far, tri, (H, W)) == 0).all()
    nan = pos.copy()
    nan[1, 0] = np.nan
    assert (R.rasterize(nan, tri[:1], (H, W)) == 0).all()
    degenerate = np.array([[0, 1, 1]], dtype=np.int32)
    assert (R.rasterize(pos, degenerate, (H, W)) == 0).all()


def _ground_scene():
    """A large ground triangle under a perspective camera at height 1 looking along -z: its near vertices lie BEHIND the eye."""
    near, far, f = 0.1, 50.0, 1.0 / np.tan(np.radians(30.0))
    proj = np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, (far + near) / (near - far), 2 * far * near / (near - far)], [0, 0, -1, 0]])
    view = np.eye(4)
    view[1, 3] = -1.0
    mvp = proj @ view

    def clip(v):
        return (np.concatenate([v, np.ones((len(v), 1))], axis=1) @ mvp.T).astype(np.float32)
    return clip


def test_near_plane_clipping():
    """A triangle with vertices at w <= 0 is clipped against the near plane, not dropped: it covers the same pixels as the
    same surface cut into a piece in front of the camera and a piece that straddles the eye plane, z/w and (u, v) agree with the
    unclipped plane, and a triangle entirely behind the camera draws nothing."""
    H = W = 48
    clip = _ground_scene()
    A, B, C = np.array([-30.0, 0, 8]), np.array([30.0, 0, 8]), np.array([0.0, 0, -40])     # A, B behind the eye (z > 0), C far ahead
    pos = clip(np.stack([A, B, C]))
    assert (pos[:2, 3] <= 0).all() and pos[2, 3] > 0
    whole = R.rasterize(pos, np.array([[0, 1, 2]], np.int32), (H, W))[0]
    assert (whole[..., 3] > 0).sum() > 200                                               # the ground is visible below the horizon
    assert (whole[H // 2 + 2:, :, 3] == 0).all()                                         # ... and nowhere above it
    # the same surface in two pieces: cut along z = -2 (in front of the near plane at z = -0.1)
    t = (-2.0 - A[2]) / (C[2] - A[2])
    P, Q = A + t * (C - A), B + t * (C - B)
    pos2 = clip(np.stack([A, B, C, P, Q]))
    parts = R.rasterize(pos2, np.array([[3, 4, 2], [0, 1, 4], [0, 4, 3]], np.int32), (H, W))[0]
    cover_w, cover_p = whole[..., 3] > 0, parts[..., 3] > 0
    assert (cover_w != cover_p).sum() <= 2                                               # (the cut's end points are rounded to float32)
    both = cover_w & cover_p
    assert np.abs(whole[..., 2] - parts[..., 2])[both].max() < 2e-5                      # same depth plane
    # barycentrics of the unclipped triangle reproduce the world position of every covered pixel's ray on the ground plane
    jj, ii = np.nonzero(cover_w)
    u, v = whole[jj, ii, 0], whole[jj, ii, 1]
    world = u[:, None] * A + v[:, None] * B + (1 - u - v)[:, None] * C
    hit = clip(world)
    ndc = hit[:, :2] / hit[:, 3:4]
    assert np.abs((ndc[:, 0] * 0.5 + 0.5) * W - (ii + 0.5)).max() < 0.02 and np.abs((ndc[:, 1] * 0.5 + 0.5) * H - (jj + 0.5)).max() < 0.02
    # polygon of the clip: two kept / one kept vertices
    assert [k for k, _ in R.clip_near(pos.astype(np.float64))] == ["i", "v", "i"]
    assert [k for k, _ in R.clip_near(pos[[2, 0, 2]].astype(np.float64))].count("v") == 2
    behind_all = clip(np.stack([A, B, np.array([0.0, 0, 3])]))
    assert (R.rasterize(behind_all, np.array([[0, 1, 2]], np.int32), (H, W)) == 0).all()


def test_interpolate_backward_is_the_adjoint():
    rng = np.random.default_rng(2)
    pos, tri = _quad(w=2.0)
    rast = R.rasterize(pos, tri, (6, 6))
    attr = rng.standard_normal((1, 4, 3))
    g = rng.standard_normal((1, 6, 6, 3))
    ga, gr = R.interpolate_backward(attr, rast, tri, g)
    d = rng.standard_normal(attr.shape)
    lhs = np.sum(R.interpolate(attr + 1e-6 * d, rast, tri) * g) - np.sum(R.interpolate(attr - 1e-6 * d, rast, tri) * g)
    assert abs(lhs / 2e-6 - np.sum(ga * d)) <= 1e-6 * abs(np.sum(ga * d)) + 1e-9
    r2 = rast.copy()
    r2[..., 0] += 1e-6
    du = (np.sum(R.interpolate(attr, r2, tri) * g) - np.sum(R.interpolate(attr, rast, tri) * g)) / 1e-6
    assert abs(du - gr[..., 0].sum()) <= 1e-5 * abs(du) + 1e-8


def test_c_abi_rejects_bad_arguments():
    lib = _capi.load()
    assert lib.tsamd_rasterize_workspace_bytes(8, 1000, 512, 512) == 8 * 512 * 512 * 8 + 8 * 1000 * 16 + 32      # keys, snapped vertices, view flags
    assert lib.tsamd_rasterize_workspace_bytes(-1, 3, 4, 4) == -1
    assert lib.tsamd_rasterize(None, 1, 3, None, 1, 4, 4, None, None, None, None) == 1          # null workspace / output
    assert b"null" in lib.tsamd_last_error()
    assert lib.tsamd_rasterize(None, 1, 3, None, 1, 20000, 4, None, None, None, None) == 1
    assert lib.tsamd_rasterize(None, 1, 3, None, 1, 9000, 4, None, None, None, None) == 1       # beyond the 8192-pixel guard-band limit
    assert b"8192" in lib.tsamd_last_error()
    assert lib.tsamd_rasterize(None, 1, 3, None, 1 << 24, 4, 4, None, None, None, None) == 1    # ids travel as float32: <= 2^24 - 1 triangles
    assert b"2^24" in lib.tsamd_last_error()
    assert lib.tsamd_interpolate(None, 2, 3, 3, None, None, 1, 4, 4, 4, None, None) == 1     # attr_batch neither 1 nor batch
    assert lib.tsamd_interpolate(None, 1, 3, 3, None, None, -1, 1, 4, 4, None, None) == 1    # negative triangle count
    assert lib.tsamd_interpolate_backward(None, 1, 3, 0, None, None, 1, 1, 4, 4, None, None, None, None) == 1
    # empty images are fine without any pointer
    assert lib.tsamd_rasterize(None, 0, 0, None, 0, 0, 0, None, None, None, None) == 0


def _octasphere(levels=2, radius=0.7):
    """A closed triangle mesh (subdivided octahedron), counter-clockwise seen from outside."""
    v = [np.array(p, dtype=np.float64) for p in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    for _ in range(levels):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        f = nf
    return (radius * np.array(v)).astype(np.float32), np.array(f, dtype=np.int32)


def test_edge_partners_of_closed_and_open_meshes():
    _, tri = _octasphere(1)
    opp = R.edge_partners(tri)
    assert (opp >= 0).all()                                       # closed: every edge has a partner
    for t in range(tri.shape[0]):
        for e in range(3):
            a, b = tri[t, (e + 1) % 3], tri[t, (e + 2) % 3]
            o = opp[3 * t + e]
            # the partner triangle holds the edge and the far vertex, and is not t itself
            holders = [u for u in range(tri.shape[0]) if u != t and {a, b, o} == set(tri[u])]
            assert len(holders) == 1
    _, quad = _quad()
    oq = R.edge_partners(quad)
    assert list(oq) == [-1, 3, -1, -1, -1, 1]                     # only the diagonal (0, 2) is shared
    # three triangles on one edge: partner = the lowest-numbered other (triangle, edge)
    fan = np.array([[0, 1, 2], [1, 0, 3], [0, 1, 4]], dtype=np.int32)
    of = R.edge_partners(fan)
    assert of[3 * 0 + 2] == 3 and of[3 * 1 + 2] == 2 and of[3 * 2 + 2] == 2


def test_antialias_of_an_axis_aligned_edge_is_exact_coverage():
    # ONE triangle covering x < x_edge on the whole screen: along every row the boundary pixel pair is blended so that the alpha
    # sum equals the covered area exactly (the closer surface reaches t pixels beyond its last covered centre).  (With a quad
    # the rows whose boundary pixel belongs to the triangle that does not own the silhouette edge are left alone -- the
    # published algorithm only looks at the edges of the pixel's own triangle.)
    H, W = 6, 16
    for x_edge in (-0.81, -0.33, 0.02, 0.4, 0.77):
        pos = np.array([[-5, 0, 0, 1], [x_edge, -3, 0, 1], [x_edge, 3, 0, 1]], dtype=np.float32)
        tri = np.array([[0, 1, 2]], dtype=np.int32)
        rast = R.rasterize(pos, tri, (H, W))
        alpha = np.clip(rast[..., 3:4], 0, 1)
        aa = R.antialias(alpha, rast, pos[None], tri)
        exact = (float(np.float32(x_edge)) * 0.5 + 0.5) * W * H               # (positions are float32)
        assert abs(aa.sum() - exact) < 1e-9 * W * H
        assert abs(alpha.sum() - exact) > 1e-3 or abs((x_edge * 0.5 + 0.5) * W % 1 - 0.5) < 1e-3     # and it did something


def test_antialias_oracle_gradients_match_finite_differences():
    v, tri = _octasphere(2)
    pos = R.transform_pos(R.orbit_mvps(2), v)
    H = W = 32
    rast = R.rasterize(pos, tri, (H, W))
    rng = np.random.default_rng(0)
    col = rng.random((2, H, W, 3)).astype(np.float32)
    g = rng.standard_normal(col.shape)
    gc, gp = R.antialias_backward(col, rast, pos, tri, g, pos_gradient_boost=1.5)
    assert np.abs(gp).sum() > 0 and (gp[..., 2] == 0).all()

    def loss(c, p):
        return float((R.antialias(c, rast, p, tri) * g).sum())
    touched = np.unique(np.nonzero(np.abs(gp).sum(-1))[1])
    for b, k, c in [(b, int(k), c) for b in range(2) for k in touched[:5] for c in (0, 1, 3)]:
        pp, pm = pos.copy(), pos.copy()
        pp[b, k, c] += 1e-3
        pm[b, k, c] -= 1e-3
        num = (loss(col, pp) - loss(col, pm)) / float(pp[b, k, c] - pm[b, k, c]) * 1.5
        assert abs(gp[b, k, c] - num) <= 2e-3 * max(1.0, abs(num)), (b, k, c, gp[b, k, c], num)
    # colour: the operator is linear in it
    d = rng.standard_normal(col.shape).astype(np.float32) * 0.25
    lin = loss(col + d, pos) - loss(col, pos)
    assert abs(lin - float((gc * d.astype(np.float64)).sum())) <= 1e-5 * max(1.0, abs(lin))


def test_rasterize_backward_oracle_matches_finite_differences():
    v, tri = _octasphere(1)
    pos = R.transform_pos(R.orbit_mvps(1), v)
    H = W = 24
    rast = R.rasterize(pos, tri, (H, W))
    rng = np.random.default_rng(2)
    g = rng.standard_normal(rast.shape)
    g[..., 2:] = 0
    gp = R.rasterize_backward(pos, tri, rast, g)
    key = R.rasterize_ids(pos[0], tri, H, W)

    def loss(p):
        # same winners, barycentrics from the moved positions (the ids are piecewise constant)
        return float((R.resolve(p[0], tri, key)[..., :2] * g[0, ..., :2]).sum())
    for k, c in [(int(k), c) for k in np.unique(tri[(rast[0, ..., 3][rast[0, ..., 3] > 0] - 1).astype(int)])[:6] for c in (0, 1, 3)]:
        pp, pm = pos.copy(), pos.copy()
        pp[0, k, c] += 1e-3
        pm[0, k, c] -= 1e-3
        num = (loss(pp) - loss(pm)) / float(pp[0, k, c] - pm[0, k, c])
        assert abs(gp[0, k, c] - num) <= 1e-3 * max(1.0, abs(num)), (k, c, gp[0, k, c], num)
    assert (gp[..., 2] == 0).all()


def test_c_abi_checks_antialias_arguments():
    lib = _capi.load()
    assert lib.tsamd_antialias_topology_workspace_bytes(-1) == -1
    assert lib.tsamd_antialias_topology_workspace_bytes(10) == 64 * 16 + 30 * 4
    assert lib.tsamd_antialias_topology(None, 5, None, None, None) == 1
    assert lib.tsamd_antialias_topology(None, 0, None, None, None) == 0
    assert lib.tsamd_antialias(None, None, None, None, None, None, 1, 3, 1, 4, 4, 0, None, None) == 1        # no channels
    assert lib.tsamd_antialias(None, None, None, None, None, None, 1, 3, 1, 4, 4, 1, None, None) == 1        # null images
    assert lib.tsamd_antialias(None, None, None, None, None, None, 0, 0, 0, 0, 0, 1, None, None) == 0        # empty image
    assert lib.tsamd_antialias_backward(None, None, None, None, None, None, 0, 0, 0, 0, 0, 1, None, 1.0, None, None, None) == 1   # no output asked for
    assert lib.tsamd_antialias_prepared_bytes(2, 5, 7, 4, 8) == 3 * 256                                       # windows | pair masks | edge flags, each padded
    assert lib.tsamd_antialias_prepared_bytes(-1, 5, 7, 4, 8) == -1
    assert lib.tsamd_antialias_prepare(None, None, None, None, None, 1, 3, 1, 4, 4, None, None) == 1                # null buffers
    assert lib.tsamd_antialias_prepare(None, None, None, None, None, 0, 0, 0, 4, 4, None, None) == 0                # nothing to do
    assert lib.tsamd_rasterize_backward(None, 1, 3, None, 1, 4, 4, None, None, None, None) == 1
    assert b"null" in lib.tsamd_last_error()


# ------------------------------------------------------------------ GPU parity ------------------------------------------------------------------

def _surface_scene(kind, spheres, n_views, seed=0):
    """Boundary triangles of a tet-sphere scene (what get_surface_vf hands the reference's renderer) + orbit cameras."""
    from tssplat_amd import geometry
    sc = scenes.make_scene(kind, spheres, seed=seed)
    vid, faces = geometry.get_surface_vf(sc.tets)
    v = scenes.deform(sc, 0.05, seed=seed + 1)[np.asarray(vid)]
    mvp = R.orbit_mvps(n_views)
    return R.transform_pos(mvp, v), np.asarray(faces, dtype=np.int32), v


@pytest.mark.gpu
@pytest.mark.parametrize("kind,spheres,views,res", [("kuhn4", 3, 2, (48, 64)), ("kuhn8", 6, 3, (128, 128)), ("kuhn19", 2, 1, (256, 256))])
def test_rasterize_ids_bit_exact_and_barycentrics(kind, spheres, views, res):
    import torch
    import tssplat_amd.dr as dr
    pos_clip, tri, _ = _surface_scene(kind, spheres, views)
    ref = R.rasterize(pos_clip, tri, res)
    ctx = dr.RasterizeCudaContext()
    rast, db = dr.rasterize(ctx, torch.from_numpy(pos_clip).cuda(), torch.from_numpy(tri).cuda(), resolution=list(res), grad_db=False)
    assert rast.shape == (views, res[0], res[1], 4) and db.shape[-1] == 0
    got = rast.cpu().numpy().astype(np.float64)
    assert (ref[..., 3] > 0).mean() > 0.02                                       # the scene is in view
    assert np.array_equal(got[..., 3], ref[..., 3]), "triangle ids must be bit-exact"
    hit = ref[..., 3] > 0
    assert np.abs(got[..., :2][hit] - ref[..., :2][hit]).max() <= 2e-4           # fp32 barycentrics of tiny triangles
    assert np.abs(got[..., 2][hit] - ref[..., 2][hit]).max() <= 2e-6
    assert (got[~hit] == 0).all()
    # a second call through the same context (workspace reuse) is identical
    rast2, _ = dr.rasterize(ctx, torch.from_numpy(pos_clip).cuda(), torch.from_numpy(tri).cuda(), resolution=list(res), grad_db=False)
    assert torch.equal(rast, rast2)


@pytest.mark.gpu
def test_rasterize_edge_cases_on_gpu():
    import torch
    import tssplat_amd.dr as dr
    ctx = dr.RasterizeCudaContext()
    rng = np.random.default_rng(3)
    # shared edges through pixel centres, equal-depth duplicates, a vertex behind the eye, NaN, a degenerate and an off-screen triangle
    pts = np.round(rng.uniform(-1.2, 1.2, (40, 2)) * 8) / 8
    w = rng.uniform(0.5, 3.0, 40).astype(np.float32)
    pos = np.concatenate([pts * w[:, None], (rng.uniform(-0.9, 0.9, 40) * w)[:, None], w[:, None]], axis=1).astype(np.float32)
    pos[5, 3] = -0.5
    pos[9, 1] = np.nan
    tri = rng.integers(0, 40, (120, 3)).astype(np.int32)
    tri = np.concatenate([tri, tri[:10], [[1, 1, 2]]]).astype(np.int32)          # duplicates (ties -> lower id) and a degenerate one
    pos = np.stack([pos, pos[::-1].copy()])
    for res in ((16, 16), (33, 20)):
        ref = R.rasterize(pos, tri, res)
        rast, _ = dr.rasterize(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), resolution=list(res), grad_db=False)
        got = rast.cpu().numpy().astype(np.float64)
        assert np.array_equal(got[..., 3], ref[..., 3])
        hit = ref[..., 3] > 0
        assert hit.any() and np.abs(got[..., :3][hit] - ref[..., :3][hit]).max() <= 5e-4
    with pytest.raises(NotImplementedError):
        dr.rasterize(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), resolution=[8, 8])        # grad_db defaults to True
    with pytest.raises(RuntimeError):
        dr.antialias(None, None, None, None)                                                                    # not GPU tensors
    with pytest.raises(RuntimeError):
        dr.rasterize(ctx, torch.from_numpy(pos), torch.from_numpy(tri), resolution=[8, 8], grad_db=False)       # CPU tensors: no fallback


@pytest.mark.gpu
def test_rasterize_large_triangles_take_the_64_bit_walk():
    """Triangles wider than 64 pixels leave the 32-bit edge functions (|E| < 2^30) for the 64-bit walk; mixed with small ones in
    the same wave, partly off-screen, interpenetrating: ids bit-exact against the oracle."""
    import torch
    import tssplat_amd.dr as dr
    ctx = dr.RasterizeCudaContext()
    rng = np.random.default_rng(12)
    H, W = 200, 300
    n = 60
    pts = rng.uniform(-1.6, 1.6, (n, 2))
    w = rng.uniform(0.7, 2.0, n)
    pos = np.concatenate([pts * w[:, None], (rng.uniform(-0.8, 0.8, n) * w)[:, None], w[:, None]], axis=1).astype(np.float32)
    big = rng.integers(0, n, (24, 3))
    small_c = rng.uniform(-0.9, 0.9, (40, 2))
    small = np.concatenate([small_c + rng.uniform(-0.05, 0.05, (40, 2)) for _ in range(3)], axis=0).reshape(3, 40, 2).transpose(1, 0, 2).reshape(-1, 2)
    spos = np.concatenate([small, rng.uniform(-0.5, 0.5, (120, 1)), np.ones((120, 1))], axis=1).astype(np.float32)
    pos = np.concatenate([pos, spos])
    tri = np.concatenate([big, n + np.arange(120).reshape(40, 3)]).astype(np.int32)
    pos[:4] = [[-3, -3, 0.5, 1], [3, -3, 0.5, 1], [3, 3, 0.5, 1], [-3, 3, 0.5, 1]]           # a screen-filling pair behind everything
    tri = np.concatenate([tri, [[0, 1, 2], [0, 2, 3]]]).astype(np.int32)
    ref = R.rasterize(pos[None], tri, (H, W))
    rast, _ = dr.rasterize(ctx, torch.from_numpy(pos[None]).cuda(), torch.from_numpy(tri).cuda(), resolution=[H, W], grad_db=False)
    got = rast.cpu().numpy().astype(np.float64)
    assert (ref[..., 3] > 0).all()                                               # the backdrop covers the screen
    assert len(np.unique(ref[..., 3])) > 30                                      # and many triangles of both kinds are visible
    assert np.array_equal(got[..., 3], ref[..., 3])
    assert np.abs(got[..., :3] - ref[..., :3]).max() <= 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("attr_batch_is_one", [True, False])
def test_interpolate_forward_backward(attr_batch_is_one):
    import torch
    import tssplat_amd.dr as dr
    views, res = 3, (96, 96)
    pos_clip, tri, v = _surface_scene("kuhn8", 5, views)
    ctx = dr.RasterizeCudaContext()
    tri_d = torch.from_numpy(tri).cuda()
    rast, _ = dr.rasterize(ctx, torch.from_numpy(pos_clip).cuda(), tri_d, resolution=list(res), grad_db=False)
    rng = np.random.default_rng(1)
    attr_np = (v[None] if attr_batch_is_one else np.stack([v * (1 + 0.1 * k) for k in range(views)])).astype(np.float32)
    attr = torch.from_numpy(attr_np).cuda().requires_grad_(True)
    rast_in = rast.clone().requires_grad_(True)
    out, da = dr.interpolate(attr, rast_in, tri_d)
    assert out.shape == (views, res[0], res[1], 3) and da.shape[-1] == 0
    rast_np = rast.cpu().numpy()
    ref = R.interpolate(attr_np, rast_np, tri)
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    g = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).cuda())
    ga, gr = R.interpolate_backward(attr_np, rast_np, tri, g)
    assert np.abs(attr.grad.cpu().numpy() - ga).max() <= 2e-5 * np.abs(ga).max()            # fp32 atomics, arbitrary order
    assert np.abs(rast_in.grad.cpu().numpy() - gr).max() <= 2e-5 * max(np.abs(gr).max(), 1e-30)
    assert (rast_in.grad[..., 2:] == 0).all()


@pytest.mark.gpu
def test_topology_table_matches_the_oracle():
    import torch
    import tssplat_amd.dr as dr
    _, tri, _ = _surface_scene("kuhn8", 4, 1)
    rng = np.random.default_rng(4)
    extra = rng.integers(0, 50, (300, 3)).astype(np.int32)                    # a soup: boundary edges, edges with 3+ triangles, repeated vertices
    for t in (tri, extra, np.concatenate([tri[:500], tri[:40]])):
        topo = dr.antialias_construct_topology_hash(torch.from_numpy(np.ascontiguousarray(t)).cuda())
        assert np.array_equal(topo.opp.cpu().numpy(), R.edge_partners(t))


@pytest.mark.gpu
@pytest.mark.parametrize("prepare", [True, False])
@pytest.mark.parametrize("kind,spheres,views,res,channels", [("kuhn4", 3, 2, (64, 64), 1), ("kuhn8", 5, 3, (128, 96), 3)])
def test_antialias_forward_backward(kind, spheres, views, res, channels, prepare, monkeypatch):
    import torch
    import tssplat_amd.dr as dr
    monkeypatch.setattr(dr, "PREPARE_ANTIALIAS", prepare)      # both forms of the analysis (tables per view / everything per pixel pair)
    monkeypatch.setattr(dr, "PAIR_MASKS_FROM_RASTERIZE", channels == 1)   # pair masks from the resolve pass / from a scan of rast
    pos_clip, tri, _ = _surface_scene(kind, spheres, views)
    ctx = dr.RasterizeCudaContext()
    tri_d = torch.from_numpy(tri).cuda()
    pos = torch.from_numpy(pos_clip).cuda().requires_grad_(True)
    rast, _ = dr.rasterize(ctx, pos, tri_d, resolution=list(res), grad_db=False)
    rast_np = rast.detach().cpu().numpy()
    rng = np.random.default_rng(7)
    if channels == 1:
        col_np = np.clip(rast_np[..., 3:4], 0, 1).astype(np.float32)          # the reference's alpha image (mesh_rasterizer.py:106)
    else:
        col_np = rng.random(rast_np.shape[:3] + (channels,)).astype(np.float32)
    col = torch.from_numpy(col_np).cuda().requires_grad_(True)
    out = dr.antialias(col, rast, pos, tri_d, topology_hash=None, pos_gradient_boost=2.0)
    ref = R.antialias(col_np, rast_np, pos_clip, tri)
    n_changed = int((np.abs(ref - col_np).sum(-1) > 0).sum())
    assert n_changed > 20                                                      # silhouettes were found
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 2e-6             # same set of blends (a different decision shows as ~0.1)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).cuda())
    gc, gp = R.antialias_backward(col_np, rast_np, pos_clip, tri, g, pos_gradient_boost=2.0)
    assert np.abs(col.grad.cpu().numpy() - gc).max() <= 1e-5 * max(1.0, np.abs(gc).max())
    # rast's (u, v) carry no gradient here (out does not depend on them), so pos.grad is the antialias term alone
    assert np.abs(gp).max() > 0
    assert np.abs(pos.grad.cpu().numpy() - gp).max() <= 2e-5 * np.abs(gp).max()
    # an explicit topology hash gives the same image (up to the order of the fp32 atomics on pixels with two blends)
    topo = dr.antialias_construct_topology_hash(tri_d)
    out2 = dr.antialias(col.detach(), rast, pos.detach(), tri_d, topology_hash=topo)
    assert (out2 - out.detach()).abs().max().item() <= 1e-6
    # the C ABI without the prepared buffer (prepared_dev = NULL: windows and pair detection per use) takes the same decisions
    from tssplat_amd import _capi
    lib = _capi.load()
    B, H, W, Cn = col.shape
    out3 = torch.empty_like(col)
    cd, pd = col.detach().contiguous(), pos.detach().contiguous()
    _capi.check(lib.tsamd_antialias(cd.data_ptr(), rast.data_ptr(), pd.data_ptr(), None, tri_d.data_ptr(), topo.opp.data_ptr(), B, pd.shape[1],
                                    tri_d.shape[0], H, W, Cn, out3.data_ptr(), None))
    gc3, gp3, gd = torch.empty_like(cd), torch.empty_like(pd), torch.from_numpy(g).cuda()
    _capi.check(lib.tsamd_antialias_backward(cd.data_ptr(), rast.data_ptr(), pd.data_ptr(), None, tri_d.data_ptr(), topo.opp.data_ptr(), B,
                                             pd.shape[1], tri_d.shape[0], H, W, Cn, gd.data_ptr(), 2.0, gc3.data_ptr(), gp3.data_ptr(), None))
    torch.cuda.synchronize()
    assert (out3 - out.detach()).abs().max().item() <= 1e-6
    assert (gc3 - col.grad).abs().max().item() <= 1e-5 * max(1.0, np.abs(gc).max())
    assert (gp3 - pos.grad).abs().max().item() <= 2e-5 * np.abs(gp).max()


@pytest.mark.gpu
def test_rasterize_backward_through_interpolate():
    import torch
    import tssplat_amd.dr as dr
    views, res = 2, (96, 96)
    pos_clip, tri, v = _surface_scene("kuhn8", 4, views)
    ctx = dr.RasterizeCudaContext()
    tri_d = torch.from_numpy(tri).cuda()
    pos = torch.from_numpy(pos_clip).cuda().requires_grad_(True)
    attr_np = v[None].astype(np.float32)
    attr = torch.from_numpy(attr_np).cuda()
    rast, _ = dr.rasterize(ctx, pos, tri_d, resolution=list(res), grad_db=False)
    out, _ = dr.interpolate(attr, rast, tri_d)
    rng = np.random.default_rng(11)
    g = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(g).cuda())
    rast_np = rast.detach().cpu().numpy()
    _, gr = R.interpolate_backward(attr_np, rast_np, tri, g)
    gp = R.rasterize_backward(pos_clip, tri, rast_np, gr)
    got = pos.grad.cpu().numpy()
    assert np.abs(gp).max() > 0 and (got[..., 2] == 0).all()
    # fp32 quotient rule on sub-pixel triangles: the edge functions are differences of nearly equal products
    scale = np.abs(gp).max()
    assert np.abs(got - gp).max() <= 2e-3 * scale
    assert np.abs(got - gp).mean() <= 2e-5 * scale


@pytest.mark.gpu
def test_renderer_operators_on_empty_inputs():
    """No triangles / nothing on screen: rasterize gives the background, interpolate zeros, antialias the identity (and its
    backward hands the upstream gradient straight to the colour); an off-screen mesh leaves no gradient on the positions."""
    import torch
    import tssplat_amd.dr as dr
    ctx = dr.RasterizeCudaContext()
    pos = torch.tensor([[[5.0, 5.0, 0.0, 1.0], [6.0, 5.0, 0.0, 1.0], [5.0, 6.0, 0.0, 1.0]]], device="cuda", requires_grad=True)   # off screen
    for tri in (torch.zeros((0, 3), dtype=torch.int32, device="cuda"), torch.tensor([[0, 1, 2]], dtype=torch.int32, device="cuda")):
        rast, _ = dr.rasterize(ctx, pos, tri, resolution=[16, 24], grad_db=False)
        assert rast.shape == (1, 16, 24, 4) and float(rast.detach().abs().max()) == 0.0
        out, _ = dr.interpolate(pos.detach()[..., :3].contiguous(), rast, tri)
        assert float(out.detach().abs().max()) == 0.0
        col = torch.rand(1, 16, 24, 3, device="cuda", requires_grad=True)
        aa = dr.antialias(col, rast, pos, tri)
        assert torch.equal(aa, col)
        g = torch.randn_like(col)
        pos.grad = None
        aa.backward(g)
        assert torch.equal(col.grad, g)
        assert pos.grad is None or float(pos.grad.abs().max()) == 0.0


def test_oracle_interpolate_treats_foreign_ids_as_background():
    """A rast image whose ids do not belong to the triangle list (longer list, corrupt id, NaN) or a triangle with a vertex
    index outside the attribute array: background in the forward, nothing scattered in the backward."""
    tri = np.array([[0, 1, 2], [0, 2, 7]], dtype=np.int32)          # second triangle: vertex 7 does not exist
    attr = np.arange(12, dtype=np.float64).reshape(1, 4, 3)
    rast = np.zeros((1, 1, 5, 4))
    rast[0, 0, :, 0] = 0.25
    rast[0, 0, :, 1] = 0.5
    rast[0, 0, :, 3] = [1.0, 2.0, 3.0, np.nan, 1e9]                  # ok | bad vertex index | beyond the list | NaN | huge
    out = R.interpolate(attr, rast, tri)
    assert np.allclose(out[0, 0, 0], 0.25 * attr[0, 0] + 0.5 * attr[0, 1] + 0.25 * attr[0, 2])
    assert np.all(out[0, 0, 1:] == 0.0)
    ga, gr = R.interpolate_backward(attr, rast, tri, np.ones((1, 1, 5, 3)))
    assert np.all(gr[0, 0, 1:] == 0.0) and np.any(gr[0, 0, 0] != 0.0)
    assert np.allclose(ga[0].sum(axis=0), 1.0) and np.all(ga[0, 3] == 0.0)   # one pixel's weights, nothing on vertex 3


@pytest.mark.gpu
def test_interpolate_bounds_on_foreign_rast():
    """tsamd_interpolate* now know the triangle count.  Same cases as the oracle test above, on the device, plus an
    EMPTY triangle list with a rast image that still carries ids (no dummy index buffer needed any more)."""
    import torch
    import tssplat_amd.dr as dr
    tri_np = np.array([[0, 1, 2], [0, 2, 7]], dtype=np.int32)
    attr_np = np.arange(12, dtype=np.float32).reshape(1, 4, 3)
    rast_np = np.zeros((1, 1, 5, 4), dtype=np.float32)
    rast_np[0, 0, :, 0] = 0.25
    rast_np[0, 0, :, 1] = 0.5
    rast_np[0, 0, :, 3] = [1.0, 2.0, 3.0, np.nan, 1e9]
    attr = torch.from_numpy(attr_np).cuda().requires_grad_(True)
    rast = torch.from_numpy(rast_np).cuda().requires_grad_(True)
    out, _ = dr.interpolate(attr, rast, torch.from_numpy(tri_np).cuda())
    ref = R.interpolate(attr_np, rast_np, tri_np)
    assert np.allclose(out.detach().cpu().numpy(), ref, atol=1e-6)
    out.backward(torch.ones_like(out))
    ga, gr = R.interpolate_backward(attr_np, rast_np, tri_np, np.ones((1, 1, 5, 3)))
    assert np.allclose(attr.grad.cpu().numpy(), ga, atol=1e-6)
    assert np.allclose(rast.grad.cpu().numpy(), gr, atol=1e-5)
    empty = torch.zeros((0, 3), dtype=torch.int32, device="cuda")
    out0, _ = dr.interpolate(attr.detach(), rast.detach(), empty)
    assert float(out0.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kind,spheres,views,res", [("kuhn8", 4, 8, (32, 32)), ("kuhn8", 4, 16, (24, 40)), ("kuhn4", 6, 66, (16, 16)), ("kuhn8", 4, 5, (32, 32))])
def test_rasterize_with_more_triangles_than_pixels(kind, spheres, views, res):
    """The launch order changes when the depth atomics are dense (more triangles than pixels and a batch that divides among the
    XCDs: every view's workgroups on one XCD): same bit-exact ids, also for batches that are not a multiple of eight and for
    the fallback order (5 views)."""
    import torch
    import tssplat_amd.dr as dr
    pos_clip, tri, _ = _surface_scene(kind, spheres, views)
    assert tri.shape[0] >= res[0] * res[1]
    ref = R.rasterize(pos_clip, tri, res)
    rast, _ = dr.rasterize(dr.RasterizeCudaContext(), torch.from_numpy(pos_clip).cuda(), torch.from_numpy(tri).cuda(), resolution=list(res), grad_db=False)
    out = rast.cpu().numpy()
    assert (ref[..., 3] > 0).mean() > 0.05
    assert np.array_equal(out[..., 3], ref[..., 3])
    assert np.abs(out[..., :3] - ref[..., :3])[ref[..., 3] > 0].max() <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("res", [(64, 64), (50, 128), (33, 20), (7, 192), (128, 96)])
def test_pair_masks_of_the_resolve_pass_equal_a_scan_of_rast(res):
    """tsamd_rasterize's optional by-product (which pixels differ from their right / upper neighbour, 2 bits per pixel) against
    antialias_detect_kernel's scan of the finished image: tiled (width % 64 == 0, also with a ragged last row group) and
    flat resolve kernels."""
    import torch
    from tssplat_amd import _capi
    lib = _capi.load()
    H, W = res
    pos_clip, tri, _ = _surface_scene("kuhn8", 4, 3)
    B, V, T = pos_clip.shape[0], pos_clip.shape[1], tri.shape[0]
    pos = torch.from_numpy(pos_clip).cuda()
    tri_d = torch.from_numpy(tri).cuda()
    ws = torch.empty(int(lib.tsamd_rasterize_workspace_bytes(B, V, H, W)), dtype=torch.uint8, device="cuda")
    rast = torch.empty((B, H, W, 4), device="cuda")
    nm = int(lib.tsamd_pair_masks_bytes(B, H, W))
    assert nm == (B * H * W + 63) // 64 * 16
    masks = torch.full((nm,), 0xAB, dtype=torch.uint8, device="cuda")
    _capi.check(lib.tsamd_rasterize(pos.data_ptr(), B, V, tri_d.data_ptr(), T, H, W, ws.data_ptr(), rast.data_ptr(), masks.data_ptr(), None))
    rast2 = torch.empty_like(rast)
    _capi.check(lib.tsamd_rasterize(pos.data_ptr(), B, V, tri_d.data_ptr(), T, H, W, ws.data_ptr(), rast2.data_ptr(), None, None))
    assert torch.equal(rast, rast2)
    opp = torch.empty(3 * T, dtype=torch.int32, device="cuda")
    tws = torch.empty(int(lib.tsamd_antialias_topology_workspace_bytes(T)), dtype=torch.uint8, device="cuda")
    _capi.check(lib.tsamd_antialias_topology(tri_d.data_ptr(), T, tws.data_ptr(), opp.data_ptr(), None))
    n = int(lib.tsamd_antialias_prepared_bytes(B, V, T, H, W))
    scanned, handed = (torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(2))
    _capi.check(lib.tsamd_antialias_prepare(rast.data_ptr(), pos.data_ptr(), tri_d.data_ptr(), opp.data_ptr(), None, B, V, T, H, W, scanned.data_ptr(), None))
    _capi.check(lib.tsamd_antialias_prepare(rast.data_ptr(), pos.data_ptr(), tri_d.data_ptr(), opp.data_ptr(), masks.data_ptr(), B, V, T, H, W,
                                            handed.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(scanned, handed)
    off = (B * V * 16 + 255) // 256 * 256                         # (the masks sit behind the window table)
    m = scanned[off:off + nm].cpu().numpy().view(np.uint64).reshape(-1, 2)
    assert torch.equal(masks, scanned[off:off + nm]) and m.any()
    # and against numpy on the image itself
    ids = rast[..., 3].cpu().numpy().reshape(B, H, W)
    c0 = np.zeros((B, H, W), bool)
    c1 = np.zeros((B, H, W), bool)
    c0[:, :, :-1] = ids[:, :, 1:] != ids[:, :, :-1]
    c1[:, :-1, :] = ids[:, 1:, :] != ids[:, :-1, :]
    bits = np.zeros((m.shape[0] * 64, 2), bool)
    bits[:B * H * W, 0], bits[:B * H * W, 1] = c0.reshape(-1), c1.reshape(-1)
    want = (bits.reshape(-1, 64, 2).astype(np.uint64) << np.arange(64, dtype=np.uint64)[None, :, None]).sum(axis=1, dtype=np.uint64)
    assert np.array_equal(m, want)


@pytest.mark.gpu
def test_near_plane_clipping_on_gpu():
    """Triangles that straddle the eye plane: ids bit-exact against the oracle's clip (ground plane under a camera, plus a random
    soup pushed through the camera), in a batch whose second view needs no clipping at all."""
    import torch
    import tssplat_amd.dr as dr
    H, W = 96, 80
    clip = _ground_scene()
    rng = np.random.default_rng(11)
    ground = np.array([[-30.0, 0, 8], [30.0, 0, 8], [0.0, 0, -40], [-30, 0, -40], [30, 0, -40], [0, 0, 20]])
    soup = rng.uniform(-3, 3, (60, 3)) + [0, 1, 0]                                   # around the eye at (0, 1, 0)
    v = np.concatenate([ground, soup])
    tri = np.concatenate([[[0, 1, 2], [0, 2, 3], [1, 4, 2], [0, 1, 5]], 6 + rng.integers(0, 60, (150, 3))]).astype(np.int32)
    pos0 = clip(v)
    assert ((pos0[:, 3] <= 0).sum() > 10) and ((pos0[:, 3] > 0).sum() > 10)
    pos1 = clip(v * [1, 1, 0.2] - [0, 0, 6])                                        # everything in front of the camera
    assert (pos1[:, 3] > 0).all()
    pos = np.stack([pos0, pos1])
    ref = R.rasterize(pos, tri, (H, W))
    # (through the "gl" context name of mesh_rasterizer.py:35-36: the same kernels)
    out, _ = dr.rasterize(dr.RasterizeGLContext(), torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), resolution=[H, W], grad_db=False)
    out = out.cpu().numpy()
    assert (ref[0, ..., 3] > 0).sum() > 500
    assert np.array_equal(out[..., 3], ref[..., 3])
    hit = ref[..., 3] > 0
    assert np.abs(out[..., :3] - ref[..., :3])[hit].max() <= 2e-4
    # the clipped triangles took part: some winning triangle of view 0 has a vertex at w <= 0
    ids = np.unique(ref[0, ..., 3][ref[0, ..., 3] > 0]).astype(int) - 1
    assert (pos0[tri[ids], 3] <= 0).any()


@pytest.mark.gpu
def test_near_plane_clipping_in_a_large_batch():
    """The clip kernel scans the per-view flags 64 views at a time: one view that needs clipping among 70 that do not."""
    import torch
    import tssplat_amd.dr as dr
    H, W = 24, 32
    clip = _ground_scene()
    v = np.array([[-30.0, 0, 8], [30.0, 0, 8], [0.0, 0, -40], [-30, 0, -40], [30, 0, -40]])
    tri = np.array([[0, 1, 2], [0, 2, 3], [1, 4, 2]], np.int32)
    front = clip(v * [1, 1, 0.2] - [0, 0, 12])                      # everything in front of the camera
    assert (front[:, 3] > 0).all()
    pos = np.stack([front] * 70)
    pos[66] = clip(v)
    ref = R.rasterize(pos[[0, 66]], tri, (H, W))
    out, _ = dr.rasterize(dr.RasterizeCudaContext(), torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), resolution=[H, W], grad_db=False)
    out = out.cpu().numpy()
    assert (ref[1, ..., 3] > 0).sum() > 100 and not np.array_equal(ref[0, ..., 3], ref[1, ..., 3])
    assert np.array_equal(out[66, ..., 3], ref[1, ..., 3])
    for b in (0, 1, 63, 64, 65, 67, 69):
        assert np.array_equal(out[b, ..., 3], ref[0, ..., 3])


def test_dropped_triangle_counter():
    """dr.count_dropped_triangles (TSSPLAT_AMD_DR_CHECK=1 makes dr.rasterize warn with it): the (view, triangle) pairs this slice
    drops whole -- a vertex that is not finite or beyond the +-16384-pixel guard band (a finite vertex at w <= 0 is clipped)."""
    import torch
    import tssplat_amd.dr as dr
    pos = torch.tensor([[[0, 0, 0, 1.0], [1, 0, 0, 1], [0, 1, 0, -1.0], [0.5, 0.5, 0, 1], [float("nan"), 0, 0, 1], [4000.0, 0, 0, 1]],
                        [[0, 0, 0, 1.0], [1, 0, 0, 1], [0, 1, 0, 1.0], [0.5, 0.5, 0, 1], [0.0, 0, 0, 1], [0.0, 0, 0, 1]]])
    tri = torch.tensor([[0, 1, 2], [0, 1, 3], [0, 1, 4], [0, 1, 5]], dtype=torch.int32)
    # view 0: w <= 0 (clipped, not dropped), fine, NaN, 4000 * 0.5 * 64 = 128 000 px -> two dropped; view 1: nothing dropped
    assert dr.count_dropped_triangles(pos, tri, 64, 64) == 2
    assert dr.count_dropped_triangles(pos[1:], tri, 64, 64) == 0
