"""float64 / integer ORACLE for the renderer slice: ``rasterize`` and ``interpolate`` --
TEST INFRASTRUCTURE, never imported by the product path.

What it restates.  The reference renders through nvdiffrast (NVlabs/nvdiffrast, NOT vendored, version unpinned:
``import nvdiffrast.torch as dr`` at /root/reference/renderers/mesh_rasterizer.py:2) and uses exactly

    rast, _ = dr.rasterize(ctx, pos_clip[B,V,4], tri[T,3], resolution=[H,W], grad_db=False)     mesh_rasterizer.py:103
    out, _  = dr.interpolate(attr[1|B,V,C], rast, tri)                                          mesh_rasterizer.py:117,145,153

The algorithm below follows nvdiffrast's PUBLISHED description (Laine et al., "Modular Primitives for High-Performance
Differentiable Rendering", 2020, section 3.2-3.3 and the library documentation): clip-space input, OpenGL conventions
(NDC = xyz / w, pixel (i, j) has its centre at NDC ((i + 0.5) / W * 2 - 1, (j + 0.5) / H * 2 - 1), row 0 is the BOTTOM
row), no face culling, nearest depth wins, output ``(u, v, z/w, triangle_id + 1)`` with ``(u, v)`` the perspective-correct
barycentrics of the triangle's first two vertices and 0 in all four channels for background, and
``interpolate = u a0 + v a1 + (1 - u - v) a2``.

PARITY UNPINNED.  nvdiffrast is not in this image and the reference holds no golden images, so nothing here is checked
against the library itself.  Where the published description leaves a choice, this file fixes one and says so:

* coverage: vertices are snapped to 1/256 pixel (round half up) and the three edge functions are evaluated EXACTLY in
  integers at the pixel centre; a pixel on an edge belongs to the triangle whose edge is "top-left" in the y-up window
  (after orienting the triangle counter-clockwise: dy < 0, or dy == 0 and dx < 0), so triangles sharing an edge cover
  every pixel exactly once (nvdiffrast's CUDA rasteriser snaps to 1/16 pixel; the rule is the same kind);
* no polygon clipping: a triangle with a vertex at w <= 0, a non-finite coordinate or a snapped coordinate beyond
  +-2^22 sub-pixel units is dropped; fragments with z/w outside [-1, 1] are dropped per pixel;
* depth test: z/w of the three vertices (rounded to float32) interpolated linearly in window space as a PLANE EQUATION
  ``z/w = z0 + zx (x - x0) + zy (y - y0)`` over the snapped coordinates -- a rasteriser's usual depth set-up: two divisions
  per triangle, two multiplies and two adds per fragment -- in FLOAT32 with one fixed operation order, every operation
  rounded on its own (no fused multiply-add), scaled to a 32-bit key; equal keys are resolved towards the LOWER triangle
  index.  The GPU kernel performs the same float32 operations in the same order (IEEE basic operations are correctly
  rounded on both sides), which is what makes the triangle ids BIT-EXACT between the two.  (Rounds 3a used float64
  edge-function weights; float32 is what a hardware or CUDA rasteriser computes depth in and costs a quarter.)
* ``(u, v, z/w)`` are then recomputed from the unsnapped clip-space positions (homogeneous 2-D edge functions, as
  nvdiffrast's fragment stage does), clamped to [0, 1] / [-1, 1]: float64 here, float32 on the GPU -- compared with a
  tolerance.
"""
from __future__ import annotations

import numpy as np

SUBPIXEL_BITS = 8
SUB = 1 << SUBPIXEL_BITS
COORD_LIMIT = 1 << 22          # snapped coordinates beyond +-2^22 sub-pixel units drop the triangle
DEPTH_SCALE = 2147483648.0     # depth key = floor((z/w + 1) * 2^31), clamped to [0, 2^32 - 1]
NO_FRAGMENT = np.uint64(0xFFFFFFFFFFFFFFFF)


def snap_vertices(pos_clip: np.ndarray, height: int, width: int):
    """Window coordinates in 1/256 pixel (int64), NDC depth (float32) and a validity mask per vertex of one view.

    Every operation is a single correctly rounded float64 operation, in this order -- the GPU kernel repeats them."""
    p = np.asarray(pos_clip, dtype=np.float32).astype(np.float64)
    x, y, z, w = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    ok = np.isfinite(p).all(axis=1) & (w > 0.0)
    ws = np.where(ok, w, 1.0)
    xs = ((x / ws) * 0.5 + 0.5) * float(width)
    ys = ((y / ws) * 0.5 + 0.5) * float(height)
    X = np.floor(xs * float(SUB) + 0.5)
    Y = np.floor(ys * float(SUB) + 0.5)
    ok &= (np.abs(X) <= COORD_LIMIT) & (np.abs(Y) <= COORD_LIMIT)
    X = np.where(ok, X, 0.0).astype(np.int64)
    Y = np.where(ok, Y, 0.0).astype(np.int64)
    zw = np.where(ok, z / ws, 0.0).astype(np.float32)
    return X, Y, zw, ok


def _top_left(dx: int, dy: int) -> bool:
    return dy < 0 or (dy == 0 and dx < 0)


def _snap_point(x, y, z, w, height: int, width: int):
    """``snap_vertices`` for one float64 clip-space point (a vertex made by the near-plane clip); None when it cannot be snapped."""
    if not (np.isfinite(x) and np.isfinite(y) and np.isfinite(z) and np.isfinite(w) and w > 0.0):
        return None
    xs = ((x / w) * 0.5 + 0.5) * float(width)
    ys = ((y / w) * 0.5 + 0.5) * float(height)
    X = np.floor(xs * float(SUB) + 0.5)
    Y = np.floor(ys * float(SUB) + 0.5)
    if not (abs(X) <= COORD_LIMIT and abs(Y) <= COORD_LIMIT):
        return None
    return int(X), int(Y), np.float32(z / w)


def clip_near(p3: np.ndarray):
    """Sutherland-Hodgman clip of ONE triangle (``p3[3, 4]`` float64 clip-space rows) against the near plane ``z + w >= 0``.

    Returns the polygon as a list of entries ``("v", k)`` (original vertex k, kept) or ``("i", point[4])`` (an intersection, always
    computed FROM the inside vertex TOWARDS the outside one -- ``a + (d_a / (d_a - d_b)) * (b - a)``, one rounding per operation --
    so that the two triangles sharing a clipped edge make the same point)."""
    d = p3[:, 2] + p3[:, 3]
    inside = d >= 0.0
    poly = []
    for k in range(3):
        n = (k + 1) % 3
        if inside[k]:
            poly.append(("v", k))
        if inside[k] != inside[n]:
            a, b = (k, n) if inside[k] else (n, k)
            t = d[a] / (d[a] - d[b])
            poly.append(("i", p3[a] + t * (p3[b] - p3[a])))
    return poly


def _cover(key, t, v0, v1, v2, height, width):
    """Coverage + depth of ONE snapped triangle ``((x, y, zw) x 3)`` into the key image, under triangle id ``t``."""
    (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = v0, v1, v2
    area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)
    if area == 0:
        return
    if area < 0:                                    # orient counter-clockwise: swap vertices 1 and 2
        x1, y1, z1, x2, y2, z2 = x2, y2, z2, x1, y1, z1
    # pixel range whose centres can lie inside (centre of pixel i = i * 256 + 128 in sub-pixel units)
    px0 = max(0, (min(x0, x1, x2) - SUB // 2 + SUB - 1) // SUB)
    px1 = min(width - 1, (max(x0, x1, x2) - SUB // 2) // SUB)
    py0 = max(0, (min(y0, y1, y2) - SUB // 2 + SUB - 1) // SUB)
    py1 = min(height - 1, (max(y0, y1, y2) - SUB // 2) // SUB)
    if px0 > px1 or py0 > py1:
        return
    cx = (np.arange(px0, px1 + 1, dtype=np.int64) * SUB + SUB // 2)[None, :]
    cy = (np.arange(py0, py1 + 1, dtype=np.int64) * SUB + SUB // 2)[:, None]
    # E_k = edge function of the edge OPPOSITE vertex k (weights of vertex k), >= 0 inside
    e0 = (x2 - x1) * (cy - y1) - (y2 - y1) * (cx - x1)
    e1 = (x0 - x2) * (cy - y2) - (y0 - y2) * (cx - x2)
    e2 = (x1 - x0) * (cy - y0) - (y1 - y0) * (cx - x0)
    inside = np.ones(e0.shape, dtype=bool)
    for e, (dx, dy) in ((e0, (x2 - x1, y2 - y1)), (e1, (x0 - x2, y0 - y2)), (e2, (x1 - x0, y1 - y0))):
        inside &= (e > 0) | ((e == 0) & _top_left(dx, dy))
    if not inside.any():
        return
    # depth plane through the three snapped vertices (area > 0 after the orientation): float32, one rounding per operation
    f32 = np.float32
    A = f32(np.float64(abs(area)))
    d1, d2 = z1 - z0, z2 - z0
    zx = (d1 * f32(y2 - y0) - d2 * f32(y1 - y0)) / A
    zy = (d2 * f32(x1 - x0) - d1 * f32(x2 - x0)) / A
    zw = (z0 + zx * (cx - x0).astype(f32)) + zy * (cy - y0).astype(f32)
    assert zw.dtype == np.float32
    inside &= (zw >= f32(-1.0)) & (zw <= f32(1.0))
    q = (zw + f32(1.0)) * f32(DEPTH_SCALE)                  # float32, in [0, 2^32]: the scaling is exact
    q = np.where(q >= f32(4294967296.0), np.uint64(0xFFFFFFFF), np.where(inside, q, f32(0)).astype(np.uint64))
    k = (q << np.uint64(32)) | np.uint64(t)
    sub = key[py0:py1 + 1, px0:px1 + 1]
    np.minimum(sub, np.where(inside, k, NO_FRAGMENT), out=sub)


def rasterize_ids(pos_clip: np.ndarray, tri: np.ndarray, height: int, width: int) -> np.ndarray:
    """Depth keys ``(depth32 << 32) | triangle`` per pixel of ONE view, ``NO_FRAGMENT`` where nothing covers the centre.

    A triangle whose three vertices are finite but not all in front of the camera (``w <= 0`` somewhere) is clipped against the
    near plane (``clip_near``), its polygon snapped like any vertex and rasterised as a fan under the triangle's own id (what
    nvdiffrast's triangle set-up does with its frustum clipper; ``resolve`` works from the unclipped vertices either way).  A
    triangle with a non-finite vertex, or one that cannot be snapped (beyond the +-16384-pixel guard band), is dropped."""
    X, Y, ZW, ok = snap_vertices(pos_clip, height, width)
    p64 = np.asarray(pos_clip, dtype=np.float32).astype(np.float64)
    finite = np.isfinite(p64).all(axis=1)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    key = np.full((height, width), NO_FRAGMENT, dtype=np.uint64)
    for t in range(tri.shape[0]):
        i = tri[t]
        if ok[i[0]] and ok[i[1]] and ok[i[2]]:
            _cover(key, t, *[(int(X[k]), int(Y[k]), ZW[k]) for k in i], height, width)
            continue
        behind = finite[i] & (p64[i, 3] <= 0.0)
        if not (ok[i] | behind).all():
            continue                                 # a vertex that is not finite, or beyond the guard band: dropped, not clipped
        pts = []
        for kind, what in clip_near(p64[i]):
            if kind == "v":
                v = i[what]
                pts.append((int(X[v]), int(Y[v]), ZW[v]) if ok[v] else None)     # (kept vertices use the shared snap)
            else:
                pts.append(_snap_point(what[0], what[1], what[2], what[3], height, width))
        if len(pts) < 3 or any(q is None for q in pts):
            continue
        for k in range(1, len(pts) - 1):
            _cover(key, t, pts[0], pts[k], pts[k + 1], height, width)
    return key


def resolve(pos_clip: np.ndarray, tri: np.ndarray, key: np.ndarray) -> np.ndarray:
    """``(u, v, z/w, id + 1)`` in float64 from the winning triangle of every pixel (nvdiffrast's fragment stage)."""
    height, width = key.shape
    p = np.asarray(pos_clip, dtype=np.float32).astype(np.float64)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    out = np.zeros((height, width, 4))
    hit = key != NO_FRAGMENT
    ids = (key & np.uint64(0xFFFFFFFF)).astype(np.int64)
    jj, ii = np.nonzero(hit)
    t = ids[jj, ii]
    fx = (ii + 0.5) / width * 2.0 - 1.0
    fy = (jj + 0.5) / height * 2.0 - 1.0
    v = p[tri[t]]                                          # [k, 3, 4]
    px = v[:, :, 0] - fx[:, None] * v[:, :, 3]
    py = v[:, :, 1] - fy[:, None] * v[:, :, 3]
    a0 = px[:, 1] * py[:, 2] - py[:, 1] * px[:, 2]
    a1 = px[:, 2] * py[:, 0] - py[:, 2] * px[:, 0]
    a2 = px[:, 0] * py[:, 1] - py[:, 0] * px[:, 1]
    s = a0 + a1 + a2
    s = np.where(s == 0.0, 1.0, s)
    b0, b1 = a0 / s, a1 / s
    b2 = 1.0 - b0 - b1
    z = b0 * v[:, 0, 2] + b1 * v[:, 1, 2] + b2 * v[:, 2, 2]
    w = b0 * v[:, 0, 3] + b1 * v[:, 1, 3] + b2 * v[:, 2, 3]
    out[jj, ii, 0] = np.clip(b0, 0.0, 1.0)
    out[jj, ii, 1] = np.clip(b1, 0.0, 1.0)
    out[jj, ii, 2] = np.clip(z / np.where(w == 0.0, 1.0, w), -1.0, 1.0)
    out[jj, ii, 3] = t + 1.0
    return out


def rasterize(pos_clip: np.ndarray, tri: np.ndarray, resolution) -> np.ndarray:
    """``dr.rasterize(ctx, pos, tri, resolution)[0]`` (mesh_rasterizer.py:103): ``[B, H, W, 4]`` float64."""
    pos_clip = np.asarray(pos_clip, dtype=np.float32)
    if pos_clip.ndim == 2:
        pos_clip = pos_clip[None]
    height, width = int(resolution[0]), int(resolution[1])
    return np.stack([resolve(pos_clip[b], tri, rasterize_ids(pos_clip[b], tri, height, width)) for b in range(pos_clip.shape[0])])


def _valid_ids(w: np.ndarray, tri: np.ndarray, n_vertices: int):
    """Triangle id of every pixel and the mask of pixels that name a usable triangle.  Background, an id beyond the triangle
    list (a ``rast`` image made with another list) and a triangle with a vertex index outside ``[0, n_vertices)`` are all treated
    as background -- zero output, zero gradient -- which is what nvdiffrast does with them."""
    w = np.where(np.isfinite(w), w, 0.0)
    ids = np.clip(w, 0.0, 2.0 ** 24).astype(np.int64) - 1
    hit = (ids >= 0) & (ids < len(tri))
    safe = np.where(hit, ids, 0)
    if len(tri):
        t = tri[safe]
        hit &= np.all((t >= 0) & (t < n_vertices), axis=-1)
    return ids, hit


def interpolate(attr: np.ndarray, rast: np.ndarray, tri: np.ndarray) -> np.ndarray:
    """``dr.interpolate(attr, rast, tri)[0]`` (mesh_rasterizer.py:117): ``u a0 + v a1 + (1 - u - v) a2``, 0 on background."""
    attr = np.asarray(attr, dtype=np.float64)
    rast = np.asarray(rast, dtype=np.float64)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    B = rast.shape[0]
    out = np.zeros(rast.shape[:3] + (attr.shape[-1],))
    for b in range(B):
        a = attr[b if attr.shape[0] > 1 else 0]
        ids, hit = _valid_ids(rast[b, ..., 3], tri, attr.shape[1])
        t = tri[ids[hit]]
        u, v = rast[b, ..., 0][hit][:, None], rast[b, ..., 1][hit][:, None]
        out[b][hit] = u * a[t[:, 0]] + v * a[t[:, 1]] + (1.0 - u - v) * a[t[:, 2]]
    return out


def interpolate_backward(attr: np.ndarray, rast: np.ndarray, tri: np.ndarray, grad_out: np.ndarray):
    """Gradients of ``interpolate`` w.r.t. ``attr`` (scatter of the barycentric weights) and w.r.t. ``rast``'s (u, v)."""
    attr = np.asarray(attr, dtype=np.float64)
    rast = np.asarray(rast, dtype=np.float64)
    g = np.asarray(grad_out, dtype=np.float64)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    grad_attr = np.zeros_like(attr)
    grad_rast = np.zeros_like(rast)
    for b in range(rast.shape[0]):
        ab = b if attr.shape[0] > 1 else 0
        ids, hit = _valid_ids(rast[b, ..., 3], tri, attr.shape[1])
        t = tri[ids[hit]]
        u, v = rast[b, ..., 0][hit][:, None], rast[b, ..., 1][hit][:, None]
        gg = g[b][hit]
        np.add.at(grad_attr[ab], t[:, 0], u * gg)
        np.add.at(grad_attr[ab], t[:, 1], v * gg)
        np.add.at(grad_attr[ab], t[:, 2], (1.0 - u - v) * gg)
        a0, a1, a2 = attr[ab][t[:, 0]], attr[ab][t[:, 1]], attr[ab][t[:, 2]]
        du = np.sum(gg * (a0 - a2), axis=1)
        dv = np.sum(gg * (a1 - a2), axis=1)
        gr = grad_rast[b]
        gr[..., 0][hit] = du
        gr[..., 1][hit] = dv
    return grad_attr, grad_rast


def rasterize_backward(pos_clip: np.ndarray, tri: np.ndarray, rast: np.ndarray, grad_rast: np.ndarray) -> np.ndarray:
    """Gradient of ``rasterize`` w.r.t. the clip-space positions, from the gradient of its ``(u, v)`` outputs (the z/w and
    id channels carry none, as in nvdiffrast; the clamps of ``resolve`` are treated as inactive).  ``[B, V, 4]`` float64."""
    p = np.asarray(pos_clip, dtype=np.float32).astype(np.float64)
    if p.ndim == 2:
        p = p[None]
    rast = np.asarray(rast, dtype=np.float64)
    g = np.asarray(grad_rast, dtype=np.float64)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    B, height, width = rast.shape[:3]
    out = np.zeros_like(p)
    for b in range(B):
        ids = rast[b, ..., 3].astype(np.int64) - 1
        jj, ii = np.nonzero(ids >= 0)
        t = tri[ids[jj, ii]]
        fx = (ii + 0.5) / width * 2.0 - 1.0
        fy = (jj + 0.5) / height * 2.0 - 1.0
        v = p[b][t]
        px = v[:, :, 0] - fx[:, None] * v[:, :, 3]
        py = v[:, :, 1] - fy[:, None] * v[:, :, 3]
        a0 = px[:, 1] * py[:, 2] - py[:, 1] * px[:, 2]
        a1 = px[:, 2] * py[:, 0] - py[:, 2] * px[:, 0]
        a2 = px[:, 0] * py[:, 1] - py[:, 0] * px[:, 1]
        s = a0 + a1 + a2
        s = np.where(s == 0.0, 1.0, s)
        u, w_ = a0 / s, a1 / s
        gu, gv = g[b, jj, ii, 0], g[b, jj, ii, 1]
        dot = gu * u + gv * w_
        da0, da1, da2 = (gu - dot) / s, (gv - dot) / s, -dot / s
        dpx = np.stack([da1 * -py[:, 2] + da2 * py[:, 1], da0 * py[:, 2] + da2 * -py[:, 0], da0 * -py[:, 1] + da1 * py[:, 0]], axis=1)
        dpy = np.stack([da1 * px[:, 2] + da2 * -px[:, 1], da0 * -px[:, 2] + da2 * px[:, 0], da0 * px[:, 1] + da1 * -px[:, 0]], axis=1)
        for k in range(3):
            np.add.at(out[b, :, 0], t[:, k], dpx[:, k])
            np.add.at(out[b, :, 1], t[:, k], dpy[:, k])
            np.add.at(out[b, :, 3], t[:, k], -fx * dpx[:, k] - fy * dpy[:, k])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# antialias  (mesh_rasterizer.py:107,128: dr.antialias(color, rast, pos_clip, tri, topology_hash=None, pos_gradient_boost=1.0))
#
# Published algorithm (Laine et al. 2020, section 3.5): every pair of horizontally / vertically adjacent pixels with different
# triangle ids is a potential discontinuity.  Take the triangle of the surface CLOSER to the camera (z/w of `rast`; the
# other pixel may be background), look at its edges: an edge is a SILHOUETTE edge if no other triangle shares it or if the
# triangle that does lies on the same side of it in the image.  If a silhouette edge passes between the two pixel centres
# -- horizontal pairs look at edges that are more vertical than horizontal, and vice versa -- the crossing point, at distance
# t in [0, 1] from the centre of the closer triangle's pixel P towards the other pixel Q, estimates how far the closer
# surface reaches: alpha = t - 1/2 > 0 blends alpha of P's colour into Q, alpha < 0 blends -alpha of Q's colour into P
# (zero at the midpoint, one half at a pixel centre).  The position gradient is the gradient of alpha.
#
# Choices fixed here (PARITY UNPINNED, as above): pairs (p, p + x) and (p, p + y) of every pixel p; ties in z/w go to the
# second pixel's triangle; window coordinates from the UNSNAPPED clip-space positions; a pair whose triangle has a vertex
# at w <= 0 is skipped; the shared-edge partner of (triangle, edge) is the lowest-numbered other (triangle, edge) on the
# same undirected vertex pair; an edge whose partner's far vertex has w <= 0 counts as a silhouette; every qualifying
# edge of the triangle contributes (normally one).  All decisions are single float64 operations in a fixed order, which
# the GPU kernel repeats -- the SET of blends is identical, their float32 sums are compared with a tolerance.
# ---------------------------------------------------------------------------------------------------------------------
def edge_partners(tri: np.ndarray) -> np.ndarray:
    """``opp[3 t + e]`` = the vertex opposite edge ``e`` (the edge not touching local vertex ``e``) in the partner triangle
    of triangle ``t`` across that edge, -1 on a boundary edge -- what nvdiffrast's topology hash answers."""
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    T = tri.shape[0]
    table = {}
    for t in range(T):
        for e in range(3):
            a, b = int(tri[t, (e + 1) % 3]), int(tri[t, (e + 2) % 3])
            table.setdefault((min(a, b), max(a, b)), []).append(3 * t + e)
    opp = np.full(3 * T, -1, dtype=np.int32)
    for ids in table.values():
        ids.sort()
        for i in ids:
            others = [j for j in ids if j != i]
            if others:
                j = others[0]
                opp[i] = tri[j // 3, j % 3]
    return opp


def _window(p4, height, width):
    """Window coordinates in pixels (y up) of one clip-space vertex, or None when it cannot be projected."""
    x, y, _, w = (float(c) for c in p4)
    if not (np.isfinite(x) and np.isfinite(y) and np.isfinite(w)) or not w > 0.0:
        return None
    return ((x / w) * 0.5 + 0.5) * float(width), ((y / w) * 0.5 + 0.5) * float(height)


def _antialias_events(rast_b, pos_b, tri, opp, pixel_gradient=False):
    """The blends of one view: tuples ``(dst (j, i), src (j, i), weight, sign, axis, (va, vb), (dA, dB))`` where ``dA / dB``
    are the derivatives of ``t`` w.r.t. the window coordinates of the edge's two vertices."""
    height, width = rast_b.shape[:2]
    ids = rast_b[..., 3].astype(np.int64) - 1
    zw = rast_b[..., 2]
    events = []
    for axis, (dj, di) in enumerate(((0, 1), (1, 0))):      # axis 0: horizontal pair, axis 1: vertical pair
        t0 = ids[:height - dj, :width - di]
        t1 = ids[dj:, di:]
        jj, ii = np.nonzero(t0 != t1)
        for j, i in zip(jj.tolist(), ii.tolist()):
            a0, a1 = int(ids[j, i]), int(ids[j + dj, i + di])
            if a0 >= 0 and a1 >= 0:
                first = float(zw[j, i]) < float(zw[j + dj, i + di])
            else:
                first = a0 >= 0
            t = a0 if first else a1
            P = (j, i) if first else (j + dj, i + di)
            Q = (j + dj, i + di) if first else (j, i)
            vid = [int(k) for k in tri[t]]
            win = [_window(pos_b[k], height, width) for k in vid]
            if any(q is None for q in win):
                continue
            cx, cy = P[1] + 0.5, P[0] + 0.5
            step = float((Q[1] - P[1]) if axis == 0 else (Q[0] - P[0]))       # +-1 along the pair's axis
            for e in range(3):
                ka, kb = (e + 1) % 3, (e + 2) % 3
                (Ax, Ay), (Bx, By), (Ox, Oy) = win[ka], win[kb], win[e]
                ex, ey = Bx - Ax, By - Ay
                o2 = int(opp[3 * t + e])
                if o2 >= 0:
                    w2 = _window(pos_b[o2], height, width)
                    if w2 is not None:
                        s1 = ex * (Oy - Ay) - ey * (Ox - Ax)
                        s2 = ex * (w2[1] - Ay) - ey * (w2[0] - Ax)
                        if (s1 > 0.0) != (s2 > 0.0):
                            continue                      # the partner continues the surface on the other side: not a silhouette
                if axis == 0:
                    if not abs(ey) >= abs(ex):
                        continue
                    sA, sB, base, span, centre = Ay - cy, By - cy, Ax, ex, cx
                else:
                    if not abs(ex) >= abs(ey):
                        continue
                    sA, sB, base, span, centre = Ax - cx, Bx - cx, Ay, ey, cy
                if (sA > 0.0) == (sB > 0.0):
                    continue
                den = sA - sB
                lam = sA / den
                tt = ((base + span * lam) - centre) * step
                if not (tt >= 0.0 and tt <= 1.0):
                    continue
                alpha = tt - 0.5
                if alpha == 0.0:
                    continue
                # d tt / d (window coordinates of A and B): along the pair's axis (1 - lam, lam), across it span * d lam
                dl_a, dl_b = -sB / (den * den), sA / (den * den)
                along = (step * (1.0 - lam), step * lam)
                across = (step * span * dl_a, step * span * dl_b)
                dA = (along[0], across[0]) if axis == 0 else (across[0], along[0])      # (d/dx, d/dy) of vertex A
                dB = (along[1], across[1]) if axis == 0 else (across[1], along[1])
                if alpha > 0.0:
                    events.append((Q, P, alpha, 1.0, axis, (vid[ka], vid[kb]), (dA, dB)))
                else:
                    events.append((P, Q, -alpha, -1.0, axis, (vid[ka], vid[kb]), (dA, dB)))
    return events


def antialias(color: np.ndarray, rast: np.ndarray, pos_clip: np.ndarray, tri: np.ndarray, opp: np.ndarray | None = None) -> np.ndarray:
    """``dr.antialias(color, rast, pos, tri)`` (mesh_rasterizer.py:107,128): ``[B, H, W, C]`` float64."""
    color = np.asarray(color, dtype=np.float32).astype(np.float64)
    rast = np.asarray(rast, dtype=np.float32).astype(np.float64)
    p = np.asarray(pos_clip, dtype=np.float32).astype(np.float64)
    if p.ndim == 2:
        p = p[None]
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    opp = edge_partners(tri) if opp is None else opp
    out = color.copy()
    for b in range(rast.shape[0]):
        for dst, src, wgt, _, _, _, _ in _antialias_events(rast[b], p[b], tri, opp):
            out[b][dst] += wgt * (color[b][src] - color[b][dst])
    return out


def antialias_backward(color, rast, pos_clip, tri, grad_out, opp=None, pos_gradient_boost: float = 1.0):
    """Gradients of ``antialias`` w.r.t. ``color`` and ``pos_clip`` (float64)."""
    color = np.asarray(color, dtype=np.float32).astype(np.float64)
    rast = np.asarray(rast, dtype=np.float32).astype(np.float64)
    p = np.asarray(pos_clip, dtype=np.float32).astype(np.float64)
    if p.ndim == 2:
        p = p[None]
    g = np.asarray(grad_out, dtype=np.float64)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    opp = edge_partners(tri) if opp is None else opp
    height, width = rast.shape[1:3]
    grad_color = g.copy()
    grad_pos = np.zeros_like(p)
    for b in range(rast.shape[0]):
        for dst, src, wgt, sign, _, (va, vb), (dA, dB) in _antialias_events(rast[b], p[b], tri, opp):
            gd = g[b][dst]
            grad_color[b][src] += wgt * gd
            grad_color[b][dst] -= wgt * gd
            dt = sign * float(np.dot(gd, color[b][src] - color[b][dst])) * pos_gradient_boost
            for vtx, (ddx, ddy) in ((va, dA), (vb, dB)):
                x, y, _, w = p[b, vtx]
                gx, gy = dt * ddx, dt * ddy                               # d L / d window (x, y) of the vertex
                grad_pos[b, vtx, 0] += gx * (0.5 * width / w)
                grad_pos[b, vtx, 1] += gy * (0.5 * height / w)
                grad_pos[b, vtx, 3] += gx * (-0.5 * width * x / (w * w)) + gy * (-0.5 * height * y / (w * w))
    return grad_color, grad_pos


def orbit_mvps(n_views: int, distance: float = 3.0, fov_deg: float = 40.0, near: float = 0.5, far: float = 8.0,
               elevation_deg: float = 20.0) -> np.ndarray:
    """``n_views`` model-view-projection matrices (float32 ``[n, 4, 4]``, OpenGL clip space) on a circle around the
    origin -- the kind of batch /root/reference/data/*.py hands to ``MeshRasterizer.forward(mvp, ...)``."""
    f = 1.0 / np.tan(np.radians(fov_deg) / 2.0)
    proj = np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, (far + near) / (near - far), 2 * far * near / (near - far)], [0, 0, -1, 0]])
    out = []
    el = np.radians(elevation_deg)
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


def transform_pos(mvp: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """``MeshRasterizer.transform_pos`` (mesh_rasterizer.py:54-78, non-ortho branch): ``[v, 1] @ mvp^T`` per view, float32."""
    posw = np.concatenate([np.asarray(pos, dtype=np.float32), np.ones((pos.shape[0], 1), dtype=np.float32)], axis=1)
    return np.matmul(posw[None], np.transpose(np.asarray(mvp, dtype=np.float32), (0, 2, 1))).astype(np.float32)
