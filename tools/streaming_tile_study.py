#!/usr/bin/env python3
"""Offline study: would a STREAMING tile -- a workgroup sweeping a tube of tets level by level with a
rolling LDS window, instead of a blob with a one-ring halo -- need fewer slots per tet than the blob tiling?

Model.  Neighbours of a tet in sweep level l lie in levels l-1, l, l+1 (levels = BFS distance inside the tube, or slabs
along the sweep axis; both are reported).  While level l is in pass 1 (F), level l-1 can do pass 2 (needs F of l-2..l)
and level l-2 pass 3 (needs H of l-3..l-1), after which the forces of levels l-3, l-2 are gathered per vertex.  Live at
once: F of 3 levels, H of 3, forces of 2 -- with in-place reuse of dead records at least 6 level-records of 48 B.  The
sweep direction needs no halo (the tube's ends are the sphere's boundary); the tube's SIDES do.  So for a tube with
per-level cross-section A (+ side halo h(A)):  LDS  >=  6 * 48 B * max_l (A_l + h_l),  slots/tet = 1 + sum h / sum A.

The script cuts every sphere into K x K tubes by recursive coordinate bisection of the two axes orthogonal to the sweep
axis, levels them, and reports the smallest K whose window fits 80 KiB (two workgroups per CU) and 160 KiB (one), next
to the blob tiling the library builds for the same mesh.

    python tools/streaming_tile_study.py            # kuhn19 and a.veg
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tssplat_amd import scenes, tet_spheres_ext as T

RECORD = 48
LIVE_LEVELS = 6


def face_adjacency(tets):
    """nbr[e, k] = the tet across the face opposite local vertex k, or -1 (sort of the 4 m sorted vertex triples)."""
    m = tets.shape[0]
    faces = np.stack([np.sort(np.delete(tets, k, axis=1), axis=1) for k in range(4)], axis=1).reshape(-1, 3).astype(np.int64)
    key = (faces[:, 0] << 42) | (faces[:, 1] << 21) | faces[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    same = ks[1:] == ks[:-1]
    nbr = np.full(4 * m, -1, dtype=np.int64)
    a, b = order[:-1][same], order[1:][same]
    nbr[a], nbr[b] = b // 4, a // 4
    return nbr.reshape(m, 4)


def tubes(cen, k, axes):
    """K x K tubes: bisect along axes[0] into k equal-count strips, each of those along axes[1] into k."""
    part = np.zeros(cen.shape[0], dtype=np.int64)
    order0 = np.argsort(cen[:, axes[0]], kind="stable")
    for i, chunk in enumerate(np.array_split(order0, k)):
        order1 = chunk[np.argsort(cen[chunk, axes[1]], kind="stable")]
        for j, c2 in enumerate(np.array_split(order1, k)):
            part[c2] = i * k + j
    return part


def study(name, rest, tets):
    m = tets.shape[0]
    nbr = face_adjacency(tets)
    cen = rest[tets].mean(axis=1)
    ext = cen.max(axis=0) - cen.min(axis=0)
    sweep = int(np.argmax(ext))
    axes = [a for a in range(3) if a != sweep]
    ts = T.TetSpheres(rest.reshape(-1), tets.reshape(-1), host_only=True)
    info = ts.plan_info()
    print(f"== {name}: {m} tets; blob tiling of the library: {info['n_tiles']} tiles, slots/tet {info['total_slots'] / m:.4f}")
    # slab thickness: one 'level' must contain all face neighbours within +-1 -> use BFS levels inside each tube
    for budget, label in ((80 * 1024, "80 KiB (2 workgroups / CU)"), (160 * 1024, "160 KiB (1 workgroup / CU)")):
        cap = budget // (LIVE_LEVELS * RECORD)
        found = None
        for k in range(1, 13):
            part = tubes(cen, k, axes)
            worst, halo_total = 0, 0
            for p in range(k * k):
                own = np.nonzero(part == p)[0]
                if own.size == 0:
                    continue
                inside = np.zeros(m, dtype=bool)
                inside[own] = True
                nb = nbr[own]
                halo = np.unique(nb[(nb >= 0) & ~inside[np.clip(nb, 0, m - 1)]])
                halo_total += halo.size
                # BFS levels over owned + halo from the tets with the smallest sweep coordinate
                members = np.concatenate([own, halo])
                local = -np.ones(m, dtype=np.int64)
                local[members] = np.arange(members.size)
                level = -np.ones(members.size, dtype=np.int64)
                start = members[cen[members, sweep] <= np.quantile(cen[members, sweep], 0.02)]
                frontier = local[start]
                level[frontier] = 0
                l = 0
                while frontier.size:
                    nn = nbr[members[frontier]].ravel()
                    nn = nn[nn >= 0]
                    nn = local[nn]
                    nn = np.unique(nn[nn >= 0])
                    nn = nn[level[nn] < 0]
                    l += 1
                    level[nn] = l
                    frontier = nn
                level[level < 0] = l                            # (disconnected leftovers of a tube: counted in the last level)
                widths = np.bincount(level)
                worst = max(worst, int(widths.max()))
            spt = 1.0 + halo_total / m
            fits = worst <= cap
            print(f"   {label}: K = {k:2d} ({k * k:3d} tubes): widest level {worst:5d} slots (capacity {cap}), slots/tet {spt:.4f}" + ("  <- fits" if fits else ""))
            if fits:
                found = (k, spt, worst)
                break
        if found is None:
            print(f"   {label}: no K <= 12 fits")
    return info["total_slots"] / m


def main():
    v, t = scenes.kuhn_ball(19)
    study("kuhn_ball(19)", v.astype(np.float32), t.astype(np.int32))
    aveg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "aveg_mesh.npz")
    if os.path.exists(aveg):
        g = np.load(aveg)
        study("a.veg", g["rest"], g["tets"])


if __name__ == "__main__":
    main()
