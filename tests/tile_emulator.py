"""float64 replay of the tile planes the HIP kernels consume -- a TEST helper.

Walks the host copy of a tiling plan (through the C ABI's introspection calls)
and performs, tile by tile, exactly the passes of
tssplat_amd/csrc/kernels.hip::tile_energy_kernel in numpy float64 -- the
scatter of every slot's four corner forces into the tile's per-vertex force
array (row table + vertex + rank) and the row-by-row sums included -- followed
by the finish kernel's staging sum.  If this matches the oracle, the plan data
(local vertices and ranks, halo, owned order, record tokens, Dm^-1 planes, row
table, destinations, finish lists) is right, and a GPU mismatch can only come
from the kernel code itself.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from tssplat_amd import _capi

ROW_TABLE_ENTRIES = 72
MAX_RANK = 64


def plan_tiles(ts):
    lib = _capi.load()
    info = ts.plan_info()
    for t in range(info["n_tiles"]):
        tv = _capi.TileView()
        _capi.check(lib.tsamd_get_tile(ts._handle(), t, C.byref(tv)))
        sp = tv.s_pad
        planes = np.ctypeslib.as_array(tv.planes, shape=(info["n_planes"], sp)).copy()
        row_start = np.ctypeslib.as_array(tv.row_start, shape=(ROW_TABLE_ENTRIES,)).copy()
        gvid = np.ctypeslib.as_array(tv.gvid, shape=(tv.n_verts,)).copy()
        vdst = np.ctypeslib.as_array(tv.vdst, shape=(tv.n_verts,)).copy()
        slot_tet = np.ctypeslib.as_array(tv.slot_tet, shape=(sp,)).copy()
        rest = np.ctypeslib.as_array(tv.rest, shape=(tv.n_verts, 4)).copy() if bool(tv.rest) else None
        yield dict(n_slots=tv.n_slots, n_owned=tv.n_owned, s_pad=sp, n_verts=tv.n_verts, n_excl=tv.n_excl,
                   stage_off=tv.stage_off, n_rows=tv.n_rows, rec_base=tv.rec_base, planes=planes, gvid=gvid, vdst=vdst,
                   slot_tet=slot_tet, row_start=row_start, rest=rest)


def finish_lists(ts):
    lib = _capi.load()
    n = C.c_int64()
    vid = C.POINTER(C.c_int32)()
    off = C.POINTER(C.c_int32)()
    idx = C.POINTER(C.c_int32)()
    _capi.check(lib.tsamd_get_finish_lists(ts._handle(), C.byref(n), C.byref(vid), C.byref(off), C.byref(idx)))
    k = n.value
    if k == 0:
        return np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32)
    off_a = np.ctypeslib.as_array(off, shape=(k + 1,)).copy()
    ne = int(off_a[-1])
    idx_a = np.ctypeslib.as_array(idx, shape=(ne,)).copy() if ne else np.zeros(0, np.int32)
    return np.ctypeslib.as_array(vid, shape=(k,)).copy(), off_a, idx_a


def adjacency(ts):
    lib = _capi.load()
    p = C.POINTER(C.c_int32)()
    _capi.check(lib.tsamd_get_adjacency(ts._handle(), C.byref(p)))
    if ts.nele == 0:
        return np.zeros((0, 4), np.int32)
    return np.ctypeslib.as_array(p, shape=(ts.nele, 4)).copy()


def _det(F):
    return (-F[:, 0, 2] * F[:, 1, 1] * F[:, 2, 0] + F[:, 0, 1] * F[:, 1, 2] * F[:, 2, 0]
            + F[:, 0, 2] * F[:, 1, 0] * F[:, 2, 1] - F[:, 0, 0] * F[:, 1, 2] * F[:, 2, 1]
            - F[:, 0, 1] * F[:, 1, 0] * F[:, 2, 2] + F[:, 0, 0] * F[:, 1, 1] * F[:, 2, 2])


def _cof(F):
    C_ = np.empty_like(F)
    C_[:, 0, 0] = F[:, 1, 1] * F[:, 2, 2] - F[:, 1, 2] * F[:, 2, 1]
    C_[:, 0, 1] = F[:, 1, 2] * F[:, 2, 0] - F[:, 1, 0] * F[:, 2, 2]
    C_[:, 0, 2] = F[:, 1, 0] * F[:, 2, 1] - F[:, 1, 1] * F[:, 2, 0]
    C_[:, 1, 0] = F[:, 0, 2] * F[:, 2, 1] - F[:, 0, 1] * F[:, 2, 2]
    C_[:, 1, 1] = F[:, 0, 0] * F[:, 2, 2] - F[:, 0, 2] * F[:, 2, 0]
    C_[:, 1, 2] = F[:, 0, 1] * F[:, 2, 0] - F[:, 0, 0] * F[:, 2, 1]
    C_[:, 2, 0] = F[:, 0, 1] * F[:, 1, 2] - F[:, 0, 2] * F[:, 1, 1]
    C_[:, 2, 1] = F[:, 0, 2] * F[:, 1, 0] - F[:, 0, 0] * F[:, 1, 2]
    C_[:, 2, 2] = F[:, 0, 0] * F[:, 1, 1] - F[:, 0, 1] * F[:, 1, 0]
    return C_


def emulate(ts, x, c1, c2, order, grad_output=1.0):
    """Returns (E, E_s, E_b, grad[n,3]) in float64 from the plan's planes."""
    c1 = float(np.float32(c1))
    c2 = float(np.float32(c2))
    x = np.asarray(x, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    n = x.shape[0]
    grad = np.full((n, 3), np.nan)
    info = ts.plan_info()
    stage = np.full((max(info["shared_vertex_copies"], 1), 3), np.nan)
    Es = Eb = 0.0
    vid, off, sdst = finish_lists(ts)
    for T in plan_tiles(ts):
        sp, pl = T["s_pad"], T["planes"]
        spt = info["slots_per_thread"]
        nq = sp // spt
        slots = np.arange(sp)
        perm = (slots % spt) * nq + slots // spt            # LDS record index of every slot = its item number in the tile
        f16 = np.stack([pl[0] & 0xffff, pl[0] >> 16, pl[1] & 0xffff, pl[1] >> 16], axis=1).astype(np.int64)
        lv, rank = f16 & 0x3ff, f16 >> 10
        real = T["slot_tet"] >= 0
        assert real.sum() == T["n_slots"] and np.all(f16[~real] == 0)
        assert lv[real].max() < T["n_verts"] <= 1023
        owned = perm < T["n_owned"]                        # (no owned bit: the items below n_owned are the owned ones)
        assert np.all(real[owned]) and np.all(perm[real] < T["n_slots"])
        # neighbour fields are record tokens rec_base / 4 + 12 * idx + rot (plan.h: record_token); a face without a usable
        # neighbour points at the slot's own record
        RB = T["rec_base"]
        assert RB % 16 == 0 and RB == 320 + (32 if T["rest"] is not None else 16) * ((T["n_verts"] + 3) & ~3) + 256
        tok = np.stack([pl[2] & 0xffff, pl[2] >> 16, pl[3] & 0xffff, pl[3] >> 16], axis=1).astype(np.int64) - RB // 4
        nb = tok // 12
        assert np.all(tok >= 0) and np.array_equal(tok % 12, (nb >> 3) & 3), "token = rec_base / 4 + 12 * idx + ((idx >> 3) & 3)"
        if T["rest"] is None:
            dminv = pl[4:13].view(np.float32).astype(np.float64).T.reshape(sp, 3, 3)
        else:                                    # rebuild_dminv plan: Dm^-1 from the tile's rest positions (exact here)
            assert pl.shape[0] == 4 and np.all(T["rest"][:, 3] == 0)
            R = T["rest"][:, :3].astype(np.float64)[lv]                      # [sp,4,3]
            Dm = np.stack([R[:, 1] - R[:, 0], R[:, 2] - R[:, 0], R[:, 3] - R[:, 0]], axis=2)
            real = T["slot_tet"] >= 0
            dminv = np.zeros((sp, 3, 3))
            dminv[real] = np.linalg.inv(Dm[real])
        assert owned.sum() == T["n_owned"]
        ZS = sp
        xs = x[T["gvid"]]
        p = xs[lv]                                            # [sp,4,3]
        Ds = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)
        F = Ds @ dminv
        J = _det(F)
        Jm = np.maximum(-J, 0)
        if order == 2:
            pen, dpen = Jm ** 2, -2 * Jm
        elif order == 4:
            pen, dpen = Jm ** 4, -4 * Jm ** 3
        else:
            pen, dpen = 0 * Jm, 0 * Jm
        Eb += float(pen[owned].sum())
        scal = np.where(owned, c2 * dpen, 0.0)
        Fz = np.zeros((sp + 1, 9))
        Fz[perm] = F.reshape(sp, 9)
        self_idx = perm                        # LDS record of every slot
        missing = nb == self_idx[:, None]
        assert nb.max() < sp, "no neighbour field may point at the zero slot any more"
        if pl.shape[0] in (13, 4):            # uniform umbrella: 4 * own - sum of the four (a missing face reads own)
            wd = 4.0 * np.ones(sp)
            wr = wc = -np.ones((sp, 4))
        else:                                 # explicit element operator: L[e,e], L[e,n_k], L[n_k,e]
            wd = pl[13].view(np.float32).astype(np.float64)
            wr = pl[14:18].view(np.float32).astype(np.float64).T
            # (a symmetric operator stores its weights once: 18 planes, pass 3 applies the row weights -- plan.h)
            wc = pl[18:22].view(np.float32).astype(np.float64).T if pl.shape[0] == 22 else wr
            assert np.all(wr[missing] == 0) and np.all(wc[missing] == 0)
        H = wd[:, None] * F.reshape(sp, 9) + (wr[:, :, None] * Fz[nb]).sum(axis=1)
        H[~owned] = 0.0
        Es += 0.5 * float((H * H).sum())
        Hz = np.zeros((sp + 1, 9))
        Hz[perm] = H
        Q = wd[:, None] * H + (wc[:, :, None] * Hz[nb]).sum(axis=1)
        P = c1 * Q.reshape(sp, 3, 3) + scal[:, None, None] * _cof(F)
        d = P @ np.transpose(dminv, (0, 2, 1))
        # the force array: entry (v, r) at row_start[r] + v, filled by the slots' scatter, summed row by row per vertex --
        # exactly the kernel's last two phases
        contrib = np.concatenate([-d.sum(axis=2)[:, None, :], np.transpose(d, (0, 2, 1))], axis=1)   # [slot, corner a, xyz]: f0 = -(f1 + f2 + f3)
        rs = T["row_start"].astype(np.int64)
        nv, nrows = T["n_verts"], T["n_rows"]
        assert rs[0] == 0 and np.all(np.diff(rs) >= 0) and nrows <= MAX_RANK
        width = np.diff(rs)
        assert np.all(width[nrows:] == 0) and (nrows == 0 or width[nrows - 1] > 0) and rs[-1] == 4 * T["n_slots"]
        assert np.all(np.diff(width[:nrows]) <= 0) and (nrows == 0 or width[0] == nv), "row r = the vertices met by more than r slots"
        arr = np.full((4 * T["n_slots"], 3), np.nan)
        ent = rs[rank[real]] + lv[real]                                          # [real slots, 4]
        assert np.all(lv[real] < width[rank[real]]), "an entry outside its row"
        assert len(np.unique(ent)) == ent.size == len(arr), "every entry of the force array is written exactly once"
        arr[ent.reshape(-1)] = contrib[real].reshape(-1, 3)
        gs = np.zeros((nv, 3))
        for r in range(nrows):                                                   # rows in order = the kernel's order of additions
            gs[:width[r]] += arr[rs[r]:rs[r] + width[r]]
        vd = T["vdst"].astype(np.int64)
        excl = vd >= 0
        assert excl.sum() == T["n_excl"] and np.array_equal(vd[excl], T["gvid"][excl])
        assert np.all(np.isnan(grad[vd[excl]])), "an exclusive vertex was written twice"
        grad[vd[excl]] = gs[excl] * grad_output
        rows = ~vd[~excl]
        assert np.array_equal(rows, sdst[T["stage_off"]:T["stage_off"] + nv - T["n_excl"]]), "vdst and the finish lists disagree"
        assert np.all(np.isnan(stage[rows])), "two tile copies write the same staging row"
        stage[rows] = gs[~excl]
    for k in range(len(vid)):
        rows = stage[off[k]:off[k + 1]]
        assert np.all(np.isnan(grad[vid[k]])), "a finish vertex was also written as exclusive"
        grad[vid[k]] = rows.sum(axis=0) * grad_output
    assert not np.isnan(grad).any(), "some vertex received no gradient"
    return c1 * Es + c2 * Eb, Es, Eb, grad
