"""float64 replay of a STREAMING plan (csrc/stream_plan.h) on the CPU, step by step as the kernel runs it -- TEST CODE.

Every ring (F, H, forces, penalty factor, positions / accumulators) is emulated with the tag of the band (or vertex) that
wrote each entry, and every read asserts the tag: a plan whose lags, ring sizes or vertex-slot reuse were wrong fails here
before it ever reaches a GPU.  The arithmetic is the oracle's (float64); results are compared with
oracle/tet_energy_oracle.py by the tests."""
import struct

import numpy as np

BAND = 256
F_RING, H_RING, D_RING, S_RING = 4, 4, 2, 6
LAG_P2, LAG_P3, LAG_SUM = 2, 4, 5
OWNED = 0x8000


def _det3(F):
    return (-F[2] * F[4] * F[6] + F[1] * F[5] * F[6] + F[2] * F[3] * F[7] - F[0] * F[5] * F[7] - F[1] * F[3] * F[8] + F[0] * F[4] * F[8])


def _cof3(F):
    return np.array([F[4] * F[8] - F[5] * F[7], F[5] * F[6] - F[3] * F[8], F[3] * F[7] - F[4] * F[6],
                     F[2] * F[7] - F[1] * F[8], F[0] * F[8] - F[2] * F[6], F[1] * F[6] - F[0] * F[7],
                     F[1] * F[5] - F[2] * F[4], F[2] * F[3] - F[0] * F[5], F[0] * F[4] - F[1] * F[3]])


def parse_tube(blob: np.ndarray, n_bands: int):
    raw = blob.tobytes()
    bands = []
    for b in range(n_bands):
        planes_off, enter_off, pairs_off, chunks_off, n_slots, n_owned, n_enter, n_pairs = struct.unpack_from("<IIIIHHHH", raw, 24 * b)
        planes = np.frombuffer(raw, dtype=np.uint32, count=13 * BAND, offset=planes_off).reshape(13, BAND)
        enter = np.frombuffer(raw, dtype=np.int32, count=2 * n_enter, offset=enter_off).reshape(n_enter, 2)
        pairs = [struct.unpack_from("<HHIi", raw, pairs_off + 12 * p) for p in range(n_pairs)]
        n_ch = max([first + (flags & 0x3fff) for _, flags, first, _ in pairs], default=0)
        chunks = np.frombuffer(raw, dtype=np.uint16, count=4 * n_ch, offset=chunks_off).reshape(n_ch, 4)
        bands.append(dict(planes=planes, dm=planes[4:13].view(np.float32).astype(np.float64), enter=enter, pairs=pairs, chunks=chunks,
                          n_slots=n_slots, n_owned=n_owned))
    return bands


def emulate(st, x, c1, c2, order, grad_output=1.0):
    """``(E, E_s, E_b, grad)`` of a ``StreamTetSpheres`` plan (host_only is enough) in float64."""
    c1 = float(np.float32(c1))
    c2 = float(np.float32(c2))
    x = np.asarray(x, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    info = st.plan_info()
    fin_vid, fin_off, n_stage = st.finish_lists()
    stage = np.full((n_stage, 3), np.nan)
    grad = np.full_like(x, np.nan)
    written = np.zeros(x.shape[0], dtype=np.int64)
    Es = Eb = 0.0
    slots = 0
    for t in range(info["n_tubes"]):
        nb, nvs, n_owned, n_slots, blob, slot_tet = st.tube(t)
        bands = parse_tube(blob, nb)
        assert sum(b["n_slots"] for b in bands) == n_slots and sum(b["n_owned"] for b in bands) == n_owned
        slots += n_slots
        Fr = np.zeros((F_RING, BAND, 9)); Ft = -np.ones(F_RING, dtype=np.int64)
        Hr = np.zeros((H_RING, BAND, 9)); Ht = -np.ones(H_RING, dtype=np.int64)
        Dr = np.zeros((D_RING, BAND + 1, 12)); Dt = -np.ones(D_RING, dtype=np.int64)
        Sr = np.zeros((S_RING, BAND)); St = -np.ones(S_RING, dtype=np.int64)
        xs = np.full((nvs, 3), np.nan); xt = -np.ones(nvs, dtype=np.int64)      # tag = global vertex id
        acc = np.full((nvs, 3), np.nan)
        slot_vertex = {}                                                          # ring slot -> (vertex, last band) for reuse checks

        def F_of(band, lane, pos):
            lv01, lv23 = int(band["planes"][0, lane]), int(band["planes"][1, lane])
            o = [(lv01 & 0x7fff) >> 4, lv01 >> 20, (lv23 & 0xffff) >> 4, lv23 >> 20]
            assert all(xt[k] >= 0 for k in o), "pass reads a position-ring slot that holds no vertex"
            p = pos[o]
            Ds = np.stack([p[1] - p[0], p[2] - p[0], p[3] - p[0]], axis=1)       # columns = edges
            return (Ds @ band["dm"][:, lane].reshape(3, 3)).reshape(9), o

        def nbr_rec(ring, tags, b, tok):
            delta, lane = ((tok >> 8) & 3) - 1, tok & 255
            r = (b + delta) % ring.shape[0]
            assert tags[r] == b + delta, f"ring slot {r} holds band {tags[r]}, wanted {b + delta}"
            return ring[r, lane]

        for s in range(-1, nb + LAG_SUM):
            # the four stages read the state left by step s - 1: compute everything, then commit
            commits = []
            # ---- group 0: prefetch band s + 1 (entering vertices), pass 1 on band s ----
            if 0 <= s < nb:
                band = bands[s]
                Fnew = np.zeros((BAND, 9)); Snew = np.zeros(BAND)
                for lane in range(band["n_slots"]):
                    F, o = F_of(band, lane, xs)
                    Fnew[lane] = F
                    if int(band["planes"][0, lane]) & OWNED:
                        assert int(band["planes"][2, lane]) & OWNED
                        J = _det3(F)
                        Jm = max(-J, 0.0)
                        if order == 2:
                            Eb += Jm * Jm
                            Snew[lane] = -2.0 * Jm
                        elif order == 4:
                            Eb += Jm ** 4
                            Snew[lane] = -4.0 * Jm ** 3
                commits.append(("F", s, Fnew, Snew))
            if s + 1 < nb:
                commits.append(("enter", s + 1, bands[s + 1]["enter"]))
            # ---- group 1: pass 2 on band s - 2 ----
            b = s - LAG_P2
            if 0 <= b < nb:
                band = bands[b]
                Hnew = np.zeros((BAND, 9))
                assert Ft[b % F_RING] == b
                for lane in range(band["n_slots"]):
                    n01, n23 = int(band["planes"][2, lane]), int(band["planes"][3, lane])
                    if not n01 & OWNED:
                        continue
                    toks = [n01 & 0x3ff, (n01 >> 16) & 0x3ff, n23 & 0x3ff, (n23 >> 16) & 0x3ff]
                    h = 4.0 * Fr[b % F_RING, lane] - sum(nbr_rec(Fr, Ft, b, tk) for tk in toks)
                    Hnew[lane] = h
                    Es += 0.5 * float(h @ h)
                commits.append(("H", b, Hnew))
            # ---- group 2: pass 3 on band s - 4 ----
            b = s - LAG_P3
            if 0 <= b < nb:
                band = bands[b]
                Dnew = np.zeros((BAND + 1, 12))
                assert Ht[b % H_RING] == b and St[b % S_RING] == b
                for lane in range(band["n_slots"]):
                    n01, n23 = int(band["planes"][2, lane]), int(band["planes"][3, lane])
                    toks = [n01 & 0x3ff, (n01 >> 16) & 0x3ff, n23 & 0x3ff, (n23 >> 16) & 0x3ff]
                    q = 4.0 * Hr[b % H_RING, lane] - sum(nbr_rec(Hr, Ht, b, tk) for tk in toks)
                    P = c1 * q
                    dp = Sr[b % S_RING, lane]
                    if dp != 0.0:
                        F, _ = F_of(band, lane, xs)
                        P = P + c2 * dp * _cof3(F)
                    d = P.reshape(3, 3) @ band["dm"][:, lane].reshape(3, 3).T           # d[i, k]: force on local vertex k + 1
                    Dnew[lane, 0:3] = -d.sum(axis=1)
                    for k in range(3):
                        Dnew[lane, 3 + 3 * k: 6 + 3 * k] = d[:, k]
                commits.append(("D", b, Dnew))
            # ---- group 3: vertex sums of band s - 5 ----
            b = s - LAG_SUM
            if 0 <= b < nb:
                band = bands[b]
                assert Dt[b % D_RING] == b
                for vslot, flags, first, out_row in band["pairs"]:
                    nch = flags & 0x3fff
                    ent = band["chunks"][first:first + nch].reshape(-1).astype(np.int64)
                    assert np.all((ent >> 2 < band["n_slots"]) | (ent == BAND << 2)), "incidence entry names a padding lane"
                    g = Dr[b % D_RING].reshape(-1)[(3 * ent)[:, None] + np.arange(3)[None, :]].sum(axis=0)
                    assert not np.isnan(acc[vslot]).any(), "accumulator of a vertex that never entered"
                    acc[vslot] += g
                    if flags & 0x8000:
                        v, last = slot_vertex[vslot]
                        assert last == b, "vertex written out before / after its last band"
                        if flags & 0x4000:
                            assert np.isnan(stage[out_row]).all(), "staging row written twice"
                            stage[out_row] = acc[vslot]
                        else:
                            assert out_row == v
                            grad[out_row] = acc[vslot]
                            written[out_row] += 1
                        acc[vslot] = np.nan
                        xt[vslot] = -2 - b                                               # free from band b on (checked at reuse)
            # ---- commit (the barrier) ----
            for c in commits:
                if c[0] == "F":
                    Fr[c[1] % F_RING] = c[2]; Ft[c[1] % F_RING] = c[1]
                    Sr[c[1] % S_RING] = c[3]; St[c[1] % S_RING] = c[1]
                elif c[0] == "H":
                    Hr[c[1] % H_RING] = c[2]; Ht[c[1] % H_RING] = c[1]
                elif c[0] == "D":
                    Dr[c[1] % D_RING] = c[2]; Dt[c[1] % D_RING] = c[1]
                else:
                    for vslot, v in c[2]:
                        assert xt[vslot] < 0, f"ring slot {vslot} reused while vertex {xt[vslot]} is still alive"
                        if xt[vslot] <= -2:
                            assert c[1] - 1 > (-2 - xt[vslot]) + LAG_SUM, "ring slot reused before its previous vertex was written out"
                        xs[vslot] = x[v]; xt[vslot] = v
                        acc[vslot] = 0.0
                        slot_vertex[vslot] = [v, -1]
            # last bands of the live vertices (from the pairs' flags) -- recorded when the band's pairs are known
            if 0 <= s + 1 < nb:
                for vslot, flags, first, out_row in bands[s + 1]["pairs"]:
                    if flags & 0x8000:
                        slot_vertex[vslot][1] = s + 1
        assert (xt < 0).all(), "a vertex was never written out"
    # finish: shared vertices = sum of their staging rows, in order
    for k, v in enumerate(fin_vid):
        rows = stage[fin_off[k]:fin_off[k + 1]]
        assert not np.isnan(rows).any(), "a staging row was never written"
        grad[v] = rows.sum(axis=0)
        written[v] += 1
    assert (written == 1).all() or x.shape[0] == 0, "a vertex was written not exactly once"
    assert slots == info["total_slots"]
    return c1 * Es + c2 * Eb, Es, Eb, grad * grad_output
