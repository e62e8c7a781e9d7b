// Step assignment of the neighbour gathers of passes 2 / 3 against LDS bank conflicts.
//
// A wave reads neighbour k of 16 lanes' tets with one ds_read_b128 per 16-lane group (MI355X_MICROARCH.md, LDS):
// the records of one step collide when their indices agree modulo 16 (48-byte records: 16-byte column =
// 3 * idx mod 16).  The ORDER of a tet's four neighbours is free, so every lane group gets a proper edge
// colouring of its lanes x columns read graph instead of round 1's greedy pick (measured on 512 x kuhn19:
// SQ_LDS_BANK_CONFLICT 58.4 M -> 55.5 M per launch, tile kernel -1 %).
//
// A search over the ASSIGNMENT of tets to lanes on top of this (swap local search against the per-group column
// histograms) was built and measured in round 2: conflicts 55.5 M -> 50.4 M, tile kernel -0.6 %, plan build
// 4.9 s -> 16.8 s.  Removed: conflict cycles are mostly hidden behind other waves (profiles/r02_experiments.md).
#include "conflict_opt.h"

#include <algorithm>
#include <cstring>

namespace tsamd {

void colour_group_reads(int nl, const uint32_t cand[][4], uint32_t zs, int from[][4])
{
    struct Edge {
        int lane, c, vnode, colour;
    };
    Edge edges[64];
    int ne = 0;
    // virtual columns: (column, copy); each takes at most 4 edges, reads of one record stay together while they fit
    int vn_count = 0, vn_col[64], vn_deg[64], col_n[16] = {}, col_vn[16][64];
    uint32_t vn_rec[64][4];
    for (int li = 0; li < nl; ++li)
        for (int c = 0; c < 4; ++c) {
            const uint32_t rec = cand[li][c];
            if (rec == zs) continue;
            const int col = int(rec & 15u);
            int pick = -1;
            // (the copies of one column, in creation order: the same choices as a scan over all virtual nodes)
            for (int k = 0; k < col_n[col] && pick < 0; ++k) {   // a copy of this column that already holds this record
                const int v = col_vn[col][k];
                if (vn_deg[v] < 4)
                    for (int i = 0; i < vn_deg[v]; ++i)
                        if (vn_rec[v][i] == rec) pick = v;
            }
            for (int k = 0; k < col_n[col] && pick < 0; ++k)
                if (vn_deg[col_vn[col][k]] < 4) pick = col_vn[col][k];
            if (pick < 0) {
                pick = vn_count++;
                vn_col[pick] = col;
                vn_deg[pick] = 0;
                col_vn[col][col_n[col]++] = pick;
            }
            vn_rec[pick][vn_deg[pick]++] = rec;
            edges[ne++] = Edge{li, c, pick, -1};
        }
    int at_lane[16][4], at_vn[64][4];   // edge of each colour at each node, -1 = free
    std::memset(at_lane, -1, sizeof(at_lane));
    std::memset(at_vn, -1, sizeof(at_vn));
    for (int e = 0; e < ne; ++e) {
        const int u = edges[e].lane, v = edges[e].vnode;
        int a = -1, b = -1;
        for (int q = 0; q < 4 && a < 0; ++q)
            if (at_lane[u][q] < 0) a = q;
        for (int q = 0; q < 4 && b < 0; ++q)
            if (at_vn[v][q] < 0) b = q;
        if (at_vn[v][a] >= 0) {
            // flip the a/b alternating path that starts at v: afterwards a is free at v (it cannot end at u)
            int path[64], np = 0;
            int x = v, ca = a, cb = b;
            bool on_vn = true;
            for (;;) {
                const int pe = on_vn ? at_vn[x][ca] : at_lane[x][ca];
                if (pe < 0) break;
                path[np++] = pe;
                x = on_vn ? edges[pe].lane : edges[pe].vnode;
                on_vn = !on_vn;
                std::swap(ca, cb);
            }
            for (int i = 0; i < np; ++i) {
                Edge &pe = edges[path[i]];
                at_lane[pe.lane][pe.colour] = -1;
                at_vn[pe.vnode][pe.colour] = -1;
            }
            for (int i = 0; i < np; ++i) {
                Edge &pe = edges[path[i]];
                pe.colour = pe.colour == a ? b : a;
                at_lane[pe.lane][pe.colour] = path[i];
                at_vn[pe.vnode][pe.colour] = path[i];
            }
        }
        edges[e].colour = a;
        at_lane[u][a] = e;
        at_vn[v][a] = e;
    }
    // reads of the zero slot take whatever steps are left
    for (int li = 0; li < nl; ++li) {
        bool used[4] = {false, false, false, false};
        for (int q = 0; q < 4; ++q) from[li][q] = -1;
        for (int q = 0; q < 4; ++q)
            if (at_lane[li][q] >= 0) {
                from[li][q] = edges[at_lane[li][q]].c;
                used[edges[at_lane[li][q]].c] = true;
            }
        int c = 0;
        for (int q = 0; q < 4; ++q)
            if (from[li][q] < 0) {
                while (used[c]) ++c;
                from[li][q] = c;
                used[c] = true;
            }
    }
}

}  // namespace tsamd
