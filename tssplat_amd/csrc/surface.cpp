// Host side of the surface glue: boundary extraction and the vertex -> face incidence lists the
// per-vertex gathers of surface_kernels.hip walk.  See surface.h for the reference call sites.
#include "surface.h"

#include <algorithm>
#include <array>

#include "../../include/tssplat_amd.h"
#include "plan.h"

namespace tsamd {

int extract_surface(const int32_t *tets, int64_t m, int64_t n, std::vector<int32_t> &surface_vid,
                    std::vector<int32_t> &faces, std::string &err)
{
    surface_vid.clear();
    faces.clear();
    for (int64_t i = 0; i < 4 * m; ++i)
        if (tets[i] < 0 || tets[i] >= n) {
            err = "tet vertex index out of range";
            return TSAMD_ERR_INVALID_ARGUMENT;
        }
    std::vector<int32_t> nbr;
    int rc = build_adjacency(tets, n, m, nbr, 0, err);
    if (rc) return rc;
    // orientation patterns of get_surface_vf (mesh_utils.py:8-13), indexed by the opposite local vertex
    static const int pat[4][3] = {{1, 2, 3}, {0, 3, 2}, {0, 1, 3}, {0, 2, 1}};
    struct Tri {
        std::array<int32_t, 3> key, org;
    };
    std::vector<Tri> tris;
    for (int64_t e = 0; e < m; ++e)
        for (int k = 0; k < 4; ++k) {
            if (nbr[size_t(4 * e + k)] >= 0) continue;
            Tri t;
            for (int j = 0; j < 3; ++j) t.org[size_t(j)] = tets[4 * e + pat[k][j]];
            t.key = t.org;
            std::sort(t.key.begin(), t.key.end());
            tris.push_back(t);
        }
    std::sort(tris.begin(), tris.end(), [](const Tri &a, const Tri &b) { return a.key < b.key; });
    std::vector<int32_t> rank(size_t(n), -1);
    for (const Tri &t : tris)
        for (int32_t v : t.org) rank[size_t(v)] = 0;
    for (int64_t v = 0; v < n; ++v)
        if (rank[size_t(v)] == 0) {
            rank[size_t(v)] = int32_t(surface_vid.size());
            surface_vid.push_back(int32_t(v));
        }
    faces.reserve(3 * tris.size());
    for (const Tri &t : tris)
        for (int32_t v : t.org) faces.push_back(rank[size_t(v)]);
    return TSAMD_OK;
}

void build_vertex_faces(const int32_t *faces, int64_t nf, int64_t nv, std::vector<int32_t> &off,
                        std::vector<int32_t> &ent)
{
    off.assign(size_t(nv + 1), 0);
    for (int64_t i = 0; i < 3 * nf; ++i) ++off[size_t(faces[i]) + 1];
    for (int64_t v = 0; v < nv; ++v) off[size_t(v + 1)] += off[size_t(v)];
    ent.assign(size_t(3 * nf), 0);
    std::vector<int32_t> cur(off.begin(), off.end() - 1);
    for (int64_t f = 0; f < nf; ++f)  // face-major fill => every list ascends by face
        for (int c = 0; c < 3; ++c) ent[size_t(cur[size_t(faces[3 * f + c])]++)] = int32_t(4 * f + c);
}

}  // namespace tsamd
