// Surface glue of the geometry module: the steps the reference performs on the
// boundary triangles of the tet mesh right before / after the energy in every iteration
// (/root/reference/geometry/tetmesh_geometry.py:27-66) and the one-off surface extraction
// (/root/reference/geometry/mesh_utils.py:5-35).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

namespace tsamd {

// Boundary triangles of a tet mesh, exactly as geometry/mesh_utils.py::get_surface_vf lists them:
// faces that belong to one tet only, ordered by their sorted vertex triple, each with the orientation
// of the tet-local pattern it came from ([1,2,3], [0,3,2], [0,1,3], [0,2,1] for the face opposite local
// vertex 0..3), vertex ids compacted to their rank among the sorted surface vertex ids.
int extract_surface(const int32_t *tets, int64_t m, int64_t n, std::vector<int32_t> &surface_vid,
                    std::vector<int32_t> &faces, std::string &err);

// vertex -> incident (face, corner) lists: entry = face * 4 + corner, ascending per vertex
void build_vertex_faces(const int32_t *faces, int64_t nf, int64_t nv, std::vector<int32_t> &off,
                        std::vector<int32_t> &ent);

struct SurfaceArgs {
    const int32_t *surface_vid;  // [nv] tet-vertex id of every surface vertex
    const int32_t *faces;        // [nf, 3] surface-vertex ids
    const int32_t *vf_off;       // [nv + 1]
    const int32_t *vf_ent;       // [3 nf] face * 4 + corner
    int64_t nv, nf, n_tet_vertices;
    bool unique_vid;
};

hipError_t launch_surface_positions(const SurfaceArgs &s, const float *tet_v, float *v_pos, hipStream_t stream);
hipError_t launch_surface_positions_backward(const SurfaceArgs &s, const float *grad_v_pos, float *grad_tet_v,
                                             hipStream_t stream);
hipError_t launch_vertex_normals(const SurfaceArgs &s, const float *v_pos, float *v_nrm, float *raw, hipStream_t stream);
hipError_t launch_vertex_normals_backward(const SurfaceArgs &s, const float *v_pos, const float *raw, const float *grad_nrm,
                                          float *workspace, float *grad_v_pos, hipStream_t stream);

}  // namespace tsamd
