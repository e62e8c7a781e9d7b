// gfx950 kernels of the surface glue.  All of them are one-thread-per-vertex
// gathers over fixed incidence lists: no atomics, bitwise repeatable, HBM/L2-bound on a few MB.
//
//   surface positions   v_pos = tet_v[surface_vid]                geometry/tetmesh_geometry.py:33
//   vertex normals      area-weighted face normals splatted to the vertices, zero -> (0,0,1), normalised
//                                                                 geometry/tetmesh_geometry.py:39-66
// and their adjoints (what torch autograd derives for the reference's index / cross / scatter_add_ /
// where / normalize chain).
#include <hip/hip_runtime.h>

#include "surface.h"

namespace tsamd {
namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 ld3(const float *p, int64_t i) { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(float *p, int64_t i, V3 v)
{
    p[3 * i] = v.x;
    p[3 * i + 1] = v.y;
    p[3 * i + 2] = v.z;
}
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__global__ __launch_bounds__(256) void surface_positions_kernel(SurfaceArgs s, const float *tet_v, float *v_pos)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < s.nv) st3(v_pos, i, ld3(tet_v, s.surface_vid[i]));
}

// adjoint of the row gather: rows of grad_tet_v that no surface vertex maps to stay 0 (memset by the caller)
template <bool ATOMIC>
__global__ __launch_bounds__(256) void surface_positions_backward_kernel(SurfaceArgs s, const float *g, float *grad_tet_v)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= s.nv) return;
    const V3 v = ld3(g, i);
    float *d = grad_tet_v + 3 * int64_t(s.surface_vid[i]);
    if (ATOMIC) {  // duplicated ids (never produced by get_surface_vf): index_put_(accumulate=True) semantics
        atomicAdd(d, v.x);
        atomicAdd(d + 1, v.y);
        atomicAdd(d + 2, v.z);
    } else {
        d[0] = v.x;
        d[1] = v.y;
        d[2] = v.z;
    }
}

// n_v = sum over incident faces of (v1 - v0) x (v2 - v0), faces in ascending order (tetmesh_geometry.py:40-54);
// |n|^2 <= 1e-20 -> (0, 0, 1) (:57-60); n / max(|n|, 1e-12) (F.normalize, :61)
__global__ __launch_bounds__(256) void vertex_normals_kernel(SurfaceArgs s, const float *v_pos, float *v_nrm, float *raw)
{
    const int64_t v = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (v >= s.nv) return;
    V3 n{0.f, 0.f, 0.f};
    for (int32_t e = s.vf_off[v]; e < s.vf_off[v + 1]; ++e) {
        const int32_t *f = s.faces + 3 * int64_t(s.vf_ent[e] >> 2);
        const V3 p0 = ld3(v_pos, f[0]);
        n = n + cross(ld3(v_pos, f[1]) - p0, ld3(v_pos, f[2]) - p0);
    }
    if (raw) st3(raw, v, n);
    if (!(dot(n, n) > 1e-20f)) n = V3{0.f, 0.f, 1.f};
    const float inv = 1.f / fmaxf(sqrtf(dot(n, n)), 1e-12f);
    st3(v_nrm, v, V3{n.x * inv, n.y * inv, n.z * inv});
}

// h_v = d(loss)/d(n_v) = (g - nhat (nhat . g)) / |n|, 0 where the normal was replaced by the constant
__global__ __launch_bounds__(256) void vertex_normals_backward_a_kernel(int64_t nv, const float *raw, const float *g, float *h)
{
    const int64_t v = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (v >= nv) return;
    const V3 n = ld3(raw, v);
    const float nn = dot(n, n);
    V3 out{0.f, 0.f, 0.f};
    if (nn > 1e-20f) {
        const float inv = 1.f / fmaxf(sqrtf(nn), 1e-12f);
        const V3 nh{n.x * inv, n.y * inv, n.z * inv};
        const V3 gv = ld3(g, v);
        const float t = dot(nh, gv);
        out = V3{(gv.x - nh.x * t) * inv, (gv.y - nh.y * t) * inv, (gv.z - nh.z * t) * inv};
    }
    st3(h, v, out);
}

// With G_f = h_i0 + h_i1 + h_i2 (every corner received the same face normal), e1 = v1 - v0, e2 = v2 - v0:
// d/dv1 = e2 x G, d/dv2 = G x e1, d/dv0 = -(both).  One thread per vertex, faces in ascending order.
__global__ __launch_bounds__(256) void vertex_normals_backward_b_kernel(SurfaceArgs s, const float *v_pos, const float *h, float *grad_v)
{
    const int64_t v = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (v >= s.nv) return;
    V3 acc{0.f, 0.f, 0.f};
    for (int32_t e = s.vf_off[v]; e < s.vf_off[v + 1]; ++e) {
        const int32_t ent = s.vf_ent[e];
        const int32_t *f = s.faces + 3 * int64_t(ent >> 2);
        const V3 p0 = ld3(v_pos, f[0]);
        const V3 e1 = ld3(v_pos, f[1]) - p0, e2 = ld3(v_pos, f[2]) - p0;
        const V3 G = ld3(h, f[0]) + ld3(h, f[1]) + ld3(h, f[2]);
        const V3 d1 = cross(e2, G), d2 = cross(G, e1);
        const int corner = ent & 3;
        if (corner == 1) acc = acc + d1;
        else if (corner == 2) acc = acc + d2;
        else acc = acc - (d1 + d2);
    }
    st3(grad_v, v, acc);
}

unsigned blocks(int64_t n) { return unsigned((n + 255) / 256); }

}  // namespace

hipError_t launch_surface_positions(const SurfaceArgs &s, const float *tet_v, float *v_pos, hipStream_t stream)
{
    if (s.nv <= 0) return hipSuccess;
    hipLaunchKernelGGL(surface_positions_kernel, dim3(blocks(s.nv)), dim3(256), 0, stream, s, tet_v, v_pos);
    return hipGetLastError();
}

hipError_t launch_surface_positions_backward(const SurfaceArgs &s, const float *grad_v_pos, float *grad_tet_v, hipStream_t stream)
{
    if (s.n_tet_vertices > 0) {
        hipError_t e = hipMemsetAsync(grad_tet_v, 0, size_t(s.n_tet_vertices) * 3 * sizeof(float), stream);
        if (e != hipSuccess) return e;
    }
    if (s.nv <= 0) return hipSuccess;
    if (s.unique_vid)
        hipLaunchKernelGGL(surface_positions_backward_kernel<false>, dim3(blocks(s.nv)), dim3(256), 0, stream, s, grad_v_pos, grad_tet_v);
    else
        hipLaunchKernelGGL(surface_positions_backward_kernel<true>, dim3(blocks(s.nv)), dim3(256), 0, stream, s, grad_v_pos, grad_tet_v);
    return hipGetLastError();
}

hipError_t launch_vertex_normals(const SurfaceArgs &s, const float *v_pos, float *v_nrm, float *raw, hipStream_t stream)
{
    if (s.nv <= 0) return hipSuccess;
    hipLaunchKernelGGL(vertex_normals_kernel, dim3(blocks(s.nv)), dim3(256), 0, stream, s, v_pos, v_nrm, raw);
    return hipGetLastError();
}

hipError_t launch_vertex_normals_backward(const SurfaceArgs &s, const float *v_pos, const float *raw, const float *grad_nrm,
                                          float *workspace, float *grad_v_pos, hipStream_t stream)
{
    if (s.nv <= 0) return hipSuccess;
    hipLaunchKernelGGL(vertex_normals_backward_a_kernel, dim3(blocks(s.nv)), dim3(256), 0, stream, s.nv, raw, grad_nrm, workspace);
    hipLaunchKernelGGL(vertex_normals_backward_b_kernel, dim3(blocks(s.nv)), dim3(256), 0, stream, s, v_pos, workspace, grad_v_pos);
    return hipGetLastError();
}

}  // namespace tsamd
