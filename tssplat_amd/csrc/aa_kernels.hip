// gfx950 kernels of the renderer slice, second part: antialias forward / backward and the edge
// partner table it needs -- dr.antialias(color, rast, pos_clip, tri, topology_hash=None, pos_gradient_boost=1.0) at
// /root/reference/renderers/mesh_rasterizer.py:107,128 (the only differentiable path from the alpha image to the
// geometry).  The specification is oracle/raster_oracle.py (edge_partners, _antialias_events, antialias,
// antialias_backward): a restatement of nvdiffrast's published algorithm with every open choice fixed there.
//
//   edge partner table   open-addressing hash of the undirected edges (64-bit key = vertex pair): per slot the two
//                        lowest (triangle, edge) ids, found with atomicMin in two passes so that the result does not depend
//                        on the order of insertion; opp[3 t + e] = far vertex of the partner triangle, -1 on a boundary
//   antialias kernels    one lane per pixel: its pair with the right and with the upper neighbour.  A pair with two different
//                        triangle ids is analysed in float64 with the oracle's operations in the oracle's order, so the SET
//                        of blends is identical; the blends themselves are float32 atomics (compared with a tolerance).
//                        Few lanes ever enter the analysis (silhouette pixels only): the kernels are bound by reading the
//                        image once (16 B of rast + 4 C B of colour per pixel).
#include <hip/hip_runtime.h>

#include "raster.h"

// Coverage, depth and silhouette decisions repeat the oracle's operations one by one: a multiply and an add must round
// separately.  The __fmul_rn / __dmul_rn family are plain operators in this toolchain's headers, so this file is compiled with
// -ffp-contract=off (tssplat_amd/_build.py: SOURCE_FLAGS); tests/test_build_metadata.py checks the flag is in place.

namespace tsamd {
namespace {

constexpr uint32_t kNoEdge = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t edge_key(int32_t a, int32_t b)
{
    const uint32_t lo = uint32_t(min(a, b)), hi = uint32_t(max(a, b));
    return ((uint64_t(lo) << 32) | uint64_t(hi)) + 1ull;   // 0 = empty slot (a key of all ones cannot occur: ids are int32)
}

__device__ __forceinline__ uint64_t edge_hash(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// pass 1: claim / find the slot of every (triangle, edge), lowest id per slot
__global__ __launch_bounds__(256) void topology_insert_kernel(const int32_t *tri, int64_t n_edges, unsigned long long *keys, uint32_t *lo, uint32_t *slot_of,
                                                              uint64_t mask)
{
    const int64_t id = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (id >= n_edges) return;
    const int64_t t = id / 3;
    const int e = int(id - 3 * t);
    const uint64_t key = edge_key(tri[3 * t + (e + 1) % 3], tri[3 * t + (e + 2) % 3]);
    uint64_t slot = edge_hash(key) & mask;
    for (;;) {
        const unsigned long long seen = atomicCAS(keys + slot, 0ull, (unsigned long long)key);
        if (seen == 0ull || seen == key) break;
        slot = (slot + 1) & mask;
    }
    slot_of[id] = uint32_t(slot);
    atomicMin(lo + slot, uint32_t(id));
}

// pass 2: second lowest id per slot
__global__ __launch_bounds__(256) void topology_second_kernel(int64_t n_edges, const uint32_t *lo, uint32_t *hi, const uint32_t *slot_of)
{
    const int64_t id = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (id >= n_edges) return;
    const uint32_t slot = slot_of[id];
    if (lo[slot] != uint32_t(id)) atomicMin(hi + slot, uint32_t(id));
}

// pass 3: partner of (t, e) = the lowest other id on the same edge; opp = the vertex of the partner triangle off the edge
__global__ __launch_bounds__(256) void topology_partner_kernel(const int32_t *tri, int64_t n_edges, const uint32_t *lo, const uint32_t *hi,
                                                               const uint32_t *slot_of, int32_t *opp)
{
    const int64_t id = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (id >= n_edges) return;
    const uint32_t slot = slot_of[id];
    const uint32_t l = lo[slot];
    const uint32_t other = l == uint32_t(id) ? hi[slot] : l;
    opp[id] = other == kNoEdge ? -1 : tri[other];   // tri[3 t' + e'] is the vertex opposite edge e' of triangle t'
}

struct Win {
    double x, y;
    bool ok;
};

// oracle/raster_oracle.py::_window, operation by operation
__device__ __forceinline__ Win window_of(const float4 p, double width, double height)
{
    Win r;
    r.ok = isfinite(p.x) && isfinite(p.y) && isfinite(p.w) && p.w > 0.f;
    const double w = r.ok ? double(p.w) : 1.0;
    r.x = __dmul_rn(__dadd_rn(__dmul_rn(__ddiv_rn(double(p.x), w), 0.5), 0.5), width);
    r.y = __dmul_rn(__dadd_rn(__dmul_rn(__ddiv_rn(double(p.y), w), 0.5), 0.5), height);
    return r;
}

struct Blend {
    int64_t dst, src;      // pixel indices inside the view
    float weight, sign;
    int32_t va, vb;        // the silhouette edge's vertices
    double dAx, dAy, dBx, dBy;   // d t / d window coordinates of the two vertices
};

// oracle/raster_oracle.py::_antialias_events for the pairs (p, p + x) and (p, p + y) of one pixel; `emit` is called per blend.
// `n_vertices` bounds every index read from `tri` / `opp` (a corrupt index skips the pair).
template <class Emit>
__device__ __forceinline__ void pixel_blends(const float4 *rast_view, const float4 *pos_view, const int32_t *tri, const int32_t *opp, int64_t n_vertices,
                                             int64_t n_tri, int height, int width, int j, int i, Emit &&emit)
{
    const float4 r0 = rast_view[int64_t(j) * width + i];
    const int64_t t0 = int64_t(r0.w) - 1;
#pragma unroll
    for (int axis = 0; axis < 2; ++axis) {
        const int dj = axis, di = 1 - axis;
        if (j + dj >= height || i + di >= width) continue;
        const float4 r1 = rast_view[int64_t(j + dj) * width + (i + di)];
        const int64_t t1 = int64_t(r1.w) - 1;
        if (t0 == t1) continue;
        const bool first = (t0 >= 0 && t1 >= 0) ? (r0.z < r1.z) : (t0 >= 0);
        const int64_t t = first ? t0 : t1;
        if (t < 0 || t >= n_tri) continue;
        const int pj = first ? j : j + dj, pi = first ? i : i + di;
        const int qj = first ? j + dj : j, qi = first ? i + di : i;
        const int32_t vid[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
        if (vid[0] < 0 || vid[1] < 0 || vid[2] < 0 || vid[0] >= n_vertices || vid[1] >= n_vertices || vid[2] >= n_vertices) continue;
        const Win w0 = window_of(pos_view[vid[0]], double(width), double(height)), w1 = window_of(pos_view[vid[1]], double(width), double(height)),
                  w2 = window_of(pos_view[vid[2]], double(width), double(height));
        if (!(w0.ok && w1.ok && w2.ok)) continue;
        const double wx[3] = {w0.x, w1.x, w2.x}, wy[3] = {w0.y, w1.y, w2.y};
        const double cx = double(pi) + 0.5, cy = double(pj) + 0.5;
        const double step = axis == 0 ? double(qi - pi) : double(qj - pj);
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int ka = (e + 1) % 3, kb = (e + 2) % 3;
            const double Ax = wx[ka], Ay = wy[ka], Bx = wx[kb], By = wy[kb], Ox = wx[e], Oy = wy[e];
            const double ex = __dsub_rn(Bx, Ax), ey = __dsub_rn(By, Ay);
            const int32_t o2 = opp[3 * t + e];
            if (o2 >= 0 && o2 < n_vertices) {
                const Win wo = window_of(pos_view[o2], double(width), double(height));
                if (wo.ok) {
                    const double s1 = __dsub_rn(__dmul_rn(ex, __dsub_rn(Oy, Ay)), __dmul_rn(ey, __dsub_rn(Ox, Ax)));
                    const double s2 = __dsub_rn(__dmul_rn(ex, __dsub_rn(wo.y, Ay)), __dmul_rn(ey, __dsub_rn(wo.x, Ax)));
                    if ((s1 > 0.0) != (s2 > 0.0)) continue;   // the partner continues the surface on the other side: not a silhouette
                }
            }
            double sA, sB, base, span, centre;
            if (axis == 0) {
                if (!(fabs(ey) >= fabs(ex))) continue;
                sA = __dsub_rn(Ay, cy), sB = __dsub_rn(By, cy), base = Ax, span = ex, centre = cx;
            } else {
                if (!(fabs(ex) >= fabs(ey))) continue;
                sA = __dsub_rn(Ax, cx), sB = __dsub_rn(Bx, cx), base = Ay, span = ey, centre = cy;
            }
            if ((sA > 0.0) == (sB > 0.0)) continue;
            const double den = __dsub_rn(sA, sB);
            const double lam = __ddiv_rn(sA, den);
            const double tt = __dmul_rn(__dsub_rn(__dadd_rn(base, __dmul_rn(span, lam)), centre), step);
            if (!(tt >= 0.0 && tt <= 1.0)) continue;
            const double alpha = __dsub_rn(tt, 0.5);
            if (alpha == 0.0) continue;
            const double den2 = den * den;
            const double dl_a = -sB / den2, dl_b = sA / den2;
            const double along_a = step * (1.0 - lam), along_b = step * lam, across_a = step * span * dl_a, across_b = step * span * dl_b;
            Blend b;
            const int64_t P = int64_t(pj) * width + pi, Q = int64_t(qj) * width + qi;
            b.dst = alpha > 0.0 ? Q : P;
            b.src = alpha > 0.0 ? P : Q;
            b.weight = float(fabs(alpha));
            b.sign = alpha > 0.0 ? 1.f : -1.f;
            b.va = vid[ka];
            b.vb = vid[kb];
            b.dAx = axis == 0 ? along_a : across_a;
            b.dAy = axis == 0 ? across_a : along_a;
            b.dBx = axis == 0 ? along_b : across_b;
            b.dBy = axis == 0 ? across_b : along_b;
            emit(b);
        }
    }
}

// Occupancy: the float64 analysis wants 100 (forward) / 130 (backward) VGPRs, but ~99 % of the lanes only compare two triangle
// ids and their speed is the occupancy the register count leaves (3-5 waves per SIMD).  Both kernels are therefore held to
// 64 VGPRs = 8 waves per SIMD and the analysis spills (2 / 92 dwords of scratch, touched by silhouette lanes only).  Measured,
// forward / forward + backward: 120 views x 512^2 of one object 0.33 / 0.78 -> 0.53 ms for both; 8 dense views 0.087 / 0.23 ->
// 0.067 / 0.26 ms.  (Detect-then-analyse with pair lists was built and measured: a global list serialises on its append counter,
// 0.42 ms for the 8 dense views; per-256-pixel lists without atomics 0.16 / 0.40 and 0.62 ms -- an extra pass over `rast` and
// a second read of every pair cost more than the spills.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void antialias_kernel(const float *color, const float4 *rast, const float4 *pos, const int32_t *tri, const int32_t *opp,
                                                        int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width, int channels, float *out)
{
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t hw = int64_t(height) * width;
    if (gid >= batch * hw) return;
    const int64_t b = gid / hw, pix = gid - b * hw;
    const int j = int(pix / width), i = int(pix - int64_t(j) * width);
    const float *cv = color + b * hw * channels;
    float *ov = out + b * hw * channels;
    pixel_blends(rast + b * hw, pos + b * n_vertices, tri, opp, n_vertices, n_tri, height, width, j, i, [&](const Blend &e) {
        for (int c = 0; c < channels; ++c) atomicAdd(ov + e.dst * channels + c, e.weight * (cv[e.src * channels + c] - cv[e.dst * channels + c]));
    });
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void antialias_backward_kernel(const float *color, const float4 *rast, const float4 *pos, const int32_t *tri,
                                                                 const int32_t *opp, int64_t batch, int64_t n_vertices, int64_t n_tri, int height,
                                                                 int width, int channels, const float *grad_out, float boost, float *grad_color,
                                                                 float4 *grad_pos)
{
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t hw = int64_t(height) * width;
    if (gid >= batch * hw) return;
    const int64_t b = gid / hw, pix = gid - b * hw;
    const int j = int(pix / width), i = int(pix - int64_t(j) * width);
    const float *cv = color + b * hw * channels;
    const float *gv = grad_out + b * hw * channels;
    const float4 *pv = pos + b * n_vertices;
    pixel_blends(rast + b * hw, pv, tri, opp, n_vertices, n_tri, height, width, j, i, [&](const Blend &e) {
        float dot = 0.f;
        for (int c = 0; c < channels; ++c) {
            const float g = gv[e.dst * channels + c];
            dot += g * (cv[e.src * channels + c] - cv[e.dst * channels + c]);
            if (grad_color) {
                float *gc = grad_color + b * hw * channels;
                atomicAdd(gc + e.src * channels + c, e.weight * g);
                atomicAdd(gc + e.dst * channels + c, -e.weight * g);
            }
        }
        if (grad_pos) {
            const double dt = double(e.sign) * double(dot) * double(boost);
            const int32_t vtx[2] = {e.va, e.vb};
            const double dx[2] = {e.dAx, e.dBx}, dy[2] = {e.dAy, e.dBy};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 p = pv[vtx[k]];
                const double x = double(p.x), y = double(p.y), w = double(p.w);
                const double gx = dt * dx[k], gy = dt * dy[k];
                float *gp = reinterpret_cast<float *>(grad_pos + b * n_vertices + vtx[k]);
                atomicAdd(gp + 0, float(gx * (0.5 * double(width) / w)));
                atomicAdd(gp + 1, float(gy * (0.5 * double(height) / w)));
                atomicAdd(gp + 3, float(gx * (-0.5 * double(width) * x / (w * w)) + gy * (-0.5 * double(height) * y / (w * w))));
            }
        }
    });
}

unsigned blocks_for(int64_t n) { return unsigned((n + 255) / 256); }

uint64_t table_slots(int64_t n_tri)
{
    uint64_t s = 64;
    while (s < uint64_t(6 * n_tri)) s <<= 1;   // load factor <= 1/2 even if no edge were shared
    return s;
}

}  // namespace

int64_t antialias_topology_workspace_bytes(int64_t n_tri)
{
    const uint64_t slots = table_slots(n_tri);
    return int64_t(slots * 16u + uint64_t(3 * n_tri) * 4u);
}

hipError_t launch_antialias_topology(const int32_t *tri, int64_t n_tri, void *workspace, int32_t *opp, hipStream_t stream)
{
    if (n_tri <= 0) return hipSuccess;
    const uint64_t slots = table_slots(n_tri);
    unsigned long long *keys = static_cast<unsigned long long *>(workspace);
    uint32_t *lo = reinterpret_cast<uint32_t *>(keys + slots), *hi = lo + slots, *slot_of = hi + slots;
    hipError_t e = hipMemsetAsync(keys, 0, slots * 8u, stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(lo, 0xFF, slots * 8u, stream)) != hipSuccess) return e;
    const int64_t n_edges = 3 * n_tri;
    hipLaunchKernelGGL(topology_insert_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, stream, tri, n_edges, keys, lo, slot_of, slots - 1);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(topology_second_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, stream, n_edges, lo, hi, slot_of);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(topology_partner_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, stream, tri, n_edges, lo, hi, slot_of, opp);
    return hipGetLastError();
}

hipError_t launch_antialias(const float *color, const float *rast, const float *pos_clip, const int32_t *tri, const int32_t *opp, int64_t batch,
                            int64_t n_vertices, int64_t n_tri, int height, int width, int channels, float *out, hipStream_t stream)
{
    const int64_t pixels = batch * int64_t(height) * width;
    if (pixels <= 0 || channels <= 0) return hipSuccess;
    hipError_t e = hipMemcpyAsync(out, color, size_t(pixels) * size_t(channels) * sizeof(float), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    if (n_tri <= 0) return hipSuccess;
    hipLaunchKernelGGL(antialias_kernel, dim3(blocks_for(pixels)), dim3(256), 0, stream, color, reinterpret_cast<const float4 *>(rast),
                       reinterpret_cast<const float4 *>(pos_clip), tri, opp, batch, n_vertices, n_tri, height, width, channels, out);
    return hipGetLastError();
}

hipError_t launch_antialias_backward(const float *color, const float *rast, const float *pos_clip, const int32_t *tri, const int32_t *opp, int64_t batch,
                                     int64_t n_vertices, int64_t n_tri, int height, int width, int channels, const float *grad_out, float boost,
                                     float *grad_color, float *grad_pos, hipStream_t stream)
{
    const int64_t pixels = batch * int64_t(height) * width;
    hipError_t e;
    if (grad_color && pixels > 0 && channels > 0) {
        e = hipMemcpyAsync(grad_color, grad_out, size_t(pixels) * size_t(channels) * sizeof(float), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    if (grad_pos && batch * n_vertices > 0) {
        e = hipMemsetAsync(grad_pos, 0, size_t(batch) * size_t(n_vertices) * 4 * sizeof(float), stream);
        if (e != hipSuccess) return e;
    }
    if (pixels <= 0 || channels <= 0 || n_tri <= 0) return hipSuccess;
    hipLaunchKernelGGL(antialias_backward_kernel, dim3(blocks_for(pixels)), dim3(256), 0, stream, color, reinterpret_cast<const float4 *>(rast),
                       reinterpret_cast<const float4 *>(pos_clip), tri, opp, batch, n_vertices, n_tri, height, width, channels, grad_out, boost, grad_color,
                       reinterpret_cast<float4 *>(grad_pos));
    return hipGetLastError();
}

}  // namespace tsamd
