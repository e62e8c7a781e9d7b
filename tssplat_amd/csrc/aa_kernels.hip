// gfx950 kernels of the renderer slice, second part: antialias forward / backward and the edge
// partner table it needs -- dr.antialias(color, rast, pos_clip, tri, topology_hash=None, pos_gradient_boost=1.0) at
// /root/reference/renderers/mesh_rasterizer.py:107,128 (the only differentiable path from the alpha image to the
// geometry).  The specification is oracle/raster_oracle.py (edge_partners, _antialias_events, antialias,
// antialias_backward): a restatement of nvdiffrast's published algorithm with every open choice fixed there.
//
//   edge partner table   open-addressing hash of the undirected edges (64-bit key = vertex pair): per slot the two
//                        lowest (triangle, edge) ids, found with atomicMin in two passes so that the result does not depend
//                        on the order of insertion; opp[3 t + e] = far vertex of the partner triangle, -1 on a boundary
//   antialias kernels    detect per pixel (its pair with the right and with the upper neighbour), analyse per workgroup: the pairs
//                        with two different triangle ids are compacted in LDS and analysed in float64 with the oracle's
//                        operations, so the SET of blends is identical; the blends themselves are float32 atomics (compared
//                        with a tolerance).
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

// The window coordinates of a (view, vertex) as the analysis reads them: either computed on the spot or, when the caller ran
// antialias_windows_kernel first, loaded from its table (x = NaN marks a vertex that is not in front of the camera).  Same
// operations either way -- the table only takes the float64 divisions out of the per-pair analysis, where every pair of
// pixels on two different triangles (every interior triangle boundary, not just the silhouette) pays for four vertices.
template <bool TABLE>
__device__ __forceinline__ Win window_at(const float4 *pos_view, const double2 *win_view, int32_t v, double width, double height)
{
    if (!TABLE) return window_of(pos_view[v], width, height);
    const double2 w = win_view[v];
    Win r;
    r.x = w.x, r.y = w.y, r.ok = !isnan(w.x);
    return r;
}

__global__ __launch_bounds__(256) void antialias_windows_kernel(const float4 *pos, int64_t n, double width, double height, double2 *out)
{
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const Win w = window_of(pos[gid], width, height);
    out[gid] = make_double2(w.ok ? w.x : __longlong_as_double(0x7ff8000000000000ll), w.y);
}

// tsamd_antialias_prepare, third part: per (view, triangle) one byte, bit e = edge e (opposite vertex e) can blend -- it has no
// partner, or the partner's far vertex lies on the SAME side of the edge as this triangle's (a fold: the silhouette); 0 for a
// triangle with a vertex that is out of range or not in front of the camera.  This is the partner test of
// oracle/raster_oracle.py::_antialias_events with its operations, taken out of the per-pair analysis because it does not
// depend on the pixel: most pairs of pixels on two different triangles are interior triangle boundaries, and with the table
// their analysis ends at one byte.
__global__ __launch_bounds__(256) void antialias_edge_flags_kernel(const double2 *windows, const int32_t *tri, const int32_t *opp, int64_t batch, int64_t n_vertices,
                                                                   int64_t n_tri, uint8_t *flags)
{
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= batch * n_tri) return;
    const int64_t b = gid / n_tri, t = gid - b * n_tri;
    const double2 *wv = windows + b * n_vertices;
    const int32_t vid[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    uint32_t f = 0;
    if (!(vid[0] < 0 || vid[1] < 0 || vid[2] < 0 || vid[0] >= n_vertices || vid[1] >= n_vertices || vid[2] >= n_vertices)) {
        const double2 w[3] = {wv[vid[0]], wv[vid[1]], wv[vid[2]]};
        if (!(isnan(w[0].x) || isnan(w[1].x) || isnan(w[2].x))) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int ka = (e + 1) % 3, kb = (e + 2) % 3;
                const double Ax = w[ka].x, Ay = w[ka].y, Ox = w[e].x, Oy = w[e].y;
                const double ex = __dsub_rn(w[kb].x, Ax), ey = __dsub_rn(w[kb].y, Ay);
                bool blend = true;
                const int32_t o2 = opp[3 * t + e];
                if (o2 >= 0 && o2 < n_vertices) {
                    const double2 wo = wv[o2];
                    if (!isnan(wo.x)) {
                        const double s1 = __dsub_rn(__dmul_rn(ex, __dsub_rn(Oy, Ay)), __dmul_rn(ey, __dsub_rn(Ox, Ax)));
                        const double s2 = __dsub_rn(__dmul_rn(ex, __dsub_rn(wo.y, Ay)), __dmul_rn(ey, __dsub_rn(wo.x, Ax)));
                        blend = (s1 > 0.0) == (s2 > 0.0);
                    }
                }
                f |= blend ? (1u << e) : 0u;
            }
        }
    }
    flags[gid] = uint8_t(f);
}

struct Blend {
    int64_t dst, src;      // pixel indices inside the view
    float weight, sign;
    int32_t va, vb;        // the silhouette edge's vertices
    double dAx, dAy, dBx, dBy;   // d t / d window coordinates of the two vertices
};

// oracle/raster_oracle.py::_antialias_events for ONE pair of pixels -- (j, i) with its right (axis 0) or its upper (axis 1)
// neighbour, which the caller has found to carry two different triangle ids; `emit` is called per blend.  `n_vertices`
// bounds every index read from `tri` / `opp` (a corrupt index skips the pair).
template <bool TABLE, class Emit>
__device__ __forceinline__ void pair_blends(const float4 *rast_view, const float4 *pos_view, const double2 *win_view, const uint8_t *flag_view, const int32_t *tri, const int32_t *opp, int64_t n_vertices,
                                            int64_t n_tri, int height, int width, int j, int i, int axis, Emit &&emit)
{
    const int dj = axis, di = 1 - axis;
    const float4 r0 = rast_view[int64_t(j) * width + i];
    const float4 r1 = rast_view[int64_t(j + dj) * width + (i + di)];
    const int64_t t0 = int64_t(r0.w) - 1, t1 = int64_t(r1.w) - 1;
    const bool first = (t0 >= 0 && t1 >= 0) ? (r0.z < r1.z) : (t0 >= 0);
    const int64_t t = first ? t0 : t1;
    if (t < 0 || t >= n_tri) return;
    uint32_t may_blend = 7u;
    if (TABLE) {
        may_blend = flag_view[t];
        if (may_blend == 0u) return;   // (also: a vertex out of range or behind the camera)
    }
    const int pj = first ? j : j + dj, pi = first ? i : i + di;
    const int qj = first ? j + dj : j, qi = first ? i + di : i;
    const int32_t vid[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    if (vid[0] < 0 || vid[1] < 0 || vid[2] < 0 || vid[0] >= n_vertices || vid[1] >= n_vertices || vid[2] >= n_vertices) return;
    const Win w0 = window_at<TABLE>(pos_view, win_view, vid[0], double(width), double(height)), w1 = window_at<TABLE>(pos_view, win_view, vid[1], double(width), double(height)),
              w2 = window_at<TABLE>(pos_view, win_view, vid[2], double(width), double(height));
    if (!(w0.ok && w1.ok && w2.ok)) return;
    const double cx = double(pi) + 0.5, cy = double(pj) + 0.5;
    const double step = axis == 0 ? double(qi - pi) : double(qj - pj);
    // edge e is opposite vertex e: (O, A, B) = vertices (e, e + 1, e + 2) mod 3, rotated in registers from one edge to the next
    // (one copy of the loop body -- and of the caller's emit -- instead of three: the backward kernel's registers)
    double Ox = w0.x, Oy = w0.y, Ax = w1.x, Ay = w1.y, Bx = w2.x, By = w2.y;
    int32_t vo = vid[0], va = vid[1], vb = vid[2];
#pragma unroll 1
    for (int e = 0; e < 3; ++e) {
        if (e > 0) {
            const double tx = Ox, ty = Oy;
            const int32_t tv = vo;
            Ox = Ax, Oy = Ay, vo = va;
            Ax = Bx, Ay = By, va = vb;
            Bx = tx, By = ty, vb = tv;
        }
        if (TABLE && !((may_blend >> e) & 1u)) continue;
        const double ex = __dsub_rn(Bx, Ax), ey = __dsub_rn(By, Ay);
        double sA, sB, base, span, centre;
        if (axis == 0) {
            if (!(fabs(ey) >= fabs(ex))) continue;
            sA = __dsub_rn(Ay, cy), sB = __dsub_rn(By, cy), base = Ax, span = ex, centre = cx;
        } else {
            if (!(fabs(ex) >= fabs(ey))) continue;
            sA = __dsub_rn(Ax, cx), sB = __dsub_rn(Bx, cx), base = Ay, span = ey, centre = cy;
        }
        if ((sA > 0.0) == (sB > 0.0)) continue;
        // (the partner test reads another vertex and divides: behind the two tests above, which reject most edges -- the
        // conditions commute, the set of blends is the oracle's)
        const int32_t o2 = TABLE ? -1 : opp[3 * t + e];   // (TABLE: the partner test is the edge's flag)
        if (o2 >= 0 && o2 < n_vertices) {
            const Win wo = window_at<TABLE>(pos_view, win_view, o2, double(width), double(height));
            if (wo.ok) {
                const double s1 = __dsub_rn(__dmul_rn(ex, __dsub_rn(Oy, Ay)), __dmul_rn(ey, __dsub_rn(Ox, Ax)));
                const double s2 = __dsub_rn(__dmul_rn(ex, __dsub_rn(wo.y, Ay)), __dmul_rn(ey, __dsub_rn(wo.x, Ax)));
                if ((s1 > 0.0) != (s2 > 0.0)) continue;   // the partner continues the surface on the other side: not a silhouette
            }
        }
        const double den = __dsub_rn(sA, sB);
        const double lam = __ddiv_rn(sA, den);
        const double tt = __dmul_rn(__dsub_rn(__dadd_rn(base, __dmul_rn(span, lam)), centre), step);
        if (!(tt >= 0.0 && tt <= 1.0)) continue;
        const double alpha = __dsub_rn(tt, 0.5);
        if (alpha == 0.0) continue;
        const double inv_den2 = 1.0 / (den * den);   // (gradient values, compared with a tolerance: one division, not two)
        const double dl_a = -sB * inv_den2, dl_b = sA * inv_den2;
        const double along_a = step * (1.0 - lam), along_b = step * lam, across_a = step * span * dl_a, across_b = step * span * dl_b;
        Blend b;
        const int64_t P = int64_t(pj) * width + pi, Q = int64_t(qj) * width + qi;
        b.dst = alpha > 0.0 ? Q : P;
        b.src = alpha > 0.0 ? P : Q;
        b.weight = float(fabs(alpha));
        b.sign = alpha > 0.0 ? 1.f : -1.f;
        b.va = va;
        b.vb = vb;
        b.dAx = axis == 0 ? along_a : across_a;
        b.dAy = axis == 0 ? across_a : along_a;
        b.dBx = axis == 0 ? along_b : across_b;
        b.dBy = axis == 0 ? across_b : along_b;
        emit(b);
    }
}

// Both kernels in two phases per workgroup of 256 consecutive pixels.  DETECT: every lane compares its triangle id with its right
// and its upper neighbour's (three dwords of `rast` per pixel) and appends the pairs that differ -- silhouette and crease pixels,
// ~1 % of an image -- to a list in LDS (one LDS atomic per wave and axis).  ANALYSE: the workgroup's pairs, dealt densely to
// lanes from lane 0 on, go through the float64 analysis.  One lane per pixel doing both (the earlier form) ran the analysis in
// every wave holding a silhouette pixel with 1-2 lanes active, both axes one after the other: on 120 views x 512^2 of one object a
// third of all waves.  Here the analysis runs once per workgroup that has pairs, mostly in one wave, and the other waves leave
// after the detect phase.  (Pair lists in GLOBAL memory were built in round 3 and lost to their append counter / second pass.)
constexpr int kAaBlock = 256;

struct PairList {
    uint32_t items[2 * kAaBlock];   // (pixel within the workgroup) << 1 | axis
    int count;
};

// the two pairs of pixel `gid`: c[axis] = its triangle id differs from its right (axis 0) / upper (axis 1) neighbour's
__device__ __forceinline__ void differing_neighbours(const float4 *rast, int64_t gid, int64_t total, int64_t hw, int height, int width, bool c[2])
{
    c[0] = c[1] = false;
    if (gid >= total) return;
    const int64_t pix = gid % hw;
    const int j = int(pix / width), i = int(pix - int64_t(j) * width);
    const float *w = reinterpret_cast<const float *>(rast + gid) + 3;
    const float t0 = w[0];
    c[0] = i + 1 < width && w[4] != t0;
    c[1] = j + 1 < height && w[4 * int64_t(width)] != t0;
}

// tsamd_antialias_prepare, second half: the pair masks of an image -- per 64 consecutive pixels two 64-bit words (axis 0, axis 1),
// bit l = pixel 64 k + l has a differing neighbour.  2 bits per pixel instead of three strided dwords of `rast`: the forward and
// the backward kernel read the masks, so the image is scanned once per antialias call pair, not twice.
__global__ __launch_bounds__(kAaBlock) void antialias_detect_kernel(const float4 *rast, int64_t total, int64_t hw, int height, int width, unsigned long long *masks)
{
    const int64_t gid = int64_t(blockIdx.x) * kAaBlock + threadIdx.x;
    bool c[2];
    differing_neighbours(rast, gid, total, hw, height, width, c);
    const unsigned long long m0 = __ballot(c[0]), m1 = __ballot(c[1]);
    if ((threadIdx.x & 63) == 0 && gid < total) {
        masks[2 * (gid >> 6)] = m0;
        masks[2 * (gid >> 6) + 1] = m1;
    }
}

// ---- drivers: call work(pixel, axis) for every pair of an image, lanes filled densely ----
//
// Without tables (prepared_dev = NULL): one workgroup per 256 consecutive pixels, detect + compaction in LDS, two barriers.
template <class Work>
__device__ __forceinline__ void for_pairs_detected(const float4 *rast, int64_t total, int64_t hw, int height, int width, PairList &L, Work &&work)
{
    const int64_t first_pixel = int64_t(blockIdx.x) * kAaBlock;
    if (threadIdx.x == 0) L.count = 0;
    __syncthreads();
    bool c[2];
    differing_neighbours(rast, first_pixel + threadIdx.x, total, hw, height, width, c);
#pragma unroll
    for (int axis = 0; axis < 2; ++axis) {
        const unsigned long long mask = __ballot(c[axis]);
        if (mask == 0ull) continue;
        int base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&L.count, __popcll(mask));
        base = __builtin_amdgcn_readfirstlane(base);
        if (c[axis]) L.items[base + int(__builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)))] = (threadIdx.x << 1) | uint32_t(axis);
    }
    __syncthreads();
    const int n = L.count;
    for (int k = threadIdx.x; k < n; k += kAaBlock) {
        const uint32_t item = L.items[k];
        work(first_pixel + (item >> 1), int(item & 1u));
    }
}

// With the pair masks of tsamd_antialias_prepare: no lane per pixel at all.  A wave takes `group` (<= 64) consecutive 64-pixel
// chunks, lane l loads the two mask words of chunk l, and the wave walks the chunks that have a pair (most have none and cost
// nothing beyond that one coalesced load): a chunk's pairs are pushed on a small per-wave stack in LDS (position from the mask's
// bit count, no atomic), and whenever the stack holds 64 they are analysed, one pair per lane -- full waves in the float64
// analysis whatever the image looks like, no barrier, no workgroup-wide step.
constexpr int kPairStack = 192;   // < 64 left over + at most 128 pushed per chunk

__device__ __forceinline__ unsigned long long read_lane_u64(unsigned long long v, int lane)
{
    const uint32_t lo = uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v)), lane)), hi = uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v >> 32)), lane));
    return ((unsigned long long)hi << 32) | lo;
}

template <class Work>
__device__ __forceinline__ void for_pairs_masked(const unsigned long long *masks, int64_t n_chunks, int group, uint16_t *stack, Work &&work)
{
    const int lane = int(threadIdx.x) & 63;
    const int64_t chunk0 = (int64_t(blockIdx.x) * (kAaBlock / 64) + __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))) * group;
    if (chunk0 >= n_chunks) return;
    unsigned long long m0 = 0ull, m1 = 0ull;
    if (lane < group && chunk0 + lane < n_chunks) {
        m0 = masks[2 * (chunk0 + lane)];
        m1 = masks[2 * (chunk0 + lane) + 1];
    }
    unsigned long long todo = __ballot((m0 | m1) != 0ull);
    int count = 0;   // wave-uniform
    for (;;) {
        while (count < 64 && todo != 0ull) {   // push the pairs of the next chunk that has any
            const int c = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            todo &= todo - 1ull;
#pragma unroll
            for (int axis = 0; axis < 2; ++axis) {
                const unsigned long long m = read_lane_u64(axis ? m1 : m0, c);
                if ((m >> lane) & 1ull)
                    stack[count + int(__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)))] = uint16_t((c << 7) | (lane << 1) | axis);
                count += __popcll(m);
            }
        }
        if (count == 0) break;
        const int n = min(count, 64);   // the top of the stack: a full wave of pairs except at the very end
        count -= n;
        __builtin_amdgcn_wave_barrier();
        if (lane < n) {
            const uint32_t item = stack[count + lane];
            work((chunk0 + (item >> 7)) * 64 + ((item >> 1) & 63u), int(item & 1u));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// chunks per wave of the masked form: enough waves to fill the chip (>= ~16 k) before a wave takes more than one chunk
int masked_group(int64_t n_chunks)
{
    int g = 1;
    while (g < 64 && n_chunks / (2 * g) >= 16384) g *= 2;
    return g;
}

template <bool TABLE>
__device__ __forceinline__ void antialias_body(const float *color, const float4 *rast, const float4 *pos, const double2 *windows, const unsigned long long *masks, const uint8_t *flags, const int32_t *tri, const int32_t *opp,
                                                        int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width, int channels, int group, float *out)
{
    const int64_t hw = int64_t(height) * width;
    auto work = [&](int64_t gid, int axis) {
        const int64_t b = gid / hw, pix = gid - b * hw;
        const int j = int(pix / width), i = int(pix - int64_t(j) * width);
        const float *cv = color + b * hw * channels;
        float *ov = out + b * hw * channels;
        pair_blends<TABLE>(rast + b * hw, pos + b * n_vertices, TABLE ? windows + b * n_vertices : nullptr, TABLE ? flags + b * n_tri : nullptr, tri, opp, n_vertices, n_tri, height, width, j, i, axis, [&](const Blend &e) {
            for (int c = 0; c < channels; ++c) atomicAdd(ov + e.dst * channels + c, e.weight * (cv[e.src * channels + c] - cv[e.dst * channels + c]));
        });
    };
    if (TABLE) {
        __shared__ uint16_t stacks[(kAaBlock / 64) * kPairStack];
        for_pairs_masked(masks, (batch * hw + 63) / 64, group, stacks + (threadIdx.x >> 6) * kPairStack, work);
    } else {
        __shared__ PairList L;
        for_pairs_detected(rast, batch * hw, hw, height, width, L, work);
    }
}

template <bool TABLE>
__device__ __forceinline__ void antialias_backward_body(const float *color, const float4 *rast, const float4 *pos, const double2 *windows, const unsigned long long *masks, const uint8_t *flags, const int32_t *tri,
                                                                 const int32_t *opp, int64_t batch, int64_t n_vertices, int64_t n_tri, int height,
                                                                 int width, int channels, int group, const float *grad_out, float boost, float *grad_color,
                                                                 float4 *grad_pos)
{
    const int64_t hw = int64_t(height) * width;
    auto work = [&](int64_t gid, int axis) {
        const int64_t b = gid / hw, pix = gid - b * hw;
        const int j = int(pix / width), i = int(pix - int64_t(j) * width);
        const float *cv = color + b * hw * channels;
        const float *gv = grad_out + b * hw * channels;
        const float4 *pv = pos + b * n_vertices;
        pair_blends<TABLE>(rast + b * hw, pv, TABLE ? windows + b * n_vertices : nullptr, TABLE ? flags + b * n_tri : nullptr, tri, opp, n_vertices, n_tri, height, width, j, i, axis, [&](const Blend &e) {
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
                for (int k2 = 0; k2 < 2; ++k2) {
                    const float4 p = pv[vtx[k2]];
                    const double x = double(p.x), y = double(p.y), iw = 1.0 / double(p.w);
                    const double gx = dt * dx[k2] * (0.5 * double(width) * iw), gy = dt * dy[k2] * (0.5 * double(height) * iw);   // d loss / d (x, y)
                    float *gp = reinterpret_cast<float *>(grad_pos + b * n_vertices + vtx[k2]);
                    atomicAdd(gp + 0, float(gx));
                    atomicAdd(gp + 1, float(gy));
                    atomicAdd(gp + 3, float(-(gx * x + gy * y) * iw));
                }
            }
        });
    };
    if (TABLE) {
        __shared__ uint16_t stacks[(kAaBlock / 64) * kPairStack];
        for_pairs_masked(masks, (batch * hw + 63) / 64, group, stacks + (threadIdx.x >> 6) * kPairStack, work);
    } else {
        __shared__ PairList L;
        for_pairs_detected(rast, batch * hw, hw, height, width, L, work);
    }
}

// The table-free kernels have a lane per pixel in their detect phase and are held to 64 VGPRs (8 waves per SIMD; the analysis
// spills a little); the masked kernels have no such phase -- few waves, all of them in the analysis -- and take the registers
// the analysis wants.
#define TSAMD_AA_ARGS                                                                                                                       \
    const float *color, const float4 *rast, const float4 *pos, const double2 *windows, const unsigned long long *masks, const uint8_t *flags, \
        const int32_t *tri, const int32_t *opp, int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width, int channels, int group
#define TSAMD_AA_PASS color, rast, pos, windows, masks, flags, tri, opp, batch, n_vertices, n_tri, height, width, channels, group

__global__ __launch_bounds__(kAaBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void antialias_kernel(TSAMD_AA_ARGS, float *out)
{
    antialias_body<false>(TSAMD_AA_PASS, out);
}
__global__ __launch_bounds__(kAaBlock) void antialias_masked_kernel(TSAMD_AA_ARGS, float *out) { antialias_body<true>(TSAMD_AA_PASS, out); }

__global__ __launch_bounds__(kAaBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void antialias_backward_kernel(TSAMD_AA_ARGS, const float *grad_out, float boost,
                                                                                                                float *grad_color, float4 *grad_pos)
{
    antialias_backward_body<false>(TSAMD_AA_PASS, grad_out, boost, grad_color, grad_pos);
}
__global__ __launch_bounds__(kAaBlock) void antialias_backward_masked_kernel(TSAMD_AA_ARGS, const float *grad_out, float boost, float *grad_color, float4 *grad_pos)
{
    antialias_backward_body<true>(TSAMD_AA_PASS, grad_out, boost, grad_color, grad_pos);
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

// prepared = [batch * n_vertices] double2 window coordinates | two 64-bit pair masks per 64 pixels | [batch * n_tri] edge flags,
// every part 256-byte aligned
struct Prepared {
    int64_t masks, flags, bytes;
};

static Prepared prepared_layout(int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width)
{
    const int64_t pixels = batch * int64_t(height) * width;
    Prepared p;
    p.masks = (batch * n_vertices * 16 + 255) / 256 * 256;
    p.flags = p.masks + ((pixels + 63) / 64 * 16 + 255) / 256 * 256;
    p.bytes = p.flags + (batch * n_tri + 255) / 256 * 256;
    return p;
}

int64_t pair_masks_bytes(int64_t batch, int height, int width) { return (batch * int64_t(height) * width + 63) / 64 * 16; }

int64_t antialias_prepared_bytes(int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width)
{
    return prepared_layout(batch, n_vertices, n_tri, height, width).bytes;
}

hipError_t launch_antialias_prepare(const float *rast, const float *pos_clip, const int32_t *tri, const int32_t *opp, const void *pair_masks, int64_t batch,
                                    int64_t n_vertices, int64_t n_tri, int height, int width, void *prepared, hipStream_t stream)
{
    const int64_t n = batch * n_vertices, pixels = batch * int64_t(height) * width;
    const Prepared lay = prepared_layout(batch, n_vertices, n_tri, height, width);
    char *base = static_cast<char *>(prepared);
    hipError_t e;
    if (n > 0) {
        hipLaunchKernelGGL(antialias_windows_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, reinterpret_cast<const float4 *>(pos_clip), n, double(width),
                           double(height), reinterpret_cast<double2 *>(base));
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (batch * n_tri > 0) {
        hipLaunchKernelGGL(antialias_edge_flags_kernel, dim3(blocks_for(batch * n_tri)), dim3(256), 0, stream, reinterpret_cast<const double2 *>(base), tri, opp,
                           batch, n_vertices, n_tri, reinterpret_cast<uint8_t *>(base + lay.flags));
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (pixels <= 0) return hipSuccess;
    if (pair_masks)   // the rasteriser found them while it resolved the image
        return hipMemcpyAsync(base + lay.masks, pair_masks, size_t(pair_masks_bytes(batch, height, width)), hipMemcpyDeviceToDevice, stream);
    hipLaunchKernelGGL(antialias_detect_kernel, dim3(blocks_for(pixels)), dim3(kAaBlock), 0, stream, reinterpret_cast<const float4 *>(rast), pixels,
                       int64_t(height) * width, height, width, reinterpret_cast<unsigned long long *>(base + lay.masks));
    return hipGetLastError();
}

hipError_t launch_antialias(const float *color, const float *rast, const float *pos_clip, const void *prepared, const int32_t *tri, const int32_t *opp,
                            int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width, int channels, float *out, hipStream_t stream)
{
    const int64_t pixels = batch * int64_t(height) * width;
    if (pixels <= 0 || channels <= 0) return hipSuccess;
    hipError_t e = hipMemcpyAsync(out, color, size_t(pixels) * size_t(channels) * sizeof(float), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    if (n_tri <= 0) return hipSuccess;
    const Prepared lay = prepared_layout(batch, n_vertices, n_tri, height, width);
    const unsigned long long *masks = prepared ? reinterpret_cast<const unsigned long long *>(static_cast<const char *>(prepared) + lay.masks) : nullptr;
    const uint8_t *flags = prepared ? reinterpret_cast<const uint8_t *>(static_cast<const char *>(prepared) + lay.flags) : nullptr;
    const int group = masked_group((pixels + 63) / 64);
    const int64_t masked_waves = ((pixels + 63) / 64 + group - 1) / group;
    hipLaunchKernelGGL(prepared ? antialias_masked_kernel : antialias_kernel, dim3(prepared ? blocks_for(masked_waves * 64) : blocks_for(pixels)), dim3(kAaBlock), 0, stream, color, reinterpret_cast<const float4 *>(rast),
                       reinterpret_cast<const float4 *>(pos_clip), static_cast<const double2 *>(prepared), masks, flags, tri, opp, batch, n_vertices, n_tri, height, width,
                       channels, group, out);
    return hipGetLastError();
}

hipError_t launch_antialias_backward(const float *color, const float *rast, const float *pos_clip, const void *prepared, const int32_t *tri,
                                     const int32_t *opp, int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width, int channels, const float *grad_out, float boost,
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
    const Prepared lay = prepared_layout(batch, n_vertices, n_tri, height, width);
    const unsigned long long *masks = prepared ? reinterpret_cast<const unsigned long long *>(static_cast<const char *>(prepared) + lay.masks) : nullptr;
    const uint8_t *flags = prepared ? reinterpret_cast<const uint8_t *>(static_cast<const char *>(prepared) + lay.flags) : nullptr;
    const int group = masked_group((pixels + 63) / 64);
    const int64_t masked_waves = ((pixels + 63) / 64 + group - 1) / group;
    hipLaunchKernelGGL(prepared ? antialias_backward_masked_kernel : antialias_backward_kernel, dim3(prepared ? blocks_for(masked_waves * 64) : blocks_for(pixels)), dim3(kAaBlock), 0, stream, color, reinterpret_cast<const float4 *>(rast),
                       reinterpret_cast<const float4 *>(pos_clip), static_cast<const double2 *>(prepared), masks, flags, tri, opp, batch, n_vertices, n_tri, height, width,
                       channels, group, grad_out, boost, grad_color,
                       reinterpret_cast<float4 *>(grad_pos));
    return hipGetLastError();
}

}  // namespace tsamd
