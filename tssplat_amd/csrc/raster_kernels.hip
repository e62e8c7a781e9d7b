// gfx950 kernels of the renderer slice: rasterize + interpolate, the two nvdiffrast operators
// /root/reference/renderers/mesh_rasterizer.py:103,117,145,153 calls: forward rasterisation (triangle id, perspective-correct
// barycentrics, z/w) and its backward, interpolate forward and backward.  (antialias: aa_kernels.hip.)
//
// The specification these kernels implement is oracle/raster_oracle.py (a restatement of nvdiffrast's published
// algorithm with every open choice fixed there); coverage and the depth test repeat its float64 / integer operations
// one by one, which is what makes the triangle ids bit-exact against it.
//
//   rasterize_snap_kernel  one lane per (view, vertex): window coordinates in 1/256 pixel + z/w in float64, 16 B
//   rasterize_bin_kernel   one lane per (view, triangle): integer edge functions over the pixel centres of the bounding
//                          box in one wave-uniform loop, depth from a float32 plane, covered fragments into a per-wave
//                          LDS queue that is drained in batches: read of the depth image, then (rarely) a 64-bit
//                          atomicMin of (depth32 << 32 | triangle)
//   rasterize_clip_kernel  the rare path: triangles with a vertex at w <= 0, clipped against the near plane, same integer rules
//   rasterize_resolve_kernel  one lane per pixel: winning triangle -> (u, v, z/w, id + 1); optional by-product: the pair masks
//                          (which pixels differ from their right / upper neighbour) that the antialias kernels start from
//   interpolate_kernel / interpolate_backward_kernel  one lane per pixel
//
// Bound: the depth image -- L2 atomics and the reads in front of them (pricing builds, profiles/r04_raster_experiments.md: walk
// 0.46 ms, fragment queue + depth reads 0.23, atomics 0.31 of 1.10 ms on the 512-sphere surface); 12 B of indices + 3 x 16 B of
// snapped vertices per triangle and view, 8 B of key and 16 B of output per pixel; no MFMA anywhere.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "raster.h"

// Coverage, depth and silhouette decisions repeat the oracle's operations one by one: a multiply and an add must round
// separately.  The __fmul_rn / __dmul_rn family are plain operators in this toolchain's headers, so this file is compiled with
// -ffp-contract=off (tssplat_amd/_build.py: SOURCE_FLAGS); tests/test_build_metadata.py checks the flag is in place.

namespace tsamd {
namespace {

constexpr int kSubBits = 8;
constexpr long long kSub = 1ll << kSubBits;
constexpr long long kCoordLimit = 1ll << 22;
constexpr unsigned long long kNoFragment = 0xFFFFFFFFFFFFFFFFull;

// One snapped vertex of one view, 16 bytes: window coordinates in 1/256 pixel (|x|, |y| <= 2^22) and z/w rounded to float32.
// x == kDropped: the vertex is not finite, behind the eye (w <= 0) or out of the coordinate range -- its triangles are dropped.
struct SnapRec {
    int32_t x, y;
    float zw;
    uint32_t pad;
};
static_assert(sizeof(SnapRec) == 16, "SnapRec is loaded as one 16-byte word");
constexpr int32_t kDropped = INT32_MIN;

// oracle/raster_oracle.py::snap_vertices, operation by operation (explicitly rounded double intrinsics: no contraction).
// One lane per (view, vertex): a vertex is snapped once per view, not once per triangle that uses it (six on a closed
// surface).
// (x, y, z, w) in float64 -> record; false (and x = kDropped) when w <= 0 or the window coordinates leave the coordinate range
__device__ __forceinline__ bool snap_point(double x, double y, double z, double w, bool finite, double width, double height, SnapRec &r)
{
    bool ok = finite && w > 0.0;
    const double ws = ok ? w : 1.0;
    const double xs = __dmul_rn(__dadd_rn(__dmul_rn(__ddiv_rn(x, ws), 0.5), 0.5), width);
    const double ys = __dmul_rn(__dadd_rn(__dmul_rn(__ddiv_rn(y, ws), 0.5), 0.5), height);
    const double X = floor(__dadd_rn(__dmul_rn(xs, double(kSub)), 0.5));
    const double Y = floor(__dadd_rn(__dmul_rn(ys, double(kSub)), 0.5));
    ok = ok && fabs(X) <= double(kCoordLimit) && fabs(Y) <= double(kCoordLimit);
    r.x = ok ? int32_t(X) : kDropped;
    r.y = ok ? int32_t(Y) : 0;
    r.zw = ok ? float(__ddiv_rn(z, ws)) : 0.f;
    r.pad = 0;
    return ok;
}

// pad = kBehind marks a FINITE vertex at w <= 0: its triangles are not dropped but clipped against the near plane
// (rasterize_clip_kernel); view_flags[b] != 0 tells that kernel that view b has such a vertex at all.
constexpr uint32_t kBehind = 1u;

__global__ __launch_bounds__(256) void rasterize_snap_kernel(const float4 *pos, int64_t n, int64_t n_vertices, double width, double height, SnapRec *out,
                                                             uint32_t *view_flags)
{
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const float4 p = pos[gid];
    const bool finite = isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && isfinite(p.w);
    SnapRec r;
    snap_point(double(p.x), double(p.y), double(p.z), double(p.w), finite, width, height, r);
    if (finite && !(p.w > 0.f)) {
        r.pad = kBehind;
        view_flags[gid / n_vertices] = 1u;   // (every writer writes the same value)
    }
    out[gid] = r;
}

__device__ __forceinline__ bool top_left(int32_t dx, int32_t dy) { return dy < 0 || (dy == 0 && dx < 0); }

// ---- fragment queue: one per wave, in LDS ----
// The depth test of a fragment is a read of the depth image followed (rarely) by an atomic: a memory round trip that must not
// sit inside the pixel walk.  Covered fragments are therefore appended to a small per-wave queue (key + pixel: 12 bytes) --
// the slot comes from the wave's ballot, no atomic -- and the queue is drained whenever it could overflow and at the end,
// three entries per lane with their three reads in flight together (a fourth costs the eighth wave per SIMD: 70 instead of 64 VGPRs).
constexpr int kQueue = 320;            // entries per wave: drained as soon as fewer than 64 are free
constexpr int kWavesPerBlock = 4;

struct WaveQueue {
    unsigned long long *keys;          // [kQueue], this wave's slice of LDS
    uint32_t *pixels;                  // [kQueue]
    unsigned long long *image;         // the view's depth keys
    int count;                         // wave-uniform
};

// read-before-atomic on the depth image: the key only ever decreases, so any value read earlier is an upper bound of the
// current one -- a fragment that does not beat it cannot win and skips the device-scope atomic (a pixel of a scene with depth
// complexity d sees ~ln d improvements)
__device__ __forceinline__ void drain(WaveQueue &q, int lane)
{
    for (int base = 0; base < q.count; base += 3 * 64) {
        unsigned long long key[3], cur[3];
        unsigned long long *slot[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = base + 64 * j + lane;
            const bool have = k < q.count;
            key[j] = have ? q.keys[have ? k : 0] : kNoFragment;
            slot[j] = q.image + (have ? q.pixels[k] : 0u);   // (an empty lane reads pixel 0 of its view, which always exists: the load stays unconditional)
        }
#pragma unroll
        // workgroup scope (served by the L2, not fetched device-coherently): the value is only a FILTER in front of the atomic, and
        // a stale one is an older = larger key -- it can let a needless atomic through, never hold back a needed one.  1.11 -> 0.99 ms
        // on the 512-sphere surface.
        for (int j = 0; j < 3; ++j) cur[j] = __hip_atomic_load(slot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (key[j] < cur[j]) atomicMin(slot[j], key[j]);              // (an empty lane's key is all ones: never smaller)
    }
    q.count = 0;
}

struct DepthPlane {   // z/w = z0 + zx (x - x0) + zy (y - y0), float32
    float z0, zx, zy;
    int32_t x0, y0;
};

// The pixel walk of a wave: every lane steps through the pixel centres of its triangle's bounding box, one candidate per
// iteration of ONE wave-uniform loop (not a loop nest per lane: the wave runs max(candidates) iterations, not max(rows) x
// max(columns)).  Edge functions E_k (edge opposite vertex k, >= 0 inside) in 32-bit integers when the triangle is small enough
// for every value to fit (|edge vector| < 2^14 sub-pixel units = 64 pixels: |E| < 2^30), in 64-bit integers otherwise -- the same
// integers either way.  The top-left rule is folded into the start values (an edge that does not own its boundary starts one
// lower), so "inside" is one sign test of the three values OR-ed together.
template <class Int>
__device__ __forceinline__ void wave_walk(bool mine, int32_t px0, int32_t px1, int32_t py0, int32_t py1, Int r0, Int r1, Int r2, Int sx0, Int sx1, Int sx2,
                                          Int sy0, Int sy1, Int sy2, const DepthPlane pl, unsigned long long tid, int width, int lane, WaveQueue &q)
{
    const int32_t hs = int32_t(kSub / 2);
    int32_t px = px0, py = py0;
    Int e0 = r0, e1 = r1, e2 = r2;
    bool active = mine;
    while (__ballot(active) != 0ull) {
        bool in = active && (e0 | e1 | e2) >= 0;
        unsigned long long key = 0;
        if (in) {
            // oracle/raster_oracle.py::rasterize_ids, operation by operation (this file is compiled without contraction)
            const float zw = (pl.z0 + pl.zx * float(px * int32_t(kSub) + hs - pl.x0)) + pl.zy * float(py * int32_t(kSub) + hs - pl.y0);
            in = zw >= -1.f && zw <= 1.f;
            const float s = (zw + 1.f) * 2147483648.f;                       // in [0, 2^32]: the scaling is exact
            const uint32_t depth = s >= 4294967296.f ? 0xFFFFFFFFu : uint32_t(s);
            key = ((unsigned long long)depth << 32) | tid;
        }
        const unsigned long long mask = __ballot(in);
        if (mask != 0ull) {
            if (in) {
                const int at = q.count + int(__builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)));
                q.keys[at] = key;
                q.pixels[at] = uint32_t(py) * uint32_t(width) + uint32_t(px);
            }
            q.count += __popcll(mask);
            if (q.count > kQueue - 64) drain(q, lane);
        }
        if (active) {
            ++px;
            e0 -= sx0;
            e1 -= sx1;
            e2 -= sx2;
            if (px > px1) {
                px = px0;
                ++py;
                r0 += sy0;
                r1 += sy1;
                r2 += sy2;
                e0 = r0;
                e1 = r1;
                e2 = r2;
                active = py <= py1;
            }
        }
    }
}

// One lane per (view, triangle); a workgroup is four independent waves (no barrier anywhere).
__global__ __launch_bounds__(64 * kWavesPerBlock) void rasterize_bin_kernel(const SnapRec *snapped, const int32_t *tri, int64_t n_vertices, int64_t n_tri,
                                                                             int blocks_per_view, int height, int width, unsigned long long *keys, int n_views)
{
    __shared__ unsigned long long queue_keys[kWavesPerBlock * kQueue];
    __shared__ uint32_t queue_pixels[kWavesPerBlock * kQueue];
    // n_views > 0: XCD-affine order -- workgroup i lands on XCD i % 8 (observed; speed only), so view v is given to the workgroups
    // with i % 8 == v % 8 and its depth image lives in ONE L2 instead of migrating between eight on every atomic.  Pays when many
    // fragments fight for a pixel (512-sphere surface, 41 per pixel: 0.99 -> 0.91 ms); costs a little balance otherwise (the launcher decides).
    int b, jb;
    if (n_views > 0) {
        const int xcd = int(blockIdx.x) & 7, jq = int(blockIdx.x) >> 3;
        b = (jq / blocks_per_view) * 8 + xcd;
        jb = jq % blocks_per_view;
        if (b >= n_views) return;
    } else {
        b = int(blockIdx.x) / blocks_per_view;
        jb = int(blockIdx.x) - b * blocks_per_view;
    }
    const int64_t t_own = int64_t(jb) * (64 * kWavesPerBlock) + threadIdx.x;
    const int lane = int(threadIdx.x) & 63, wave = int(threadIdx.x) >> 6;
    WaveQueue q;
    q.keys = queue_keys + wave * kQueue;
    q.pixels = queue_pixels + wave * kQueue;
    q.image = keys + size_t(b) * size_t(height) * size_t(width);
    q.count = 0;

    // ---- triangle fetch ----
    const int32_t hs = int32_t(kSub / 2);
    const int64_t t = t_own;
    bool valid = t < n_tri;
    int32_t i0 = 0, i1 = 0, i2 = 0;
    if (valid) {
        i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        valid = !(i0 < 0 || i1 < 0 || i2 < 0 || i0 >= n_vertices || i1 >= n_vertices || i2 >= n_vertices);
    }
    SnapRec s0 = {}, s1 = {}, s2 = {};
    if (valid) {
        const SnapRec *sv = snapped + int64_t(b) * n_vertices;
        s0 = sv[i0], s1 = sv[i1], s2 = sv[i2];
        valid = !(s0.x == kDropped || s1.x == kDropped || s2.x == kDropped);
    }
    // ---- triangle set-up ----
    long long area = 0;
    int32_t px0 = 0, px1 = -1, py0 = 0, py1 = -1;
    if (valid) {
        area = (long long)(s1.x - s0.x) * (s2.y - s0.y) - (long long)(s1.y - s0.y) * (s2.x - s0.x);
        if (area < 0) {   // orient counter-clockwise
            const SnapRec tmp = s1;
            s1 = s2;
            s2 = tmp;
        }
        const int32_t minx = min(s0.x, min(s1.x, s2.x)), maxx = max(s0.x, max(s1.x, s2.x));
        const int32_t miny = min(s0.y, min(s1.y, s2.y)), maxy = max(s0.y, max(s1.y, s2.y));
        // pixel i has its centre at i * 256 + 128; >> is an arithmetic shift (floor) on negative values
        px0 = max(0, (minx - hs + int32_t(kSub) - 1) >> kSubBits), px1 = min(width - 1, (maxx - hs) >> kSubBits);
        py0 = max(0, (miny - hs + int32_t(kSub) - 1) >> kSubBits), py1 = min(height - 1, (maxy - hs) >> kSubBits);
        valid = area != 0 && px0 <= px1 && py0 <= py1;   // no pixel centre inside the bounding box: nothing to do
    }
    // depth plane through the snapped vertices, float32, every operation rounded on its own (oracle/raster_oracle.py::rasterize_ids)
    DepthPlane pl;
    {
        const float A = float(double(area < 0 ? -area : area));
        const float d1 = s1.zw - s0.zw, d2 = s2.zw - s0.zw;
        pl.zx = (d1 * float(s2.y - s0.y) - d2 * float(s1.y - s0.y)) / A;
        pl.zy = (d2 * float(s1.x - s0.x) - d1 * float(s2.x - s0.x)) / A;
        pl.z0 = s0.zw;
        pl.x0 = s0.x;
        pl.y0 = s0.y;
    }
    // E_k: d/dx = -(dy) * 256 per pixel, d/dy = dx * 256; an edge that is not top-left starts one lower
    const int32_t dx0 = s2.x - s1.x, dy0 = s2.y - s1.y, dx1 = s0.x - s2.x, dy1 = s0.y - s2.y, dx2 = s1.x - s0.x, dy2 = s1.y - s0.y;
    const int32_t b0 = top_left(dx0, dy0) ? 0 : 1, b1 = top_left(dx1, dy1) ? 0 : 1, b2 = top_left(dx2, dy2) ? 0 : 1;
    const int32_t cx0 = px0 * int32_t(kSub) + hs, cy0 = py0 * int32_t(kSub) + hs;
    const int32_t reach = max(max(abs(dx0), abs(dy0)), max(max(abs(dx1), abs(dy1)), max(abs(dx2), abs(dy2))));
    const bool small = reach < (1 << 14);
    // |c - v| <= reach + 256 for every pixel centre of the bounding box, so |E| < 2 * 2^14 * (2^14 + 2^8) < 2^30 for a small triangle
    wave_walk<int32_t>(valid && small, px0, px1, py0, py1, dx0 * (cy0 - s1.y) - dy0 * (cx0 - s1.x) - b0, dx1 * (cy0 - s2.y) - dy1 * (cx0 - s2.x) - b1,
                       dx2 * (cy0 - s0.y) - dy2 * (cx0 - s0.x) - b2, dy0 * int32_t(kSub), dy1 * int32_t(kSub), dy2 * int32_t(kSub), dx0 * int32_t(kSub),
                       dx1 * int32_t(kSub), dx2 * int32_t(kSub), pl, (unsigned long long)t, width, lane, q);
    if (__ballot(valid && !small) != 0ull)
        wave_walk<long long>(valid && !small, px0, px1, py0, py1, (long long)dx0 * (cy0 - s1.y) - (long long)dy0 * (cx0 - s1.x) - b0,
                             (long long)dx1 * (cy0 - s2.y) - (long long)dy1 * (cx0 - s2.x) - b1,
                             (long long)dx2 * (cy0 - s0.y) - (long long)dy2 * (cx0 - s0.x) - b2, (long long)dy0 * kSub, (long long)dy1 * kSub,
                             (long long)dy2 * kSub, (long long)dx0 * kSub, (long long)dx1 * kSub, (long long)dx2 * kSub, pl, (unsigned long long)t, width,
                             lane, q);
    drain(q, lane);
}

// ---- near-plane clipping (rare path) ----
// oracle/raster_oracle.py::clip_near + rasterize_ids: a triangle with finite vertices of which some (not all) lie at w <= 0 is
// clipped against z + w >= 0 in float64 (Sutherland-Hodgman, intersections computed from the inside vertex towards the outside
// one), the polygon's new vertices are snapped like any vertex, and the fan is rasterised with the same integer edge functions,
// the same float32 depth plane and the triangle's own id.  Only the views whose flag the snap kernel raised are visited -- no view
// of an object inside the frustum -- with plain per-lane loops and direct atomics: correctness path, not speed.
__device__ __forceinline__ void cover_clipped(SnapRec s0, SnapRec s1, SnapRec s2, unsigned long long tid, int height, int width, unsigned long long *image)
{
    const int32_t hs = int32_t(kSub / 2);
    long long area = (long long)(s1.x - s0.x) * (s2.y - s0.y) - (long long)(s1.y - s0.y) * (s2.x - s0.x);
    if (area == 0) return;
    if (area < 0) {
        const SnapRec tmp = s1;
        s1 = s2;
        s2 = tmp;
        area = -area;
    }
    const int32_t minx = min(s0.x, min(s1.x, s2.x)), maxx = max(s0.x, max(s1.x, s2.x));
    const int32_t miny = min(s0.y, min(s1.y, s2.y)), maxy = max(s0.y, max(s1.y, s2.y));
    const int32_t px0 = max(0, (minx - hs + int32_t(kSub) - 1) >> kSubBits), px1 = min(width - 1, (maxx - hs) >> kSubBits);
    const int32_t py0 = max(0, (miny - hs + int32_t(kSub) - 1) >> kSubBits), py1 = min(height - 1, (maxy - hs) >> kSubBits);
    if (px0 > px1 || py0 > py1) return;
    const float A = float(double(area));
    const float d1 = s1.zw - s0.zw, d2 = s2.zw - s0.zw;
    const float zx = (d1 * float(s2.y - s0.y) - d2 * float(s1.y - s0.y)) / A;
    const float zy = (d2 * float(s1.x - s0.x) - d1 * float(s2.x - s0.x)) / A;
    const long long dx0 = s2.x - s1.x, dy0 = s2.y - s1.y, dx1 = s0.x - s2.x, dy1 = s0.y - s2.y, dx2 = s1.x - s0.x, dy2 = s1.y - s0.y;
    const long long b0 = top_left(int32_t(dx0), int32_t(dy0)) ? 0 : 1, b1 = top_left(int32_t(dx1), int32_t(dy1)) ? 0 : 1,
                    b2 = top_left(int32_t(dx2), int32_t(dy2)) ? 0 : 1;
    for (int32_t py = py0; py <= py1; ++py)
        for (int32_t px = px0; px <= px1; ++px) {
            const long long cx = (long long)px * kSub + hs, cy = (long long)py * kSub + hs;
            const long long e0 = dx0 * (cy - s1.y) - dy0 * (cx - s1.x) - b0, e1 = dx1 * (cy - s2.y) - dy1 * (cx - s2.x) - b1,
                            e2 = dx2 * (cy - s0.y) - dy2 * (cx - s0.x) - b2;
            if ((e0 | e1 | e2) < 0) continue;
            const float zw = (s0.zw + zx * float(int32_t(cx) - s0.x)) + zy * float(int32_t(cy) - s0.y);
            if (!(zw >= -1.f && zw <= 1.f)) continue;
            const float q = (zw + 1.f) * 2147483648.f;
            const uint32_t depth = q >= 4294967296.f ? 0xFFFFFFFFu : uint32_t(q);
            atomicMin(image + size_t(py) * size_t(width) + size_t(px), ((unsigned long long)depth << 32) | tid);
        }
}

__device__ __forceinline__ void clip_one(const float4 *pos, const SnapRec *snapped, const int32_t *tri, int b, int64_t t, int64_t n_vertices, int height,
                                         int width, unsigned long long *keys)
{
    const int32_t idx[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0 || idx[0] >= n_vertices || idx[1] >= n_vertices || idx[2] >= n_vertices) return;
    const SnapRec *sv = snapped + int64_t(b) * n_vertices;
    const SnapRec s[3] = {sv[idx[0]], sv[idx[1]], sv[idx[2]]};
    bool any_behind = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (s[k].x == kDropped && s[k].pad != kBehind) return;   // not finite, or beyond the guard band: dropped, not clipped
        any_behind = any_behind || s[k].pad == kBehind;
    }
    if (!any_behind) return;   // (rasterize_bin_kernel has done this triangle)
    const float4 *pv = pos + int64_t(b) * n_vertices;
    double p[3][4], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 v = pv[idx[k]];
        p[k][0] = double(v.x), p[k][1] = double(v.y), p[k][2] = double(v.z), p[k][3] = double(v.w);
        d[k] = __dadd_rn(p[k][2], p[k][3]);
    }
    SnapRec poly[4];
    int n_poly = 0;
    bool fail = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = (k + 1) % 3;
        const bool in_k = d[k] >= 0.0, in_n = d[n] >= 0.0;
        if (in_k) {
            fail = fail || s[k].x == kDropped;   // inside the near plane and yet at w <= 0: not a projection this slice handles
            poly[n_poly++] = s[k];
        }
        if (in_k != in_n) {
            const int a = in_k ? k : n, o = in_k ? n : k;
            const double tt = __ddiv_rn(d[a], __dsub_rn(d[a], d[o]));
            double q[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) q[c] = __dadd_rn(p[a][c], __dmul_rn(tt, __dsub_rn(p[o][c], p[a][c])));
            SnapRec r;
            const bool finite = isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]) && isfinite(q[3]);
            fail = !snap_point(q[0], q[1], q[2], q[3], finite, double(width), double(height), r) || fail;
            poly[n_poly++] = r;
        }
    }
    if (fail || n_poly < 3) return;
    unsigned long long *image = keys + size_t(b) * size_t(height) * size_t(width);
    for (int k = 1; k + 1 < n_poly; ++k) cover_clipped(poly[0], poly[k], poly[k + 1], (unsigned long long)t, height, width, image);
}

// a small fixed grid strides over the triangles of the flagged views only: a batch that needs no clipping costs one flag read per view
constexpr int kClipBlocks = 256;   // (a workgroup costs ~25 ns to dispatch: 1024 of them were 28 us for a batch that needs no clipping)

__global__ __launch_bounds__(256) void rasterize_clip_kernel(const float4 *pos, const SnapRec *snapped, const int32_t *tri, const uint32_t *view_flags,
                                                             int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width, unsigned long long *keys)
{
    const int lane = int(threadIdx.x) & 63;
    for (int64_t base = 0; base < batch; base += 64) {   // the flags of 64 views in one load; only flagged views are visited
        unsigned long long flagged = __ballot(base + lane < batch && view_flags[base + lane] != 0u);
        while (flagged != 0ull) {
            const int64_t b = base + __builtin_amdgcn_readfirstlane(__ffsll((long long)flagged) - 1);
            flagged &= flagged - 1ull;
            for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < n_tri; t += int64_t(gridDim.x) * 256)
                clip_one(pos, snapped, tri, int(b), t, n_vertices, height, width, keys);
        }
    }
}

// A wave holds 64 consecutive pixels of the flattened [view, row, column] image -- what the pair masks are indexed by, and the
// order in which keys are read and `rast` is written as one stream.  (Four rows x 64 columns per workgroup, the upper neighbour
// handed over in LDS, was tried: the strided streams alone cost 138 -> 172 us on 120 views x 512^2.)
constexpr int kResolvePixels = 4;   // 64-pixel chunks per wave, all their key loads in flight together

__global__ __launch_bounds__(256) void rasterize_resolve_kernel(const float4 *pos, const int32_t *tri, int64_t n_vertices, int64_t batch,
                                                                int height, int width, const unsigned long long *keys, float4 *rast,
                                                                unsigned long long *pair_masks)
{
    constexpr int K = kResolvePixels;
    const int64_t hw = int64_t(height) * width, total = batch * hw;
    const int lane = int(threadIdx.x) & 63;
    // pixel k of this lane: chunk (first chunk of the wave + k), lane-th pixel -- every load of a wave is one contiguous 512 bytes
    const int64_t gid0 = (int64_t(blockIdx.x) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))) * (64 * K);
    int64_t gid[K], b[K];
    int px[K], py[K];
    bool have[K];
    // view / row / column without a per-lane 64-bit division: the view of the wave's first pixel on the scalar unit (a wave crosses
    // at most one view boundary unless the image has fewer than 64 K pixels)
    const int64_t b0 = hw >= 64 * K ? gid0 / hw : 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        gid[k] = gid0 + 64 * k + lane;
        have[k] = gid[k] < total;
        uint32_t pix;
        if (hw >= 64 * K) {
            const int64_t off = gid[k] - b0 * hw;
            b[k] = off >= hw ? b0 + 1 : b0;
            pix = uint32_t(off >= hw ? off - hw : off);
        } else {
            b[k] = gid[k] / hw;
            pix = uint32_t(gid[k] - b[k] * hw);
        }
        py[k] = int(pix / uint32_t(width));   // (pix < 2^26: height, width <= 8192)
        px[k] = int(pix - uint32_t(py[k]) * uint32_t(width));
    }
    // all global loads first, together: the kernel is bound by the latency of its key loads (a second round trip behind the first
    // costs as much again)
    unsigned long long key[K], key_up[K], key_right = 0ull;
    bool want_up[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        want_up[k] = pair_masks && have[k] && py[k] + 1 < height;
        key[k] = have[k] ? keys[gid[k]] : kNoFragment;
        key_up[k] = want_up[k] ? keys[gid[k] + width] : 0ull;
    }
    // (the right neighbour of a chunk's last pixel is the next chunk's first: a lane of this wave, except behind the last chunk)
    const bool want_right = pair_masks && have[K - 1] && px[K - 1] + 1 < width && lane == 63;
    if (want_right) key_right = keys[gid[K - 1] + 1];
    if (pair_masks) {
        // by-product for tsamd_antialias_prepare: does the pixel's triangle differ from its right / upper neighbour's?  Two 64-bit
        // words per 64 pixels, what antialias_detect_kernel would otherwise find by reading the whole `rast` image back.
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t id = uint32_t(key[k]);
            const uint32_t next = uint32_t(__shfl_down(int(id), 1));   // (every lane takes part: lane 62 reads lane 63)
            uint32_t beyond = id;                                       // lane 63's right neighbour
            if (k + 1 < K)
                beyond = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(key[k + 1 < K ? k + 1 : k]))));
            else if (want_right)
                beyond = uint32_t(key_right);
            const uint32_t right = lane == 63 ? beyond : next;
            const uint32_t up = want_up[k] ? uint32_t(key_up[k]) : id;
            const bool c0 = have[k] && px[k] + 1 < width && right != id;
            const bool c1 = have[k] && py[k] + 1 < height && up != id;
            const unsigned long long m0 = __ballot(c0), m1 = __ballot(c1);
            if (lane == 0 && have[k]) {
                pair_masks[2 * (gid[k] >> 6)] = m0;
                pair_masks[2 * (gid[k] >> 6) + 1] = m1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!have[k]) continue;
        if (key[k] == kNoFragment) {
            rast[gid[k]] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const int64_t t = int64_t(key[k] & 0xFFFFFFFFull);
        const float4 *pv = pos + b[k] * n_vertices;
        const float4 v0 = pv[tri[3 * t]], v1 = pv[tri[3 * t + 1]], v2 = pv[tri[3 * t + 2]];
        const float fx = (float(px[k]) + 0.5f) / float(width) * 2.f - 1.f, fy = (float(py[k]) + 0.5f) / float(height) * 2.f - 1.f;
        const float p0x = v0.x - fx * v0.w, p0y = v0.y - fy * v0.w;
        const float p1x = v1.x - fx * v1.w, p1y = v1.y - fy * v1.w;
        const float p2x = v2.x - fx * v2.w, p2y = v2.y - fy * v2.w;
        const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
        float s = a0 + a1 + a2;
        s = s == 0.f ? 1.f : s;
        const float b0f = a0 / s, b1f = a1 / s, b2f = 1.f - b0f - b1f;
        const float z = b0f * v0.z + b1f * v1.z + b2f * v2.z;
        float w = b0f * v0.w + b1f * v1.w + b2f * v2.w;
        w = w == 0.f ? 1.f : w;
        rast[gid[k]] = make_float4(fminf(fmaxf(b0f, 0.f), 1.f), fminf(fmaxf(b1f, 0.f), 1.f), fminf(fmaxf(z / w, -1.f), 1.f), float(t + 1));
    }
}

// ---- scatter with runs ----
// The backward scatters add one value per pixel to the three vertices of the pixel's triangle.  Consecutive pixels of a row
// mostly belong to the same triangle (a 41 k-tet object at 512^2: runs of ~5), so the lanes of a run are summed inside the wave
// first -- a segmented scan over equal keys -- and only the last lane of every run issues the atomics.  A wave without any run
// (a dense scene of sub-pixel triangles) skips the scan.  EVERY lane of the wave must call these (no early return before).
struct Runs {
    int start;      // lane index of the first lane of this lane's run
    bool last;      // this lane ends its run: it holds the run's sums after run_sum()
    bool any;       // wave-uniform: some run is longer than one lane
};

__device__ __forceinline__ Runs find_runs(long long key)
{
    const int lane = int(threadIdx.x) & 63;
    const long long prev = __shfl_up(key, 1);
    const bool head = lane == 0 || key != prev;
    const unsigned long long heads = __ballot(head);
    Runs r;
    r.start = 63 - __clzll(heads & (~0ull >> (63 - lane)));
    r.last = lane == 63 || ((heads >> (lane + 1)) & 1ull) != 0ull;
    r.any = heads != ~0ull;
    return r;
}

__device__ __forceinline__ float run_sum(const Runs &r, float v)
{
    if (!r.any) return v;
    const int lane = int(threadIdx.x) & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float up = __shfl_up(v, d);
        if (lane - d >= r.start) v += up;
    }
    return v;
}

// d (u, v) / d clip-space positions of the winning triangle (oracle/raster_oracle.py::rasterize_backward): the barycentrics
// are ratios of the homogeneous edge functions of `resolve`, so this is their quotient rule; z/w and the id carry no gradient
// (as in nvdiffrast), the clamps of the forward are treated as inactive.  fp32 atomics into grad_pos.
__global__ __launch_bounds__(256) void rasterize_backward_kernel(const float4 *pos, const int32_t *tri, int64_t n_vertices, int64_t n_tri, int64_t batch,
                                                                 int height, int width, const float4 *rast, const float4 *grad_rast, float4 *grad_pos)
{
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t hw = int64_t(height) * width;
    bool valid = gid < batch * hw;
    int64_t t = -1;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        t = int64_t(rast[gid].w) - 1;
        valid = t >= 0 && t < n_tri;
    }
    if (valid) {
        g = grad_rast[gid];
        valid = !(g.x == 0.f && g.y == 0.f);
    }
    const int64_t b = valid ? gid / hw : 0, pix = valid ? gid - b * hw : 0;
    int32_t i0 = 0, i1 = 0, i2 = 0;
    if (valid) {
        i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        valid = !(i0 < 0 || i1 < 0 || i2 < 0 || i0 >= n_vertices || i1 >= n_vertices || i2 >= n_vertices);
    }
    float d[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // (x, y, w) of the three vertices
    if (valid) {
        const int py = int(pix / width), px = int(pix - int64_t(py) * width);
        const float4 *pv = pos + b * n_vertices;
        const float4 v0 = pv[i0], v1 = pv[i1], v2 = pv[i2];
        const float fx = (float(px) + 0.5f) / float(width) * 2.f - 1.f, fy = (float(py) + 0.5f) / float(height) * 2.f - 1.f;
        const float p0x = v0.x - fx * v0.w, p0y = v0.y - fy * v0.w;
        const float p1x = v1.x - fx * v1.w, p1y = v1.y - fy * v1.w;
        const float p2x = v2.x - fx * v2.w, p2y = v2.y - fy * v2.w;
        const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
        float s = a0 + a1 + a2;
        s = s == 0.f ? 1.f : s;
        const float is = 1.f / s;
        const float u = a0 * is, v = a1 * is;
        const float dot = g.x * u + g.y * v;
        const float da0 = (g.x - dot) * is, da1 = (g.y - dot) * is, da2 = -dot * is;
        d[0] = da1 * -p2y + da2 * p1y, d[1] = da1 * p2x + da2 * -p1x;
        d[3] = da0 * p2y + da2 * -p0y, d[4] = da0 * -p2x + da2 * p0x;
        d[6] = da0 * -p1y + da1 * p0y, d[7] = da0 * p1x + da1 * -p0x;
        d[2] = -fx * d[0] - fy * d[1];
        d[5] = -fx * d[3] - fy * d[4];
        d[8] = -fx * d[6] - fy * d[7];
    }
    // lanes of one triangle (and view) in a row: one set of atomics per run
    const Runs runs = find_runs(valid ? b * (int64_t(1) << 32) + t : -1 - int64_t(threadIdx.x));
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = run_sum(runs, d[k]);
    if (valid && runs.last) {
        float *gp = reinterpret_cast<float *>(grad_pos + b * n_vertices);
        const int64_t at[3] = {4 * int64_t(i0), 4 * int64_t(i1), 4 * int64_t(i2)};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            atomicAdd(gp + at[k] + 0, d[3 * k]);
            atomicAdd(gp + at[k] + 1, d[3 * k + 1]);
            atomicAdd(gp + at[k] + 3, d[3 * k + 2]);
        }
    }
}

// Triangle named by a rast pixel's fourth channel (id + 1 as a float, exact up to 2^24 - 1 triangles: tsamd_rasterize
// refuses longer lists), or -1 for background / an id outside [0, n_tri) / a non-finite value.
__device__ __forceinline__ int64_t triangle_of(float w, int64_t n_tri)
{
    if (!(w >= 1.f) || !(w <= 16777216.f)) return -1;
    const int64_t t = int64_t(w) - 1;
    return t < n_tri ? t : -1;
}
// The three vertex indices of triangle t; false when one of them lies outside [0, n_vertices).
__device__ __forceinline__ bool triangle_indices(const int32_t *tri, int64_t t, int64_t n_vertices, int32_t &i0, int32_t &i1, int32_t &i2)
{
    i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    return uint64_t(i0) < uint64_t(n_vertices) && uint64_t(i1) < uint64_t(n_vertices) && uint64_t(i2) < uint64_t(n_vertices)
           && i0 >= 0 && i1 >= 0 && i2 >= 0;
}

__global__ __launch_bounds__(256) void interpolate_kernel(const float *attr, int64_t attr_batch, int64_t n_vertices, int channels,
                                                          const float4 *rast, const int32_t *tri, int64_t n_tri, int64_t batch, int64_t hw, float *out)
{
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= batch * hw) return;
    const float4 r = rast[gid];
    float *o = out + gid * channels;
    const int64_t t = triangle_of(r.w, n_tri);
    int32_t i0 = 0, i1 = 0, i2 = 0;
    // a pixel whose id names no triangle of THIS list, or a triangle with a vertex index outside the attribute array, is
    // background (what nvdiffrast does with them): the rast image may come from another triangle list, or from the caller
    if (t < 0 || !triangle_indices(tri, t, n_vertices, i0, i1, i2)) {
        for (int c = 0; c < channels; ++c) o[c] = 0.f;
        return;
    }
    const int64_t b = gid / hw;
    const float *a = attr + (attr_batch > 1 ? b : 0) * n_vertices * channels;
    const float *a0 = a + int64_t(i0) * channels, *a1 = a + int64_t(i1) * channels, *a2 = a + int64_t(i2) * channels;
    const float u = r.x, v = r.y, w = 1.f - r.x - r.y;
    for (int c = 0; c < channels; ++c) o[c] = u * a0[c] + v * a1[c] + w * a2[c];
}

// d out / d attr: scatter of the three barycentric weights, summed over runs of equal triangles inside the wave first (fp32
// global atomics: the order of the additions, and so the last bits of the result, vary from run to run -- like nvdiffrast's
// own backward); d out / d (u, v) per pixel.
__global__ __launch_bounds__(256) void interpolate_backward_kernel(const float *attr, int64_t attr_batch, int64_t n_vertices, int channels,
                                                                   const float4 *rast, const int32_t *tri, int64_t n_tri, int64_t batch, int64_t hw,
                                                                   const float *grad_out, float *grad_attr, float4 *grad_rast)
{
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool inside = gid < batch * hw;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) r = rast[gid];
    const int64_t t = triangle_of(r.w, n_tri);
    int32_t i0 = 0, i1 = 0, i2 = 0;
    const bool valid = inside && t >= 0 && triangle_indices(tri, t, n_vertices, i0, i1, i2);   // (else background: no reads, no atomics)
    if (inside && !valid && grad_rast) grad_rast[gid] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t b = valid ? gid / hw : 0;
    const int64_t base = (attr_batch > 1 ? b : 0) * n_vertices * channels;
    int64_t o0 = 0, o1 = 0, o2 = 0;
    if (valid) o0 = base + int64_t(i0) * channels, o1 = base + int64_t(i1) * channels, o2 = base + int64_t(i2) * channels;
    const float u = r.x, v = r.y, w = 1.f - r.x - r.y;
    // lanes of one triangle (and attribute batch) in a row: one set of atomics per run
    const Runs runs = find_runs(valid ? (attr_batch > 1 ? b : 0) * (int64_t(1) << 32) + t : -1 - int64_t(threadIdx.x));
    float du = 0.f, dv = 0.f;
    for (int c = 0; c < channels; ++c) {
        const float gc = valid ? grad_out[gid * channels + c] : 0.f;
        const float s0 = run_sum(runs, u * gc), s1 = run_sum(runs, v * gc), s2 = run_sum(runs, w * gc);
        if (valid) {
            if (runs.last) {
                atomicAdd(grad_attr + o0 + c, s0);
                atomicAdd(grad_attr + o1 + c, s1);
                atomicAdd(grad_attr + o2 + c, s2);
            }
            const float a2 = attr[o2 + c];
            du += gc * (attr[o0 + c] - a2);
            dv += gc * (attr[o1 + c] - a2);
        }
    }
    if (valid && grad_rast) grad_rast[gid] = make_float4(du, dv, 0.f, 0.f);
}

unsigned blocks_for(int64_t n) { return unsigned((n + 255) / 256); }

}  // namespace

hipError_t launch_rasterize(const float *pos_clip, int64_t batch, int64_t n_vertices, const int32_t *tri, int64_t n_tri, int height, int width,
                            void *workspace, float *rast, void *pair_masks, hipStream_t stream)
{
    const int64_t pixels = batch * int64_t(height) * width;
    if (pixels <= 0) return hipSuccess;
    unsigned long long *keys = static_cast<unsigned long long *>(workspace);
    SnapRec *snapped = reinterpret_cast<SnapRec *>(keys + ((size_t(pixels) + 1) & ~size_t(1)));   // 16-byte aligned
    uint32_t *view_flags = reinterpret_cast<uint32_t *>(snapped + size_t(batch) * size_t(n_vertices));
    hipError_t e = hipMemsetAsync(keys, 0xFF, size_t(pixels) * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    if (batch * n_tri > 0 && batch * n_vertices > 0) {
        if ((e = hipMemsetAsync(view_flags, 0, size_t(batch) * sizeof(uint32_t), stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(rasterize_snap_kernel, dim3(blocks_for(batch * n_vertices)), dim3(256), 0, stream, reinterpret_cast<const float4 *>(pos_clip),
                           batch * n_vertices, n_vertices, double(width), double(height), snapped, view_flags);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        const int64_t blocks_per_view = (n_tri + 255) / 256;   // (256 = 64 * kWavesPerBlock triangles per workgroup)
        // XCD-affine views (rasterize_bin_kernel) when the depth atomics are dense -- more triangles than pixels -- and the views
        // divide evenly among the eight XCDs
        const bool xcd_views = n_tri >= int64_t(height) * width && (batch % 8 == 0 || batch >= 64);
        const int64_t bin_blocks = (xcd_views ? (batch + 7) / 8 * 8 : batch) * blocks_per_view;
        hipLaunchKernelGGL(rasterize_bin_kernel, dim3(unsigned(bin_blocks)), dim3(256), 0, stream, snapped, tri, n_vertices, n_tri,
                           int(blocks_per_view), height, width, keys, xcd_views ? int(batch) : 0);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // triangles that straddle the eye plane (none in a view without a vertex at w <= 0)
        hipLaunchKernelGGL(rasterize_clip_kernel, dim3(unsigned(std::min<int64_t>(kClipBlocks, blocks_per_view))), dim3(256), 0, stream,
                           reinterpret_cast<const float4 *>(pos_clip), snapped, tri, view_flags, batch, n_vertices, n_tri, height, width, keys);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(rasterize_resolve_kernel, dim3(blocks_for((pixels + kResolvePixels - 1) / kResolvePixels)), dim3(256), 0, stream, reinterpret_cast<const float4 *>(pos_clip), tri,
                       n_vertices, batch, height, width, keys, reinterpret_cast<float4 *>(rast), static_cast<unsigned long long *>(pair_masks));
    return hipGetLastError();
}

hipError_t launch_rasterize_backward(const float *pos_clip, int64_t batch, int64_t n_vertices, const int32_t *tri, int64_t n_tri, int height, int width,
                                     const float *rast, const float *grad_rast, float *grad_pos, hipStream_t stream)
{
    if (batch * n_vertices > 0) {
        const hipError_t e = hipMemsetAsync(grad_pos, 0, size_t(batch) * size_t(n_vertices) * 4 * sizeof(float), stream);
        if (e != hipSuccess) return e;
    }
    const int64_t pixels = batch * int64_t(height) * width;
    if (pixels <= 0 || n_tri <= 0) return hipSuccess;
    hipLaunchKernelGGL(rasterize_backward_kernel, dim3(blocks_for(pixels)), dim3(256), 0, stream, reinterpret_cast<const float4 *>(pos_clip), tri, n_vertices,
                       n_tri, batch, height, width, reinterpret_cast<const float4 *>(rast), reinterpret_cast<const float4 *>(grad_rast),
                       reinterpret_cast<float4 *>(grad_pos));
    return hipGetLastError();
}

hipError_t launch_interpolate(const float *attr, int64_t attr_batch, int64_t n_vertices, int channels, const float *rast, const int32_t *tri,
                              int64_t n_tri, int64_t batch, int height, int width, float *out, hipStream_t stream)
{
    const int64_t hw = int64_t(height) * width;
    if (batch * hw <= 0) return hipSuccess;
    hipLaunchKernelGGL(interpolate_kernel, dim3(blocks_for(batch * hw)), dim3(256), 0, stream, attr, attr_batch, n_vertices, channels,
                       reinterpret_cast<const float4 *>(rast), tri, n_tri, batch, hw, out);
    return hipGetLastError();
}

hipError_t launch_interpolate_backward(const float *attr, int64_t attr_batch, int64_t n_vertices, int channels, const float *rast, const int32_t *tri,
                                       int64_t n_tri, int64_t batch, int height, int width, const float *grad_out, float *grad_attr, float *grad_rast,
                                       hipStream_t stream)
{
    const int64_t hw = int64_t(height) * width;
    hipError_t e = hipMemsetAsync(grad_attr, 0, size_t(attr_batch) * size_t(n_vertices) * size_t(channels) * sizeof(float), stream);
    if (e != hipSuccess) return e;
    if (batch * hw <= 0) return hipSuccess;
    hipLaunchKernelGGL(interpolate_backward_kernel, dim3(blocks_for(batch * hw)), dim3(256), 0, stream, attr, attr_batch, n_vertices, channels,
                       reinterpret_cast<const float4 *>(rast), tri, n_tri, batch, hw, grad_out, grad_attr, reinterpret_cast<float4 *>(grad_rast));
    return hipGetLastError();
}

}  // namespace tsamd
