// gfx950 (MI355X / CDNA4) STREAMING tile kernel for the tet-sphere geometry energy (experimental, round 3).
//
// One workgroup of 1 024 threads sweeps one TUBE band by band (stream_plan.h).  Its four wave groups (4 waves = 256
// lanes = one band slot per lane) each run ONE stage, on four different bands, between two workgroup barriers:
//
//   group 0   step s:  prefetch band s+1 (11 planes, entering vertices -> position ring), pass 1 on band s
//                      F = Ds Dm^-1 -> F ring, det / penalty -> factor ring          <- cusparseSpMV(G,x) + cuda_forward_det
//   group 1   pass 2 on band s-2:  H = L F from the F ring (bands s-3 .. s-1) -> H ring, 1/2 |H|^2   <- SpMV(GTLTLG) + Sdot
//   group 2   pass 3 on band s-4:  Q = L^T H from the H ring, + (c2 / c1) pen' cof F, d = P Dm^-T -> force ring
//                                                                                     <- SpMV(GTLTLG) + cuda_backward_det
//   group 3   vertex sums of band s-5: per (vertex, band) the incident forces -> 12-byte accumulator per live vertex;
//             a vertex' last band writes the sum out (exclusive -> grad, shared with another tube -> staging row)
//                                                                                     <- cusparseSpMV(TRANSPOSE, G)
// (reference: /root/reference/tssplat_ext/tet_spheres/tet_spheres_cuda.cu:118-263.)  The stages' LDS and VALU bursts overlap
// by construction -- the blob kernel alternates them (profiles/r03_issue_model.md) -- and only a tube's SIDES need halo
// slots.  Same arithmetic per slot as kernels.hip (same record layout, same gather formulas), so results agree with it
// to rounding; the summation ORDER of a vertex' forces differs (per band, then across bands).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stream_kernels.h"

namespace tsamd {
namespace {

constexpr int kWave = 64;
constexpr int kGroup = 256;                       // lanes per stage group
constexpr uint32_t kBandBytes = 48u * kBand;      // one band of 48-byte records
constexpr uint32_t kDBandBytes = 48u * (kBand + 1);   // + the all-zero record the incidence padding points at
constexpr uint32_t F_BASE = 0, H_BASE = F_BASE + kFRing * kBandBytes, D_BASE = H_BASE + kHRing * kBandBytes,
                   XS_BASE = D_BASE + kDRing * kDBandBytes;
static_assert(kFRing == 4 && kHRing == 4 && kDRing == 2, "ring indices are computed with & 3 / & 1");
static_assert(XS_BASE % 16 == 0, "position ring must be 16-byte aligned");

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
template <class T>
__device__ __forceinline__ GLOBAL_AS T *as_global(T *p)
{
    return (GLOBAL_AS T *)p;
}
template <class T>
__device__ __forceinline__ LDS_AS T *lds_at(uint32_t byte_addr)
{
    return (LDS_AS T *)(uintptr_t)byte_addr;   // the kernel's only LDS object is the dynamic array at LDS address 0 (checked on the host)
}

__device__ __forceinline__ float det3(const float *F)
{
    return -F[2] * F[4] * F[6] + F[1] * F[5] * F[6] + F[2] * F[3] * F[7] - F[0] * F[5] * F[7] - F[1] * F[3] * F[8] + F[0] * F[4] * F[8];
}
__device__ __forceinline__ void cof3(const float *F, float *C)
{
    C[0] = F[4] * F[8] - F[5] * F[7];
    C[1] = F[5] * F[6] - F[3] * F[8];
    C[2] = F[3] * F[7] - F[4] * F[6];
    C[3] = F[2] * F[7] - F[1] * F[8];
    C[4] = F[0] * F[8] - F[2] * F[6];
    C[5] = F[1] * F[6] - F[0] * F[7];
    C[6] = F[1] * F[5] - F[2] * F[4];
    C[7] = F[2] * F[3] - F[0] * F[5];
    C[8] = F[0] * F[4] - F[1] * F[3];
}

// 48-byte record = [tail quad | entries 0..3 | entries 4..7]; the ninth entry sits in the tail quad at a dword that rotates
// with the lane (kernels.hip: record tokens).  `t` = byte address of the ninth entry.
__device__ __forceinline__ uint32_t rec_addr(uint32_t band_base, uint32_t lane) { return band_base + 48u * lane + ((lane >> 1) & 12u); }
struct Mat9 {
    v2f p01, p23, p45, p67;
    float p8;
};
__device__ __forceinline__ Mat9 load_rec(uint32_t t)
{
    const uint32_t s = t & ~15u;
    const v4f a = *lds_at<const v4f>(s + 16), b = *lds_at<const v4f>(s + 32);
    Mat9 m;
    m.p01 = a.xy; m.p23 = a.zw; m.p45 = b.xy; m.p67 = b.zw;
    m.p8 = *lds_at<const float>(t);
    return m;
}
__device__ __forceinline__ void store_rec(uint32_t t, const float *m)
{
    const uint32_t s = t & ~15u;
    *lds_at<v4f>(s + 16) = v4f{m[0], m[1], m[2], m[3]};
    *lds_at<v4f>(s + 32) = v4f{m[4], m[5], m[6], m[7]};
    *lds_at<float>(t) = m[8];
}
__device__ __forceinline__ void sub9(Mat9 &acc, const Mat9 &g)
{
    acc.p01 -= g.p01; acc.p23 -= g.p23; acc.p45 -= g.p45; acc.p67 -= g.p67;
    acc.p8 -= g.p8;
}

// F of one slot from the position ring (byte offsets into it) and the slot's Dm^-1
__device__ __forceinline__ void slot_F(uint32_t xs, uint32_t w0, uint32_t w1, const float *dm, float *F)
{
    const v4u r0 = *lds_at<const v4u>(xs + (w0 & 0x7fffu)), r1 = *lds_at<const v4u>(xs + (w0 >> 16)), r2 = *lds_at<const v4u>(xs + (w1 & 0xffffu)),
              r3 = *lds_at<const v4u>(xs + (w1 >> 16));
    asm volatile("" : : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
    const float x0 = __uint_as_float(r0.x), y0 = __uint_as_float(r0.y), z0 = __uint_as_float(r0.z);
    const float Ds[9] = {__uint_as_float(r1.x) - x0, __uint_as_float(r2.x) - x0, __uint_as_float(r3.x) - x0,
                         __uint_as_float(r1.y) - y0, __uint_as_float(r2.y) - y0, __uint_as_float(r3.y) - y0,
                         __uint_as_float(r1.z) - z0, __uint_as_float(r2.z) - z0, __uint_as_float(r3.z) - z0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) F[3 * i + j] = Ds[3 * i + 0] * dm[j] + Ds[3 * i + 1] * dm[3 + j] + Ds[3 * i + 2] * dm[6 + j];
}

__device__ __forceinline__ float wave_sum(float v)
{
#define TSAMD_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
    v += TSAMD_DPP(v, 0xB1);
    v += TSAMD_DPP(v, 0x4E);
    v += TSAMD_DPP(v, 0x124);
    v += TSAMD_DPP(v, 0x128);
#undef TSAMD_DPP
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)),
                r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

struct StreamKernelArgs {
    const StreamTubeDesc *tubes;
    const uint8_t *blob;
    const float *x;
    const float *grad_out;
    float *grad;
    float *stage;
    double *partials;
    float c1, c2;
    int order;
    int n_tubes;
    int tubes_per_xcd;
};

// experiments: -DTSAMD_STREAM_SKIP=mask switches stage bodies off (bit 0: pass 1 compute, 1: pass 2, 2: pass 3, 3: vertex sums;
// results are wrong while a bit is set) -- prices the stages.
#ifndef TSAMD_STREAM_SKIP
#define TSAMD_STREAM_SKIP 0
#endif

struct BandInfo {   // StreamBandDesc unpacked
    uint32_t planes_off, enter_off, pairs_off, chunks_off, n_slots, n_owned, n_enter, n_pairs;
};

// record address of a gathered neighbour: token = (band delta + 1) << 8 | lane, ring of four bands
__device__ __forceinline__ uint32_t nbr_addr(uint32_t ring_base, uint32_t band, uint32_t tok)
{
    const uint32_t r = (band + 3u + ((tok >> 8) & 3u)) & 3u;   // (band + delta) mod 4
    return rec_addr(ring_base + r * kBandBytes, tok & 255u);
}

template <bool WITH_GRAD>
__global__ __launch_bounds__(1024, 4) void stream_tube_kernel(const StreamKernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tube = xcd * a.tubes_per_xcd + jb;
    if (jb >= a.tubes_per_xcd || tube >= a.n_tubes) return;
    const StreamTubeDesc td = a.tubes[tube];
    const int tid = threadIdx.x, group = tid >> 8, lane = tid & (kGroup - 1);
    const int nb = td.n_bands;
    const uint32_t VR16 = 16u * uint32_t((td.n_vslots + 3) & ~3);
    const uint32_t ACC_BASE = XS_BASE + VR16, SCAL_BASE = ACC_BASE + VR16, RED_BASE = SCAL_BASE + kScalRing * kBand * 4u,
                   DESC_BASE = RED_BASE + 64u;
    const auto g_blob = as_global(a.blob) + td.blob_off;
    const auto g_x = as_global(a.x);

    // the tube's band descriptors -> LDS (every stage reads the descriptor of its band every step: an LDS broadcast read
    // instead of a dependent scalar load from HBM at the head of the step)
    for (int w = tid; w < nb * 6; w += 1024) lds_at<uint32_t>(DESC_BASE)[w] = reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob)[w];
    // the all-zero force records the incidence padding points at
    if (tid < 2 * 12) lds_at<float>(D_BASE + uint32_t(tid / 12) * kDBandBytes + 48u * kBand)[tid % 12] = 0.f;
    __syncthreads();
    auto band_desc = [&](int b) {
        const LDS_AS uint32_t *w = lds_at<const uint32_t>(DESC_BASE + uint32_t(b) * 24u);
        BandInfo d;
        d.planes_off = w[0];
        d.enter_off = w[1];
        d.pairs_off = w[2];
        d.chunks_off = w[3];
        d.n_slots = w[4] & 0xffffu;
        d.n_owned = w[4] >> 16;
        d.n_enter = w[5] & 0xffffu;
        d.n_pairs = w[5] >> 16;
        return d;
    };
    auto plane = [&](const BandInfo &d, int q) { return reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + d.planes_off)[q * kBand + lane]; };
    auto planef = [&](const BandInfo &d, int q) { return reinterpret_cast<const GLOBAL_AS float *>(g_blob + d.planes_off)[q * kBand + lane]; };

    const float k_c1 = a.c1, k_c2 = a.c2;
    const float ratio = k_c2 / k_c1;
    const bool factored = k_c1 != 0.f && __builtin_fabsf(ratio) <= 0x1p+40f;
    const float s_pen = factored ? ratio : k_c2, out_scale = factored ? k_c1 : 1.f, q_scale = factored ? 1.f : k_c1;

    float e_acc = 0.f;   // group 0: sum of penalties; group 1: sum of 1/2 |H|^2
    const int n_steps = nb + (WITH_GRAD ? kLagSum : kLagP2);
    // Every group runs the same step loop (one barrier per step), from s = -2: the two leading steps only prefetch.  Each
    // group keeps the global data of its NEXT band (or next two) in registers, requested a step (or two) ahead, so that no
    // step starts with a dependent memory latency.

    if (group == 0) {
        // ---- stream + pass 1 ----
        uint32_t c_lv01 = 0, c_lv23 = 0;
        float c_dm[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) c_dm[c] = 0.f;
        uint32_t c_nslots = 0;
        int32_t e1_slot = -1, e1_vid = 0;   // this lane's entering vertex of band s + 1 (its list entry was loaded a step ago)
        for (int s = -2; s < n_steps; ++s) {
            // (a) list entry of band s + 2
            int32_t e2_slot = -1, e2_vid = 0;
            if (s + 2 < nb) {
                const BandInfo d2 = band_desc(s + 2);
                if (uint32_t(lane) < d2.n_enter) {
                    const v2u ev = reinterpret_cast<const GLOBAL_AS v2u *>(g_blob + d2.enter_off)[lane];
                    e2_slot = int32_t(ev.x);
                    e2_vid = int32_t(ev.y);
                }
            }
            // (b) planes of band s + 1 and the position of this lane's entering vertex of band s + 1
            uint32_t n_lv01 = 0, n_lv23 = 0;
            float n_dm[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) n_dm[c] = 0.f;
            float px = 0.f, py = 0.f, pz = 0.f;
            const bool have_next = s + 1 >= 0 && s + 1 < nb;
            BandInfo dn = {};
            if (have_next) {
                dn = band_desc(s + 1);
                n_lv01 = plane(dn, 0);
                n_lv23 = plane(dn, 1);
#pragma unroll
                for (int c = 0; c < 9; ++c) n_dm[c] = planef(dn, 4 + c);
                if (e1_slot >= 0) {
                    const size_t gv = size_t(e1_vid) * 3;
                    px = g_x[gv];
                    py = g_x[gv + 1];
                    pz = g_x[gv + 2];
                }
            }
            // (c) pass 1 on band s (lanes beyond the band's slots have nothing to do: nobody reads their records)
            if (!(TSAMD_STREAM_SKIP & 1) && s >= 0 && s < nb && uint32_t(lane) < c_nslots) {
                float F[9];
                slot_F(XS_BASE, c_lv01, c_lv23, c_dm, F);
                float scal = 0.f;
                if (c_lv01 & kOwnedBit) {
                    const float J = det3(F);
                    const float Jm = fmaxf(-J, 0.f);
                    float pen = 0.f, dpen = 0.f;
                    if (a.order == 2) {
                        pen = Jm * Jm;
                        dpen = -2.f * Jm;
                    } else if (a.order == 4) {
                        pen = Jm * Jm * Jm * Jm;
                        dpen = -4.f * Jm * Jm * Jm;
                    }
                    e_acc += pen;
                    scal = s_pen * dpen;
                }
                store_rec(rec_addr(F_BASE + uint32_t(s & 3) * kBandBytes, uint32_t(lane)), F);
                if (WITH_GRAD) lds_at<float>(SCAL_BASE + uint32_t(s % kScalRing) * kBand * 4u)[lane] = scal;
            }
            // (d) entering vertices of band s + 1: position -> ring slot, accumulator = 0 (visible after this step's barrier)
            if (have_next) {
                if (e1_slot >= 0) {
                    *lds_at<v4f>(XS_BASE + 16u * uint32_t(e1_slot)) = v4f{px, py, pz, 0.f};
                    *lds_at<v4f>(ACC_BASE + 16u * uint32_t(e1_slot)) = v4f{0.f, 0.f, 0.f, 0.f};
                }
                for (int q = lane + kGroup; q < int(dn.n_enter); q += kGroup) {   // more entering vertices than lanes (rare)
                    const v2u ev = reinterpret_cast<const GLOBAL_AS v2u *>(g_blob + dn.enter_off)[q];
                    const size_t gv = size_t(ev.y) * 3;
                    *lds_at<v4f>(XS_BASE + 16u * ev.x) = v4f{g_x[gv], g_x[gv + 1], g_x[gv + 2], 0.f};
                    *lds_at<v4f>(ACC_BASE + 16u * ev.x) = v4f{0.f, 0.f, 0.f, 0.f};
                }
            }
            c_lv01 = n_lv01;
            c_lv23 = n_lv23;
            c_nslots = have_next ? dn.n_slots : 0u;
#pragma unroll
            for (int c = 0; c < 9; ++c) c_dm[c] = n_dm[c];
            e1_slot = e2_slot;
            e1_vid = e2_vid;
            __syncthreads();
        }
    } else if (group == 1) {
        // ---- pass 2 on band s - 2 ----
        uint32_t c_n01 = 0, c_n23 = 0, c_nslots = 0;
        for (int s = -2; s < n_steps; ++s) {
            const int b = s - kLagP2;
            uint32_t n_n01 = 0, n_n23 = 0, n_nslots = 0;
            if (b + 1 >= 0 && b + 1 < nb) {
                const BandInfo dn = band_desc(b + 1);
                n_n01 = plane(dn, 2);
                n_n23 = plane(dn, 3);
                n_nslots = dn.n_slots;
            }
            if (!(TSAMD_STREAM_SKIP & 2) && b >= 0 && b < nb && uint32_t(lane) < c_nslots) {
                const uint32_t n01 = c_n01, n23 = c_n23;
                float H[9];
                if (n01 & kOwnedBit) {
                    const uint32_t t0 = nbr_addr(F_BASE, uint32_t(b), n01 & 0x3ffu), t1 = nbr_addr(F_BASE, uint32_t(b), (n01 >> 16) & 0x3ffu),
                                   t2 = nbr_addr(F_BASE, uint32_t(b), n23 & 0x3ffu), t3 = nbr_addr(F_BASE, uint32_t(b), (n23 >> 16) & 0x3ffu);
                    Mat9 h = load_rec(rec_addr(F_BASE + uint32_t(b & 3) * kBandBytes, uint32_t(lane)));
                    const Mat9 g0 = load_rec(t0), g1 = load_rec(t1), g2 = load_rec(t2), g3 = load_rec(t3);
                    h.p01 *= 4.f; h.p23 *= 4.f; h.p45 *= 4.f; h.p67 *= 4.f;
                    h.p8 *= 4.f;
                    sub9(h, g0);
                    sub9(h, g1);
                    sub9(h, g2);
                    sub9(h, g3);
                    v2f sq = h.p01 * h.p01;
                    sq = __builtin_elementwise_fma(h.p23, h.p23, sq);
                    sq = __builtin_elementwise_fma(h.p45, h.p45, sq);
                    sq = __builtin_elementwise_fma(h.p67, h.p67, sq);
                    e_acc += 0.5f * (sq.x + sq.y + h.p8 * h.p8);
                    H[0] = h.p01.x; H[1] = h.p01.y; H[2] = h.p23.x; H[3] = h.p23.y;
                    H[4] = h.p45.x; H[5] = h.p45.y; H[6] = h.p67.x; H[7] = h.p67.y;
                    H[8] = h.p8;
                } else {
#pragma unroll
                    for (int c = 0; c < 9; ++c) H[c] = 0.f;
                }
                if (WITH_GRAD) store_rec(rec_addr(H_BASE + uint32_t(b & 3) * kBandBytes, uint32_t(lane)), H);
            }
            c_n01 = n_n01;
            c_n23 = n_n23;
            c_nslots = n_nslots;
            __syncthreads();
        }
    } else if (group == 2) {
        // ---- pass 3 on band s - 4 ----
        uint32_t c_pl[4] = {0, 0, 0, 0}, c_nslots = 0;
        float c_dm[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) c_dm[c] = 0.f;
        for (int s = -2; s < n_steps; ++s) {
            const int b = s - kLagP3;
            uint32_t n_pl[4] = {0, 0, 0, 0}, n_nslots = 0;
            float n_dm[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) n_dm[c] = 0.f;
            if (WITH_GRAD && b + 1 >= 0 && b + 1 < nb) {
                const BandInfo dn = band_desc(b + 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) n_pl[q] = plane(dn, q);
#pragma unroll
                for (int c = 0; c < 9; ++c) n_dm[c] = planef(dn, 4 + c);
                n_nslots = dn.n_slots;
            }
            if (WITH_GRAD && !(TSAMD_STREAM_SKIP & 4) && b >= 0 && b < nb && uint32_t(lane) < c_nslots) {
                const uint32_t n01 = c_pl[2], n23 = c_pl[3];
                const float scal = lds_at<const float>(SCAL_BASE + uint32_t(b % kScalRing) * kBand * 4u)[lane];
                const uint32_t t0 = nbr_addr(H_BASE, uint32_t(b), n01 & 0x3ffu), t1 = nbr_addr(H_BASE, uint32_t(b), (n01 >> 16) & 0x3ffu),
                               t2 = nbr_addr(H_BASE, uint32_t(b), n23 & 0x3ffu), t3 = nbr_addr(H_BASE, uint32_t(b), (n23 >> 16) & 0x3ffu);
                Mat9 q = load_rec(rec_addr(H_BASE + uint32_t(b & 3) * kBandBytes, uint32_t(lane)));
                const Mat9 g0 = load_rec(t0), g1 = load_rec(t1), g2 = load_rec(t2), g3 = load_rec(t3);
                q.p01 *= 4.f; q.p23 *= 4.f; q.p45 *= 4.f; q.p67 *= 4.f;
                q.p8 *= 4.f;
                sub9(q, g0);
                sub9(q, g1);
                sub9(q, g2);
                sub9(q, g3);
                float P[9] = {q.p01.x, q.p01.y, q.p23.x, q.p23.y, q.p45.x, q.p45.y, q.p67.x, q.p67.y, q.p8};
                if (!factored) {
#pragma unroll
                    for (int c = 0; c < 9; ++c) P[c] *= q_scale;
                }
                if (scal != 0.f) {   // inverted owned tet: F again, from the position ring
                    float F[9], C[9];
                    slot_F(XS_BASE, c_pl[0], c_pl[1], c_dm, F);
                    cof3(F, C);
#pragma unroll
                    for (int c = 0; c < 9; ++c) P[c] += scal * C[c];
                }
                float D[9];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 3; ++i) D[3 * k + i] = P[3 * i + 0] * c_dm[3 * k + 0] + P[3 * i + 1] * c_dm[3 * k + 1] + P[3 * i + 2] * c_dm[3 * k + 2];
                const uint32_t r = D_BASE + uint32_t(b & 1) * kDBandBytes + 48u * uint32_t(lane);
                *lds_at<v4f>(r) = v4f{-(D[0] + D[3] + D[6]), -(D[1] + D[4] + D[7]), -(D[2] + D[5] + D[8]), D[0]};
                *lds_at<v4f>(r + 16) = v4f{D[1], D[2], D[3], D[4]};
                *lds_at<v4f>(r + 32) = v4f{D[5], D[6], D[7], D[8]};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) c_pl[q] = n_pl[q];
#pragma unroll
            for (int c = 0; c < 9; ++c) c_dm[c] = n_dm[c];
            c_nslots = n_nslots;
            __syncthreads();
        }
    } else {
        // ---- vertex sums of band s - 5 ----
        const float gscale = (a.grad_out ? *as_global(a.grad_out) : 1.f) * out_scale;
        const auto g_grad = as_global(a.grad);
        const auto g_stage = as_global(a.stage);
        // this lane's pair of band b (p0) with its first two chunks (ch0), of band b + 1 (p1: its chunks are requested now),
        // of band b + 2 (requested now)
        uint32_t p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0};
        v2u ch0[2] = {v2u{0, 0}, v2u{0, 0}};
        bool has0 = false, has1 = false;
        constexpr uint32_t kPadChunk = uint32_t(kBand << 2) * 0x10001u;
        for (int s = -2; s < n_steps; ++s) {
            const int b = s - kLagSum;
            uint32_t p2[3] = {0, 0, 0};
            bool has2 = false;
            v2u ch1[2] = {v2u{kPadChunk, kPadChunk}, v2u{kPadChunk, kPadChunk}};
            if (WITH_GRAD && b + 2 >= 0 && b + 2 < nb) {
                const BandInfo d2 = band_desc(b + 2);
                if (uint32_t(lane) < d2.n_pairs) {
                    const GLOBAL_AS uint32_t *pr = reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + d2.pairs_off) + 3 * size_t(lane);
                    p2[0] = pr[0];
                    p2[1] = pr[1];
                    p2[2] = pr[2];
                    has2 = true;
                }
            }
            if (WITH_GRAD && has1) {
                const BandInfo d1 = band_desc(b + 1);
                const uint32_t nch = (p1[0] >> 16) & 0x3fffu;
                const GLOBAL_AS v2u *ch = reinterpret_cast<const GLOBAL_AS v2u *>(g_blob + d1.chunks_off) + p1[1];
                ch1[0] = ch[0];                       // (every pair has at least one chunk)
                if (nch > 1) ch1[1] = ch[1];
            }
            if (WITH_GRAD && !(TSAMD_STREAM_SKIP & 8) && b >= 0 && b < nb) {
                const BandInfo d = band_desc(b);
                const uint32_t dbase = D_BASE + uint32_t(b & 1) * kDBandBytes;
                auto gather4 = [&](const v2u wv, float &gx, float &gy, float &gz) {
                    const uint32_t ent[4] = {wv.x & 0xffffu, wv.x >> 16, wv.y & 0xffffu, wv.y >> 16};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const LDS_AS float *f = lds_at<const float>(dbase + ent[q] * 12u);
                        gx += f[0];
                        gy += f[1];
                        gz += f[2];
                    }
                };
                auto finish_pair = [&](uint32_t w0, int32_t out_row, float gx, float gy, float gz) {
                    const uint32_t vslot = w0 & 0xffffu, flags = w0 >> 16;
                    const uint32_t aa = ACC_BASE + 16u * vslot;
                    v4f acc = *lds_at<v4f>(aa);
                    acc.x += gx;
                    acc.y += gy;
                    acc.z += gz;
                    if (flags & kPairLast) {
                        const bool shared = (flags & kPairShared) != 0;
                        GLOBAL_AS float *dst = (shared ? g_stage : g_grad) + size_t(out_row) * 3;
                        const float sc = shared ? out_scale : gscale;
                        dst[0] = acc.x * sc;
                        dst[1] = acc.y * sc;
                        dst[2] = acc.z * sc;
                    } else {
                        *lds_at<v4f>(aa) = acc;
                    }
                };
                if (has0) {
                    const uint32_t nch = (p0[0] >> 16) & 0x3fffu;
                    float gx = 0.f, gy = 0.f, gz = 0.f;
                    gather4(ch0[0], gx, gy, gz);
                    gather4(ch0[1], gx, gy, gz);
                    if (nch > 2) {   // longer lists: the remaining chunks straight from memory
                        const GLOBAL_AS v2u *ch = reinterpret_cast<const GLOBAL_AS v2u *>(g_blob + d.chunks_off) + p0[1];
                        for (uint32_t c = 2; c < nch; ++c) gather4(ch[c], gx, gy, gz);
                    }
                    finish_pair(p0[0], int32_t(p0[2]), gx, gy, gz);
                }
                for (int p = lane + kGroup; p < int(d.n_pairs); p += kGroup) {   // more pairs than lanes (rare)
                    const GLOBAL_AS uint32_t *pr = reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + d.pairs_off) + 3 * size_t(p);
                    const uint32_t w0 = pr[0], first = pr[1];
                    const int32_t out_row = int32_t(pr[2]);
                    const uint32_t nch = (w0 >> 16) & 0x3fffu;
                    const GLOBAL_AS v2u *ch = reinterpret_cast<const GLOBAL_AS v2u *>(g_blob + d.chunks_off) + first;
                    float gx = 0.f, gy = 0.f, gz = 0.f;
                    for (uint32_t c = 0; c < nch; ++c) gather4(ch[c], gx, gy, gz);
                    finish_pair(w0, out_row, gx, gy, gz);
                }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                p0[q] = p1[q];
                p1[q] = p2[q];
            }
            ch0[0] = ch1[0];
            ch0[1] = ch1[1];
            has0 = has1;
            has1 = has2;
            __syncthreads();
        }
    }

    // ---- energy partials of the tube: group 0 holds the penalties, group 1 the smoothness terms ----
    if (group < 2) {
        const float ws = wave_sum(e_acc);
        if ((tid & (kWave - 1)) == 0) lds_at<double>(RED_BASE)[tid / kWave] = double(ws);
    }
    __syncthreads();
    if (tid == 0) {
        const LDS_AS double *red = lds_at<const double>(RED_BASE);
        a.partials[2 * size_t(tube)] = (red[4] + red[5]) + (red[6] + red[7]);       // E_s
        a.partials[2 * size_t(tube) + 1] = (red[0] + red[1]) + (red[2] + red[3]);   // E_b
    }
}

}  // namespace

int32_t stream_lds_bytes(int32_t max_vslots, int32_t max_bands)
{
    const uint32_t vr16 = 16u * uint32_t((max_vslots + 3) & ~3);
    return int32_t(XS_BASE + 2 * vr16 + kScalRing * kBand * 4u + 64u + 24u * uint32_t(max_bands) + 16u);
}

hipError_t configure_stream_kernels(int lds_bytes)
{
    static int configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds_bytes <= configured[dev]) return hipSuccess;
    const void *fns[] = {reinterpret_cast<const void *>(&stream_tube_kernel<true>), reinterpret_cast<const void *>(&stream_tube_kernel<false>)};
    for (const void *fn : fns) {
        hipFuncAttributes attr;
        hipError_t e = hipFuncGetAttributes(&attr, fn);
        if (e != hipSuccess) return e;
        if (attr.sharedSizeBytes != 0) return hipErrorInvalidDeviceFunction;   // absolute LDS addressing needs the dynamic array at address 0
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
    }
    configured[dev] = lds_bytes;
    return hipSuccess;
}

hipError_t launch_stream_eval(const StreamEvalArgs &e, hipStream_t stream, hipEvent_t *ev)
{
    hipError_t err;
    if (ev && (err = hipEventRecord(ev[0], stream)) != hipSuccess) return err;
    if (e.n_tubes > 0) {
        StreamKernelArgs k;
        k.tubes = e.tubes;
        k.blob = e.blob;
        k.x = e.x;
        k.grad_out = e.grad_out;
        k.grad = e.grad;
        k.stage = e.stage;
        k.partials = e.partials;
        k.c1 = e.c1;
        k.c2 = e.c2;
        k.order = e.order;
        k.n_tubes = int(e.n_tubes);
        k.tubes_per_xcd = int((e.n_tubes + 7) / 8);
        const dim3 grid(unsigned(8 * k.tubes_per_xcd)), block(1024);
        if (e.grad)
            hipLaunchKernelGGL(stream_tube_kernel<true>, grid, block, size_t(e.lds_bytes), stream, k);
        else
            hipLaunchKernelGGL(stream_tube_kernel<false>, grid, block, size_t(e.lds_bytes), stream, k);
        if ((err = hipGetLastError()) != hipSuccess) return err;
    }
    if (ev && (err = hipEventRecord(ev[1], stream)) != hipSuccess) return err;
    err = launch_finish(e.fin_vid, e.fin_off, e.grad ? e.n_finish : 0, e.stage, e.grad, e.grad_out, e.partials, e.n_tubes, e.c1, e.c2,
                        e.energy, e.terms, stream);
    if (err != hipSuccess) return err;
    if (ev && (err = hipEventRecord(ev[2], stream)) != hipSuccess) return err;
    return hipSuccess;
}

}  // namespace tsamd
