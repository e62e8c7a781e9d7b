// gfx950 (MI355X / CDNA4) kernels for the tet-sphere geometry energy.
//
// One workgroup evaluates one *tile*: a cluster of up to ~3 800 tets (owned +
// one-ring face halo) whose deformation gradients F fit the CU's 160 KiB LDS.
// Per evaluation the tile's 13 dword planes (16-bit local indices + fp32
// Dm^-1, 52 B per slot) stream from HBM exactly once with perfectly coalesced
// 16 B/lane loads; F, L F, L^T L F and the vertex accumulation never leave the
// CU.  No MFMA: 3x3 algebra at ~4 flop/B is bandwidth bound.
//
// What each stage stands for in the reference
// (/root/reference/tssplat_ext/tet_spheres/tet_spheres_cuda.cu):
//   pass 1  F = Ds Dm^-1, det, penalty     <- cusparseSpMV(G,x) :167,:221 + cuda_forward_det :48-66
//   pass 2  H = L F, 1/2|H|^2              <- cusparseSpMV(GTLTLG,x) :131 + cublasSdot :154 (factored, no M)
//   pass 3  P = c1 L^T H + c2 dpen cof(F)  <- cusparseSpMV(GTLTLG,x) :216 + cuda_backward_det :68-102
//           d = P Dm^-T, per-vertex gather <- cusparseSpMV(TRANSPOSE, G) :248 (their transposed COO SpMV
//                                             scatters with atomics; LDS f32 atomics measured ~3 clk/lane
//                                             on gfx950, so each vertex sums its incident tets instead)
//   finish  sum shared-vertex partials, reduce energy, * grad_out
//                                          <- cublasSasum :185, host combine :191, cublasSscal :258
// There is no host synchronisation anywhere (the reference blocks three times
// per forward+backward: :154, :185, :257).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace tsamd {
namespace {

constexpr int kWave = 64;

// Scheduling fence between the four slots a lane processes in a pass (build with
// -DTSAMD_NO_SLOT_FENCE to let the compiler interleave them).
#ifdef TSAMD_NO_SLOT_FENCE
#define SLOT_FENCE() ((void)0)
#else
#define SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float det3(const float *F)
{
    // row-major F[3*i+j]; the six products of tet_spheres_cuda.cu:24-29
    return -F[2] * F[4] * F[6] + F[1] * F[5] * F[6] + F[2] * F[3] * F[7] - F[0] * F[5] * F[7] - F[1] * F[3] * F[8] +
           F[0] * F[4] * F[8];
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

__device__ __forceinline__ uint32_t comp(const uint4 &v, int p)
{
    return p == 0 ? v.x : (p == 1 ? v.y : (p == 2 ? v.z : v.w));
}
__device__ __forceinline__ float comp(const float4 &v, int p)
{
    return p == 0 ? v.x : (p == 1 ? v.y : (p == 2 ? v.z : v.w));
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// F of one slot from the staged positions and the slot's Dm^-1 (held in registers)
__device__ __forceinline__ void slot_F(const float4 *xs, uint32_t lv0, uint32_t lv1, uint32_t lv2, uint32_t lv3,
                                       const float4 *dm, int p, float *F)
{
    // xs holds (x, y, z, 0); reading it as 4 x u32 keeps the compiler from narrowing the access to
    // ds_read_b96, which costs 8 LDS cycles against 4 for ds_read_b128
    const v4u r0 = reinterpret_cast<const v4u *>(xs)[lv0], r1 = reinterpret_cast<const v4u *>(xs)[lv1],
              r2 = reinterpret_cast<const v4u *>(xs)[lv2], r3 = reinterpret_cast<const v4u *>(xs)[lv3];
    asm volatile("" : : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
    const float3 x0 = make_float3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
    const float3 x1 = make_float3(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z));
    const float3 x2 = make_float3(__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z));
    const float3 x3 = make_float3(__uint_as_float(r3.x), __uint_as_float(r3.y), __uint_as_float(r3.z));
    const float Ds[9] = {x1.x - x0.x, x2.x - x0.x, x3.x - x0.x, x1.y - x0.y, x2.y - x0.y,
                         x3.y - x0.y, x1.z - x0.z, x2.z - x0.z, x3.z - x0.z};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            F[3 * i + j] = Ds[3 * i + 0] * comp(dm[j], p) + Ds[3 * i + 1] * comp(dm[3 + j], p) +
                           Ds[3 * i + 2] * comp(dm[6 + j], p);
}

struct KernelArgs {
    const TileDesc *tiles;
    const uint8_t *blob;
    const int32_t *gvid;
    const float *x;
    const float *grad_out;
    float *grad;
    float *stage;
    double *partials;
    float c1, c2;
    int order;
    int n_tiles;
    int tiles_per_xcd;
    int dbg;  // ablation switches, honoured only by -DTSAMD_ABLATION builds (tools/ablate.py)
};

enum : int { DBG_LOCAL_GATHER3 = 2, DBG_LOCAL_GATHER2 = 4, DBG_SKIP_P3 = 8, DBG_SKIP_P2 = 16,
             DBG_EXIT_AFTER_P1 = 32, DBG_EXIT_AFTER_LOAD = 64, DBG_SKIP_VGATHER = 128, DBG_SKIP_OUT = 256 };

#ifdef TSAMD_ABLATION
#define DBG(flag) ((a.dbg & (flag)) != 0)
#else
#define DBG(flag) false
#endif

// One workgroup = one tile.  LDS map (SA = s_pad + 4 slots incl. the all-zero slot at index s_pad,
// VP = vertices rounded up to 4):
//   [0, 16 SA)        FA: float4 per slot  (F[0..3], later H, addressed by lds_index(slot))
//   [16 SA, 32 SA)    FB: float4 per slot  (F[4..7])
//   [32 SA, 36 SA)    FC: float  per slot  (F[8])
//   [36 SA, +16 VP)   xs: float4 per local vertex (staged positions)
//   [0, 48 SA)        DV: 4 planes (one per local tet vertex) of 3 floats per slot -- the per-tet
//                     vertex forces, written over F/H/xs once those are dead
//   then 256 B of reduction scratch.
// BLOCK only sets the VGPR budget (1024 threads = 4 waves/SIMD = 128 VGPRs).
template <bool WITH_GRAD, int BLOCK>
__global__ __launch_bounds__(BLOCK) void tile_energy_kernel(const KernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // XCD-aware tile order: workgroup b lands on XCD b % 8 (observed, speed only), so give each
    // XCD a contiguous run of tiles -- the tiles of one sphere then share one L2 and the halo
    // planes two neighbouring tiles both read are served from it.
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = xcd * a.tiles_per_xcd + j;
    if (j >= a.tiles_per_xcd || tile >= a.n_tiles) return;

    const TileDesc td = a.tiles[tile];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nq = td.s_pad >> 2;
    const int SA = td.s_pad + 4;
    const int VP = (td.n_verts + 3) & ~3;
    const uint32_t ZS = uint32_t(td.s_pad);

    float4 *FA = reinterpret_cast<float4 *>(smem);
    float4 *FB = FA + SA;
    float *FC = reinterpret_cast<float *>(FB + SA);
    float4 *xs = reinterpret_cast<float4 *>(FC + SA);
    float *DV = reinterpret_cast<float *>(smem);
    const int lds_main = 36 * SA + 16 * VP > 48 * SA ? 36 * SA + 16 * VP : 48 * SA;
    double *red = reinterpret_cast<double *>(smem + lds_main);

    const bool active = tid < nq;
    const uint4 *pl = reinterpret_cast<const uint4 *>(a.blob + td.blob_off);

    // ---- stream the tile: 13 coalesced 16 B/lane loads per thread (4 consecutive slots each) ----
    uint4 q_lv01 = make_uint4(0, 0, 0, 0), q_lv23 = q_lv01, q_nb01 = q_lv01, q_nb23 = q_lv01;
    float4 dm[9];
    if (active) {
        q_lv01 = pl[0 * nq + tid];
        q_lv23 = pl[1 * nq + tid];
        q_nb01 = pl[2 * nq + tid];
        q_nb23 = pl[3 * nq + tid];
        const float4 *dpl = reinterpret_cast<const float4 *>(pl + 4 * nq);
#pragma unroll
        for (int c = 0; c < 9; ++c) dm[c] = dpl[c * nq + tid];
    }
    // ---- stage the tile's vertex positions ----
    for (int v = tid; v < td.n_verts; v += nthr) {
        const size_t gv = size_t(a.gvid[td.vert_off + v]) * 3;
        xs[v] = make_float4(a.x[gv], a.x[gv + 1], a.x[gv + 2], 0.f);
    }
    if (tid == 0) {
        FA[ZS] = make_float4(0.f, 0.f, 0.f, 0.f);
        FB[ZS] = make_float4(0.f, 0.f, 0.f, 0.f);
        FC[ZS] = 0.f;
    }
    __syncthreads();
    if (DBG(DBG_EXIT_AFTER_LOAD)) {
        float chk = dm[0].x + dm[4].y + dm[8].z + float(q_lv01.x ^ q_lv23.y ^ q_nb01.z ^ q_nb23.w) + xs[tid % td.n_verts].x;
        if (chk == 12345.678f) a.partials[0] = chk;
        return;
    }

    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----
    float scal[4] = {0.f, 0.f, 0.f, 0.f};  // c2 * d(penalty)/d(det F), 0 unless owned and inverted
    float e_b = 0.f, e_s = 0.f;
    if (active) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t w0 = comp(q_lv01, p), w1 = comp(q_lv23, p);
            const bool owned = (w0 & kOwnedBit) != 0;
            float F[9];
            slot_F(xs, w0 & 0x7fffu, w0 >> 16, w1 & 0xffffu, w1 >> 16, dm, p, F);
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
            if (owned) {
                e_b += pen;
                scal[p] = a.c2 * dpen;
            }
            const int s = p * nq + tid;  // lds_index(4 * tid + p)
            FA[s] = make_float4(F[0], F[1], F[2], F[3]);
            FB[s] = make_float4(F[4], F[5], F[6], F[7]);
            FC[s] = F[8];
            SLOT_FENCE();
        }
    }
    __syncthreads();
    if (DBG(DBG_EXIT_AFTER_P1)) {
        if (e_b == 12345.678f) a.partials[0] = e_b;
        return;
    }

    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----
    // With the balanced slot order, position p of lane t is item p * nq + t and items below n_owned
    // are the owned ones, so `owned` is uniform across all but one wave per position: halo slots
    // skip their twelve gathers with a real branch.
    float H[4][9];
    if (active && !DBG(DBG_SKIP_P2)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t n01 = comp(q_nb01, p), n23 = comp(q_nb23, p);
#pragma unroll
            for (int c = 0; c < 9; ++c) H[p][c] = 0.f;
            if (n01 & kOwnedBit) {
                uint32_t nb[4] = {n01 & 0x7fffu, n01 >> 16, n23 & 0xffffu, n23 >> 16};
                const int s = p * nq + tid;
                if (DBG(DBG_LOCAL_GATHER2)) nb[0] = nb[1] = nb[2] = nb[3] = uint32_t(s);
                const float4 fa = FA[s], fb = FB[s];
                const float fc = FC[s];
                const float deg = float(int(nb[0] != ZS) + int(nb[1] != ZS) + int(nb[2] != ZS) + int(nb[3] != ZS));
                float acc[9] = {deg * fa.x, deg * fa.y, deg * fa.z, deg * fa.w, deg * fb.x,
                                deg * fb.y, deg * fb.z, deg * fb.w, deg * fc};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 ga = FA[nb[k]], gb = FB[nb[k]];
                    const float gc = FC[nb[k]];
                    acc[0] -= ga.x; acc[1] -= ga.y; acc[2] -= ga.z; acc[3] -= ga.w;
                    acc[4] -= gb.x; acc[5] -= gb.y; acc[6] -= gb.z; acc[7] -= gb.w;
                    acc[8] -= gc;
                }
                float sq = 0.f;
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    sq += acc[c] * acc[c];
                    H[p][c] = acc[c];
                }
                e_s += 0.5f * sq;
            }
            SLOT_FENCE();  // keep one slot's gathers in flight, not four (VGPR budget)
        }
    }
    __syncthreads();  // every read of F is done; overwrite it with H in place

    if (WITH_GRAD) {
        if (active) {
            // Dm^-1 and the vertex ids are needed again by pass 3.  Re-issuing their 11 loads here
            // instead of pinning 44 VGPRs across pass 2 keeps the kernel inside 128 VGPRs
            // (16 waves/CU) without scratch spills.
            q_lv01 = pl[0 * nq + tid];
            q_lv23 = pl[1 * nq + tid];
            const float4 *dpl = reinterpret_cast<const float4 *>(pl + 4 * nq);
#pragma unroll
            for (int c = 0; c < 9; ++c) dm[c] = dpl[c * nq + tid];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int s = p * nq + tid;
                FA[s] = make_float4(H[p][0], H[p][1], H[p][2], H[p][3]);
                FB[s] = make_float4(H[p][4], H[p][5], H[p][6], H[p][7]);
                FC[s] = H[p][8];
            }
        }
        __syncthreads();

        // ---- pass 3: P = c1 L^T H + c2 dpen cof(F);  d = P Dm^-T (per-tet vertex forces) ----
        // d[k] = P Dminv[k,:]^T is the force on local vertex k+1; vertex 0 gets -(d1+d2+d3).
        // Held in registers across the barrier, then written over F/H/xs (all dead by then).
        float D[4][9];
        if (active && !DBG(DBG_SKIP_P3)) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t n01 = comp(q_nb01, p), n23 = comp(q_nb23, p);
                uint32_t nb[4] = {n01 & 0x7fffu, n01 >> 16, n23 & 0xffffu, n23 >> 16};
                const float deg = float(int(nb[0] != ZS) + int(nb[1] != ZS) + int(nb[2] != ZS) + int(nb[3] != ZS));
                if (DBG(DBG_LOCAL_GATHER3)) nb[0] = nb[1] = nb[2] = nb[3] = uint32_t(p * nq + tid);
                // own H back from LDS (it is 0 on halo slots) rather than 36 VGPRs held across the barrier
                const int so = p * nq + tid;
                const float4 ha = FA[so], hb = FB[so];
                float P[9] = {deg * ha.x, deg * ha.y, deg * ha.z, deg * ha.w, deg * hb.x,
                              deg * hb.y, deg * hb.z, deg * hb.w, deg * FC[so]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 ga = FA[nb[k]], gb = FB[nb[k]];
                    const float gc = FC[nb[k]];
                    P[0] -= ga.x; P[1] -= ga.y; P[2] -= ga.z; P[3] -= ga.w;
                    P[4] -= gb.x; P[5] -= gb.y; P[6] -= gb.z; P[7] -= gb.w;
                    P[8] -= gc;
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) P[c] *= a.c1;
                if (scal[p] != 0.f) {  // inverted owned tet: rebuild F (it was overwritten by H)
                    const uint32_t w0 = comp(q_lv01, p), w1 = comp(q_lv23, p);
                    float F[9], C[9];
                    slot_F(xs, w0 & 0x7fffu, w0 >> 16, w1 & 0xffffu, w1 >> 16, dm, p, F);
                    cof3(F, C);
#pragma unroll
                    for (int c = 0; c < 9; ++c) P[c] += scal[p] * C[c];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        D[p][3 * k + i] = P[3 * i + 0] * comp(dm[3 * k + 0], p) + P[3 * i + 1] * comp(dm[3 * k + 1], p) +
                                          P[3 * i + 2] * comp(dm[3 * k + 2], p);
                SLOT_FENCE();
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < 9; ++c) D[p][c] = 0.f;
        }
        // prefetch this thread's vertex incidence chunks (thread t gathers local vertex t) so their
        // HBM latency hides behind the barriers and the d write
        constexpr int kPre = 6;
        const uint2 *inc = reinterpret_cast<const uint2 *>(pl + kPlanes * nq);
        const uint16_t *inc_off = reinterpret_cast<const uint16_t *>(inc + td.n_inc4);
        int pc0 = 0, pc1 = 0;
        uint2 pre[kPre];
        if (tid < td.n_verts) {
            pc0 = inc_off[tid];
            pc1 = inc_off[tid + 1];
#pragma unroll
            for (int q = 0; q < kPre; ++q) pre[q] = pc0 + q < pc1 ? inc[pc0 + q] : make_uint2(0u, 0u);
        }
        __syncthreads();
        // ---- write the vertex forces: plane a holds (fx, fy, fz) of local vertex a for every slot ----
        const int plane = 3 * SA;  // floats per plane
        if (active) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float *d0 = DV + 3 * (p * nq + tid);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    d0[i] = -(D[p][i] + D[p][3 + i] + D[p][6 + i]);
                    d0[plane + i] = D[p][i];
                    d0[2 * plane + i] = D[p][3 + i];
                    d0[3 * plane + i] = D[p][6 + i];
                }
            }
        }
        if (tid < 12) DV[(tid / 3) * plane + 3 * ZS + tid % 3] = 0.f;  // the zero slot of every plane
        __syncthreads();

        // ---- per-vertex gather of the incident tets' forces: fixed order, no atomics ----
        // Entry = (lds slot << 2) | local vertex; its force sits at DV[a * plane + 3 * slot].
        // Exclusive vertices go straight to grad, vertices shared with other tiles to the staging rows.
        // (The first kPre chunks of this thread's first vertex were prefetched before the barriers.)
        const float gscale = a.grad_out ? *a.grad_out : 1.f;
        for (int v = tid; v < td.n_verts; v += nthr) {
            const int c0 = v == tid ? pc0 : int(inc_off[v]), c1 = v == tid ? pc1 : int(inc_off[v + 1]);
            float gx = 0.f, gy = 0.f, gz = 0.f;
            auto gather4 = [&](const uint2 w) {
                const uint32_t ent[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float *f = DV + (ent[q] & 3u) * plane + 3 * (ent[q] >> 2);
                    gx += f[0];
                    gy += f[1];
                    gz += f[2];
                }
            };
            if (!DBG(DBG_SKIP_VGATHER)) {
                int c = c0;
                if (v == tid) {
#pragma unroll
                    for (int q = 0; q < kPre; ++q)
                        if (c0 + q < c1) gather4(pre[q]);
                    c = c0 + kPre;
                }
                for (; c < c1; ++c) gather4(inc[c]);
            }
            float *dst = v < td.n_excl ? a.grad + size_t(a.gvid[td.vert_off + v]) * 3
                                       : a.stage + (size_t(td.stage_off) + size_t(v - td.n_excl)) * 3;
            const float sc = v < td.n_excl ? gscale : 1.f;
            if (DBG(DBG_SKIP_OUT)) {
                if (gx == 1234.5f) dst[0] = gy + gz;
                continue;
            }
            dst[0] = gx * sc;
            dst[1] = gy * sc;
            dst[2] = gz * sc;
        }
    }

    // ---- deterministic block reduction of the two energy terms (fixed order, double) ----
    double ds = wave_sum(double(e_s)), db = wave_sum(double(e_b));
    const int wave = tid / kWave, lane = tid % kWave, nw = (nthr + kWave - 1) / kWave;
    if (lane == 0) {
        red[2 * wave] = ds;
        red[2 * wave + 1] = db;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0, b = 0.0;
        for (int w = 0; w < nw; ++w) {
            s += red[2 * w];
            b += red[2 * w + 1];
        }
        a.partials[2 * size_t(tile)] = s;
        a.partials[2 * size_t(tile) + 1] = b;
    }
}

struct FinishArgs {
    const int32_t *fin_vid, *fin_off, *fin_idx;
    int64_t n_finish;
    const float *stage;
    float *grad;
    const float *grad_out;
    const double *partials;
    int64_t n_tiles;
    float c1, c2;
    float *energy;
    double *terms;
};

// Vertices touched by several tiles: sum their staged partial gradients in plan order
// (deterministic).  The last workgroup folds the per-tile energy partials, again in fixed order.
__global__ __launch_bounds__(256) void finish_kernel(const FinishArgs a)
{
    __shared__ double red[2 * 256];
    const int tid = threadIdx.x;
    if (blockIdx.x == gridDim.x - 1) {
        if (!a.energy) return;
        double s = 0.0, b = 0.0;
        for (int64_t t = tid; t < a.n_tiles; t += 256) {
            s += a.partials[2 * t];
            b += a.partials[2 * t + 1];
        }
        red[tid] = s;
        red[256 + tid] = b;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                red[tid] += red[tid + off];
                red[256 + tid] += red[256 + tid + off];
            }
            __syncthreads();
        }
        if (tid == 0) {
            a.terms[0] = red[0];
            a.terms[1] = red[256];
            a.energy[0] = float(double(a.c1) * red[0] + double(a.c2) * red[256]);
        }
        return;
    }
    if (!a.grad) return;
    const float gscale = a.grad_out ? *a.grad_out : 1.f;
    const int64_t stride = int64_t(gridDim.x - 1) * 256;
    for (int64_t k = int64_t(blockIdx.x) * 256 + tid; k < a.n_finish; k += stride) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int32_t e = a.fin_off[k]; e < a.fin_off[k + 1]; ++e) {
            const float *r = a.stage + size_t(a.fin_idx[e]) * 3;
            gx += r[0];
            gy += r[1];
            gz += r[2];
        }
        float *g = a.grad + size_t(a.fin_vid[k]) * 3;
        g[0] = gx * gscale;
        g[1] = gy * gscale;
        g[2] = gz * gscale;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(const float *in, const float *scalar, float *out, int64_t n)
{
    const float s = *scalar;
    const int64_t stride = int64_t(gridDim.x) * 256;
    const int64_t n4 = n >> 2;
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (vec) {
        for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += stride) {
            float4 v = in4[i];
            out4[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
        }
        for (int64_t i = 4 * n4 + int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i] * s;
    } else {
        for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i] * s;
    }
}

// max |g| over the array, as the bit pattern of a non-negative float (monotone under uint compare)
__global__ __launch_bounds__(256) void absmax_kernel(const float *g, int64_t n, unsigned int *out)
{
    __shared__ float red[256 / kWave];
    float m = 0.f;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(g[i]));
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, kWave));
    if (threadIdx.x % kWave == 0) red[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; ++w) m = fmaxf(m, red[w]);
        atomicMax(out, __float_as_uint(m));
    }
}

__global__ __launch_bounds__(256) void clamp_kernel(float *g, int64_t n, const unsigned int *mx, float thr, float s)
{
    const float m = __uint_as_float(*mx);
    if (!(m > thr)) return;
    const float f = s / m;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) g[i] *= f;
}

int grid_for(int64_t n, int per_block, int cap)
{
    int64_t b = (n + per_block - 1) / per_block;
    return int(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

hipError_t launch_eval_kernels(const EvalArgs &e, hipStream_t stream, hipEvent_t *ev);

hipError_t configure_kernels(int lds_bytes)
{
    const void *fns[4] = {reinterpret_cast<const void *>(&tile_energy_kernel<true, 1024>),
                          reinterpret_cast<const void *>(&tile_energy_kernel<false, 1024>),
                          reinterpret_cast<const void *>(&tile_energy_kernel<true, 768>),
                          reinterpret_cast<const void *>(&tile_energy_kernel<false, 768>)};
    for (const void *fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_eval(const EvalArgs &e, hipStream_t stream, hipEvent_t *ev)
{
    if (ev) {
        hipError_t err = hipEventRecord(ev[0], stream);
        if (err != hipSuccess) return err;
    }
    hipError_t rc = launch_eval_kernels(e, stream, ev);
    if (rc == hipSuccess && ev) rc = hipEventRecord(ev[2], stream);
    return rc;
}

hipError_t launch_eval_kernels(const EvalArgs &e, hipStream_t stream, hipEvent_t *ev)
{
    if (e.n_tiles > 0) {
        KernelArgs k;
        k.tiles = e.tiles;
        k.blob = e.blob;
        k.gvid = e.gvid;
        k.x = e.x;
        k.grad_out = e.grad_out;
        k.grad = e.grad;
        k.stage = e.stage;
        k.partials = e.partials;
        k.c1 = e.c1;
        k.c2 = e.c2;
        k.order = e.order;
        k.n_tiles = int(e.n_tiles);
        k.tiles_per_xcd = int((e.n_tiles + 7) / 8);
        k.dbg = e.dbg;
        const dim3 grid(unsigned(8 * k.tiles_per_xcd)), block(unsigned(e.block_threads));
        const bool small = false;  // the 768-thread instantiation (168 VGPRs) schedules worse and spills more; unused
        if (e.grad && small)
            hipLaunchKernelGGL((tile_energy_kernel<true, 768>), grid, block, size_t(e.lds_bytes), stream, k);
        else if (e.grad)
            hipLaunchKernelGGL((tile_energy_kernel<true, 1024>), grid, block, size_t(e.lds_bytes), stream, k);
        else if (small)
            hipLaunchKernelGGL((tile_energy_kernel<false, 768>), grid, block, size_t(e.lds_bytes), stream, k);
        else
            hipLaunchKernelGGL((tile_energy_kernel<false, 1024>), grid, block, size_t(e.lds_bytes), stream, k);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
    }
    if (ev) {
        hipError_t err = hipEventRecord(ev[1], stream);
        if (err != hipSuccess) return err;
    }
    FinishArgs f;
    f.fin_vid = e.fin_vid;
    f.fin_off = e.fin_off;
    f.fin_idx = e.fin_idx;
    f.n_finish = e.grad ? e.n_finish : 0;
    f.stage = e.stage;
    f.grad = e.grad;
    f.grad_out = e.grad_out;
    f.partials = e.partials;
    f.n_tiles = e.n_tiles;
    f.c1 = e.c1;
    f.c2 = e.c2;
    f.energy = e.energy;
    f.terms = e.terms;
    if (f.n_finish == 0 && !f.energy) return hipSuccess;
    const int vb = f.n_finish > 0 ? grid_for(f.n_finish, 256, 2048) : 0;
    hipLaunchKernelGGL(finish_kernel, dim3(unsigned(vb + 1)), dim3(256), 0, stream, f);
    return hipGetLastError();
}

hipError_t launch_scale(const float *in, const float *scalar, float *out, int64_t n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scale_kernel, dim3(unsigned(grid_for(n, 1024, 2048))), dim3(256), 0, stream, in, scalar, out, n);
    return hipGetLastError();
}

hipError_t launch_grad_limit(float *grad, int64_t n, float thr, float s, void *workspace, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    unsigned int *mx = static_cast<unsigned int *>(workspace);
    hipError_t e = hipMemsetAsync(mx, 0, sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    const int g = grid_for(n, 1024, 2048);
    hipLaunchKernelGGL(absmax_kernel, dim3(unsigned(g)), dim3(256), 0, stream, grad, n, mx);
    hipLaunchKernelGGL(clamp_kernel, dim3(unsigned(g)), dim3(256), 0, stream, grad, n, mx, thr, s);
    return hipGetLastError();
}

}  // namespace tsamd
