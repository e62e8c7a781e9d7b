// gfx950 (MI355X / CDNA4) kernels for the tet-sphere geometry energy.
//
// One workgroup evaluates one *tile*: a cluster of tets (owned + one-ring face halo; ~1 500 slots at the default
// 768 threads x 2 slots per lane and <= 80 KiB of LDS, two workgroups per CU) whose deformation gradients F fit
// the LDS.  Per evaluation the tile's dword planes (10-bit local vertices + 6-bit ranks, 16-bit record tokens of the
// face neighbours, fp32 Dm^-1: 13 planes = 52 B per slot; 22 with an explicit element operator, 4 with
// rebuild_dminv -- plan.h) stream from HBM exactly once, coalesced; F, L F, L^T L F and the vertex accumulation
// never leave the CU.  No MFMA: 3x3 algebra at ~4 flop/B.
//
// What each stage stands for in the reference
// (/root/reference/tssplat_ext/tet_spheres/tet_spheres_cuda.cu):
//   pass 1  F = Ds Dm^-1, det, penalty     <- cusparseSpMV(G,x) :167,:221 + cuda_forward_det :48-66
//   pass 2  H = L F, 1/2|H|^2              <- cusparseSpMV(GTLTLG,x) :131 + cublasSdot :154 (factored, no M)
//   pass 3  P = c1 L^T H + c2 dpen cof(F)  <- cusparseSpMV(GTLTLG,x) :216 + cuda_backward_det :68-102
//           d = P Dm^-T, per-vertex sums   <- cusparseSpMV(TRANSPOSE, G) :248 (their transposed COO SpMV
//                                             scatters with atomics; LDS f32 atomics measured ~3 clk/lane
//                                             on gfx950, so each slot writes its four corner forces to plan-assigned
//                                             entries of a per-vertex array and a lane per vertex adds them up)
//   finish  sum shared-vertex partials, reduce energy, * grad_out
//                                          <- cublasSasum :185, host combine :191, cublasSscal :258
// There is no host synchronisation anywhere (the reference blocks three times
// per forward+backward: :154, :185, :257).
#include <hip/hip_runtime.h>

#include <vector>

#include "kernels.h"

namespace tsamd {
namespace {

constexpr int kWave = 64;

// Scheduling fence between the slots a lane processes in a pass: one slot's gathers in flight at a time (VGPR budget;
// letting the compiler interleave them measured +0.1 %).
#define SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
// Wave priority by stage: the further a workgroup is through its tile, the higher its waves' issue priority (pass 1: 0 ... the tail: 3).
// Two workgroups share a CU; the one that is nearer its end frees its LDS sooner when the other -- which is mostly waiting for its
// stream anyway -- yields the issue slots.  Measured: 512 x kuhn19 -0.6 %, a.veg x 952 -1.4 %; the REVERSE order costs +14 %, and giving
// the head of a tile (the load issue) top priority costs +8 % (profiles/r05_experiments.md) -- a workgroup in its compute stages must
// not be disturbed, which is also why workgroups that live for more than one tile lose: they never become the "older" one.
#define STAGE_PRIORITY(n) __builtin_amdgcn_s_setprio(n)
// A register array that only the active lanes of a tile ever read still needs a definition on the inactive lanes' path (an
// array left undefined on a path is given registers from the kernel entry on).  An empty asm statement "defines" it there at no
// cost; zero-filling instead cost ~40 v_mov_b32 per wave and tile (tile kernel -1.8 %, profiles/r05_experiments.md).
#define UNDEF(x) asm volatile("" : "=v"(x))
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

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

// Explicit global-address-space views of the kernel's pointers.  In the out-of-line tile body the compiler
// cannot infer the address space of pointers read from KernelArgs and would emit flat_* loads, which share
// the LDS counter (lgkmcnt): every LDS wait would then also wait for outstanding HBM loads.
#define GLOBAL_AS __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ GLOBAL_AS T *as_global(T *p)
{
    return (GLOBAL_AS T *)p;
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// LDS accesses by absolute byte address.  The kernels' only LDS object is the dynamic array, which starts at LDS
// address 0 as long as a kernel has no static LDS object: configure_kernels() checks exactly that for every tile
// kernel instantiation (hipFuncGetAttributes: sharedSizeBytes == 0) and refuses to run otherwise (a device-side check
// in the prologue costs 26 VGPRs, see profiles/r02_experiments.md); addressing it as `smem + offset` makes the compiler add the array's link-time
// address -- a literal 0 -- to every computed offset: one wasted VALU instruction per access in a kernel that is
// bound by instruction issue.
#define LDS_AS __attribute__((address_space(3)))
template <class T>
__device__ __forceinline__ LDS_AS T *lds_at(uint32_t byte_addr)
{
    return (LDS_AS T *)(uintptr_t)byte_addr;
}

// F of one slot from the staged positions (byte offsets o0..o3 behind LDS address xs) and the slot's Dm^-1.
// The staging area holds (x, y, z, 0); reading it as 4 x u32 behind an asm fence keeps the compiler from
// narrowing the access to ds_read_b96, which costs 8 LDS cycles against 4 for ds_read_b128.
template <class VF>
__device__ __forceinline__ void slot_F(uint32_t xs, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3,
                                       const VF *dm, int p, float *F)
{
    const v4u r0 = *lds_at<const v4u>(xs + o0), r1 = *lds_at<const v4u>(xs + o1),
              r2 = *lds_at<const v4u>(xs + o2), r3 = *lds_at<const v4u>(xs + o3);
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
            F[3 * i + j] = Ds[3 * i + 0] * dm[j][p] + Ds[3 * i + 1] * dm[3 + j][p] + Ds[3 * i + 2] * dm[6 + j][p];
}

// Dm^-1 of one slot from the staged REST positions (rebuild_dminv plans, plan.h: kPlanesRebuild): Dm has the rest
// edges x1-x0, x2-x0, x3-x0 as columns, its inverse is cofactor^T / det -- the formula plan.cpp evaluates in double
// for the streamed planes, here in fp32 (1 / det through v_rcp_f32 and one Newton step).  dm[3 i + k][p] = Dm^-1[i][k].
template <class VF>
__device__ __forceinline__ void rebuild_dminv(uint32_t rs, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3,
                                              VF *dm, int p)
{
    const v4u r0 = *lds_at<const v4u>(rs + o0), r1 = *lds_at<const v4u>(rs + o1),
              r2 = *lds_at<const v4u>(rs + o2), r3 = *lds_at<const v4u>(rs + o3);
    asm volatile("" : : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
    const float x0 = __uint_as_float(r0.x), y0 = __uint_as_float(r0.y), z0 = __uint_as_float(r0.z);
    // D[3 i + k] = coordinate i of edge k
    const float D[9] = {__uint_as_float(r1.x) - x0, __uint_as_float(r2.x) - x0, __uint_as_float(r3.x) - x0,
                        __uint_as_float(r1.y) - y0, __uint_as_float(r2.y) - y0, __uint_as_float(r3.y) - y0,
                        __uint_as_float(r1.z) - z0, __uint_as_float(r2.z) - z0, __uint_as_float(r3.z) - z0};
    float C[9];
    cof3(D, C);
    const float det = D[0] * C[0] + D[1] * C[1] + D[2] * C[2];
    float r = __builtin_amdgcn_rcpf(det);
    r = __builtin_fmaf(__builtin_fmaf(-det, r, 1.f), r, r);
    r = det != 0.f ? r : 0.f;   // padding slots (all four vertices = vertex 0): Dm = 0, nothing reads their records
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) dm[3 * i + k][p] = C[3 * k + i] * r;
}

// One 48-byte LDS record (plan.h: record_token) = [tail quad | entries 0..3 | entries 4..7] holds F, later H, later
// the 4 x 3 vertex forces.  A record is named by the byte address `t` of its ninth matrix entry, which sits in the
// tail quad at a dword that rotates with bits 3-4 of the record index (at a fixed position the 48-byte stride
// would fold the 4-byte gathers of it onto 8 of the 32 banks); `t & ~15` is the record base.  The neighbour
// planes store t / 4, so a gathered record costs one shift and one mask of address arithmetic -- the kernel is
// bound by instruction issue (profiles/r02_experiments.md), and the round-1 form (index * 48, then the rotation
// recomputed from the index) spent five VALU instructions per record on it.
struct Mat9 {
    v2f p01, p23, p45, p67;
    float p8;
};

__device__ __forceinline__ uint32_t own_token_addr(uint32_t idx) { return 48u * idx + ((idx >> 1) & 12u); }

__device__ __forceinline__ Mat9 load_slot(uint32_t t)
{
    const uint32_t s = t & ~15u;
    const v4f a = *lds_at<const v4f>(s + 16), b = *lds_at<const v4f>(s + 32);
    Mat9 m;
    m.p01 = a.xy; m.p23 = a.zw; m.p45 = b.xy; m.p67 = b.zw;
    m.p8 = *lds_at<const float>(t);
    return m;
}

// Stores cost 2 cycles per source dword on the VGPR -> LDS path, so the tail goes out as one dword, not a quad.
__device__ __forceinline__ void store_slot(uint32_t t, const float *m)
{
    const uint32_t s = t & ~15u;
    *lds_at<v4f>(s + 16) = v4f{m[0], m[1], m[2], m[3]};
    *lds_at<v4f>(s + 32) = v4f{m[4], m[5], m[6], m[7]};
    *lds_at<float>(t) = m[8];
}

// acc = 4 * own - sum of the four neighbour records.  A face without a neighbour points at the slot's own record
// (plan.h), so it contributes own - own = 0: no degree, no dummy record, no branch.
__device__ __forceinline__ void sub9(Mat9 &acc, const Mat9 &g)
{
    acc.p01 -= g.p01; acc.p23 -= g.p23; acc.p45 -= g.p45; acc.p67 -= g.p67;
    acc.p8 -= g.p8;
}

__device__ __forceinline__ Mat9 mat9_of(const float *m)
{
    Mat9 r;
    r.p01 = v2f{m[0], m[1]}; r.p23 = v2f{m[2], m[3]}; r.p45 = v2f{m[4], m[5]}; r.p67 = v2f{m[6], m[7]};
    r.p8 = m[8];
    return r;
}

__device__ __forceinline__ Mat9 laplace_gather(Mat9 acc, const uint32_t *nb)
{
    Mat9 g0 = load_slot(nb[0]);
    acc.p01 *= 4.f; acc.p23 *= 4.f; acc.p45 *= 4.f; acc.p67 *= 4.f;
    acc.p8 *= 4.f;
    Mat9 g1 = load_slot(nb[1]);
    sub9(acc, g0);
    g0 = load_slot(nb[2]);
    sub9(acc, g1);
    g1 = load_slot(nb[3]);
    sub9(acc, g0);
    sub9(acc, g1);
    return acc;
}

// explicit element operator: acc = dg * own + sum_k w[k] * neighbour_k  (the weights carry their sign; 0 for a
// face that points at the slot itself)
__device__ __forceinline__ Mat9 operator_gather(const Mat9 &own, float dg, const float *w, const uint32_t *nb)
{
    Mat9 acc;
    acc.p01 = own.p01 * dg; acc.p23 = own.p23 * dg; acc.p45 = own.p45 * dg; acc.p67 = own.p67 * dg;
    acc.p8 = own.p8 * dg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const Mat9 g = load_slot(nb[k]);
        acc.p01 += g.p01 * w[k]; acc.p23 += g.p23 * w[k]; acc.p45 += g.p45 * w[k]; acc.p67 += g.p67 * w[k];
        acc.p8 += g.p8 * w[k];
    }
    return acc;
}

// Sum of a double over the 16 lanes of a DPP row, in every lane of the row (same butterfly as wave_sum; a 64-bit value
// moves as two dwords).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const v2u w = __builtin_bit_cast(v2u, v);
    const v2u r = {uint32_t(__builtin_amdgcn_update_dpp(0, int(w.x), CTRL, 0xf, 0xf, false)),
                   uint32_t(__builtin_amdgcn_update_dpp(0, int(w.y), CTRL, 0xf, 0xf, false))};
    return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double row_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x124>(v);  // row_ror:4
    v += dpp_f64<0x128>(v);  // row_ror:8
    return v;
}

// Sum over the 64 lanes of a wave, in a fixed order, returned in every lane.  DPP adds inside each row of 16 lanes
// (no LDS traffic, a few cycles each), then the four row sums through v_readlane.  `__shfl_down` is ds_bpermute on
// gfx950: twelve dependent LDS round trips for the two energy terms, ~2 000 cycles per tile (they used to sit at the very
// end of the kernel, where nothing of the workgroup is left to hide them; the sums now go out behind the H stores).
__device__ __forceinline__ float wave_sum(float v)
{
#define TSAMD_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
    v += TSAMD_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
    v += TSAMD_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
    v += TSAMD_DPP(v, 0x124);   // row_ror:4
    v += TSAMD_DPP(v, 0x128);   // row_ror:8  -> every lane holds the sum of its row of 16
#undef TSAMD_DPP
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)),
                r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

struct KernelArgs {
    const TileDesc *tiles;
    const uint8_t *blob;
    const int32_t *gvid;
    const int32_t *vdst;  // per tile vertex: >= 0 row of grad, < 0 staging row ~vdst (plan.h)
    const float *x;
    const float *grad_out;
    float *grad;
    float *stage;
    double *partials;
    float c1, c2;
    float ratio;        // c2 / c1, divided on the host (set_coefficients): a division costs every wave of every tile a dozen instructions
    const float *coef;  // optional: (c1, c2) read on the device instead (tsamd_evaluate_dev_coef); the kernel then divides itself
    int order;
    int n_tiles;
    int tiles_per_xcd;
    int vert_stride;   // gvid / vdst entries per tile: tile t's vertex ids start at t * vert_stride (= its descriptor's vert_off)
    int n_planes;      // dword planes per slot (explicit-operator plans: 22, or 18 for a symmetric operator -- plan.h)
};

constexpr uint32_t kXS = kRowTabBytes;   // LDS byte address of the staged positions (plan.h: LDS map of a tile)
constexpr uint32_t kZeroEntry = 4u * 66u;   // three zero dwords behind the 65 row starts: what a lane without an entry in a row reads

// byte offset (16 v) of a corner's staged position from its 16-bit vertex field (low / high half of a plane dword)
__device__ __forceinline__ uint32_t pos_lo(uint32_t w) { return (w & kVertMask) << 4; }
__device__ __forceinline__ uint32_t pos_hi(uint32_t w) { return ((w >> 16) & kVertMask) << 4; }

// One workgroup = one tile.  LDS map (plan.h): row table at 0, staged positions at kXS, reduction scratch, then at
// td.rec_base one 48 B record per slot: F (9 floats + pad), overwritten by H after pass 2; after pass 3 the same bytes hold
// the tile's force array, 12 bytes per (vertex, slot) incidence, row r = the r-th slot of every vertex that has one.
// The passes are separated by six workgroup barriers: staged positions | F | (F reads done) H | H | (H reads done) forces | sums.
// Lane t's p-th slot is item p * nq + t and lives at record index p * nq + t, so a wave's own-slot accesses walk consecutive
// 48 B records (conflict-free for 16 B accesses: 12 l mod 64 is a permutation of the 4-dword columns); the items below
// n_owned are the owned ones.
// SPT slots per lane (the plan is laid out for it); the default, two, with launch bounds <768 threads, 6 waves per SIMD> gives
// the 80-VGPR budget at which two workgroups share a CU.
template <bool WITH_GRAD, bool WEIGHTED, bool REBUILD, int SPT>
__device__ __forceinline__ void tile_body(const KernelArgs &a, const int tile, int32_t gv0)
{
    // named here, not passed in: a pointer parameter would be a generic pointer and every LDS access of the
    // out-of-line copy would turn into a flat_* instruction
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef uint32_t VU __attribute__((ext_vector_type(SPT)));
    typedef float VF __attribute__((ext_vector_type(SPT)));
    static_assert(!(REBUILD && WEIGHTED), "rebuild_dminv is built for the built-in operator");
    // A slot's own F stays in registers from pass 1 to pass 2, and its own H from pass 2 to pass 3, instead of being read
    // back from LDS.  The explicit-operator build reads both back (its weights take the registers).
    constexpr int kKeepF = SPT, kKeepH = WEIGHTED ? 0 : SPT;
    const TileDesc td = a.tiles[tile];
    const int tid = threadIdx.x, nthr = blockDim.x;
    // Device-side scalars -- the coefficients of a graph replay, autograd's grad_output -- are read through the constant address
    // space: scalar loads that come back with the tile descriptor, not vector loads the head (or, for grad_output, the very end)
    // of the tile would wait for.
    typedef const __attribute__((address_space(4))) float *ConstF;
    const float k_c1 = a.coef ? ((ConstF)(uintptr_t)a.coef)[0] : a.c1, k_c2 = a.coef ? ((ConstF)(uintptr_t)a.coef)[1] : a.c2;
    const float grad_out_scale = WITH_GRAD && a.grad_out ? *(ConstF)(uintptr_t)a.grad_out : 1.f;
    const auto g_blob = as_global(a.blob);
    const auto g_gvid = as_global(a.gvid);
    const auto g_vdst = as_global(a.vdst);
    const auto g_x = as_global(a.x);
    const auto g_grad = as_global(a.grad);
    const auto g_stage = as_global(a.stage);
    const auto g_partials = as_global(a.partials);
    const uint32_t VP = uint32_t(td.n_verts + 3) & ~3u;
    const uint32_t RS = kXS + 16u * VP;                         // staged rest positions (rebuild_dminv plans)
    const uint32_t RB = uint32_t(td.rec_base);                  // record 0
    LDS_AS double *red = lds_at<double>(RB - 256u);
    const int nq = td.s_pad / SPT;
    const bool active = tid < nq;
    const GLOBAL_AS uint32_t *pl = reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + td.blob_off);  // the planes, s_pad dwords each
    // lanes beyond the tile read lane 0's entries (and never use them): with every load unconditional the stream
    // is straight-line code and the compiler can count its waits (vmcnt(13)) instead of draining to vmcnt(0)
    const int lt = active ? tid : 0;
    // (a lane's SPT plane entries are loaded as one vector; the typedefs carry the alignment the address really has --
    // SPT dwords, and only 4 bytes for SPT == 3, where clang would otherwise assume the 16 of a padded 3-vector)
    typedef VU PlaneU __attribute__((aligned(SPT == 3 ? 4 : 4 * SPT)));
    typedef VF PlaneF __attribute__((aligned(SPT == 3 ? 4 : 4 * SPT)));
    // Non-temporal loads: a tile's planes are read once, by one CU -- kept out of the way of what IS re-read (positions shared with
    // the neighbouring tiles, the staging rows the finish kernel reads back).  Round 6, three scenes, same box each: tile kernel
    // -1.2 % (kuhn19), -1.3 % (a.veg), -1.7 % (Delaunay); finish kernel -5 % (round 1's kernel had measured +1.4 %).
    auto plane_u = [&](int q) -> VU { return __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS PlaneU *>(pl + q * td.s_pad + SPT * lt)); };
    auto plane_f = [&](int q) -> VF { return __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS PlaneF *>(pl + q * td.s_pad + SPT * lt)); };
    // Two planes that are always loaded together are ONE load of twice the width (plan.h: planes_paired): in the device image
    // of the blob planes q and q + 1 are interleaved per lane -- [plane q: SPT dwords | plane q + 1: SPT dwords] -- so the
    // default layout's thirteen 8-byte loads per lane become six 16-byte loads and one 8-byte load.
    constexpr bool kPaired = planes_paired(SPT);
    typedef uint32_t VU2 __attribute__((ext_vector_type(2 * SPT)));
    typedef float VF2 __attribute__((ext_vector_type(2 * SPT)));
    auto pair_u = [&](int q, VU &lo, VU &hi) {
        if (kPaired) {
            const VU2 t = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS VU2 *>(pl + q * td.s_pad + 2 * SPT * lt));
#pragma unroll
            for (int p = 0; p < SPT; ++p) lo[p] = t[p], hi[p] = t[SPT + p];
        } else {
            lo = plane_u(q), hi = plane_u(q + 1);
        }
    };
    auto pair_f = [&](int q, VF &lo, VF &hi) {
        if (kPaired) {
            const VF2 t = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS VF2 *>(pl + q * td.s_pad + 2 * SPT * lt));
#pragma unroll
            for (int p = 0; p < SPT; ++p) lo[p] = t[p], hi[p] = t[SPT + p];
        } else {
            lo = plane_f(q), hi = plane_f(q + 1);
        }
    };
    // ---- stream the tile ----
    // A CU pulls cold data at ~11 bytes per cycle whatever the rest of the chip does (tools/ubench_ingest.hip), in the order the
    // loads were issued, wave after wave.  The order below is built around that FIFO:
    //   1. (kernel entry) the vertex ids;  2. the two vertex planes -- 12 KB per workgroup, about what the id's latency covers;
    //   3. the positions, as soon as the ids are back: they reach the LDS behind 12 KB instead of behind the whole stream, so the
    //      barrier that publishes them falls EARLY in the stream, not at its end;
    //   4. the nine Dm^-1 planes: every wave computes its F as soon as ITS planes have landed;
    //   5. the neighbour planes (and an explicit operator's weights) and the row table, which nobody needs before pass 2, go
    //      out after the barrier.
    VU q_lv01, q_lv23, q_nb01, q_nb23;
    pair_u(0, q_lv01, q_lv23);
    VF dm[9];
    const int kBasePlanes = REBUILD ? kPlanesRebuild : (WEIGHTED ? a.n_planes : kPlanes);   // (a compile-time constant unless WEIGHTED)
    const GLOBAL_AS uint16_t *g_rowtab = reinterpret_cast<const GLOBAL_AS uint16_t *>(pl + size_t(kBasePlanes) * td.s_pad);
    // rest positions of the tile's vertices (rebuild_dminv plans): one coalesced float4 per lane, staged behind the positions
    const GLOBAL_AS v4f *g_rest = reinterpret_cast<const GLOBAL_AS v4f *>(g_rowtab + kRowTabEntries);
    v4f rest0 = v4f{0.f, 0.f, 0.f, 0.f};
    if (REBUILD) rest0 = g_rest[tid < td.n_verts ? tid : 0];
    // the positions: unconditional loads (lanes beyond the tile's vertices hold vertex 0 and do not store), so that the
    // stream stays straight-line code whose waits the compiler can count
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(gv0));  // (the wait for the id lands here: vmcnt(2), the two vertex planes stay in flight)
    float px, py, pz;
    {
        const size_t gv = size_t(gv0) * 3;
        px = g_x[gv], py = g_x[gv + 1], pz = g_x[gv + 2];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!REBUILD) {
#pragma unroll
        for (int c = 0; c < 8; c += 2) pair_f(4 + c, dm[c], dm[c + 1]);
        dm[8] = plane_f(12);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tid < td.n_verts) {
        *lds_at<v4f>(kXS + 16u * uint32_t(tid)) = v4f{px, py, pz, 0.f};   // (waits for the positions only: vmcnt(9))
        if (REBUILD) *lds_at<v4f>(RS + 16u * uint32_t(tid)) = rest0;
    }
    for (int v = tid + nthr; v < td.n_verts; v += nthr) {  // tiles with more vertices than lanes (rare)
        const size_t gv = size_t(g_gvid[td.vert_off + v]) * 3;
        *lds_at<v4f>(kXS + 16u * uint32_t(v)) = v4f{g_x[gv], g_x[gv + 1], g_x[gv + 2], 0.f};
        if (REBUILD) *lds_at<v4f>(RS + 16u * uint32_t(v)) = g_rest[v];
    }
    __syncthreads();
    // behind the barrier: what pass 2 and the end of the tile need
    pair_u(2, q_nb01, q_nb23);
    // explicit element operator (plans built with one): diagonal + the four row weights for pass 2; the column
    // weights for pass 3 replace them after pass 2
    VF wd, wk[4];
    if (WEIGHTED) {
        wd = plane_f(13);
        pair_f(14, wk[0], wk[1]);
        pair_f(16, wk[2], wk[3]);
    }
    // row table: start of row `tid` of the force array, in 12-byte entries.  An unconditional load, like the planes: behind the
    // join of a divergent branch the compiler waits for EVERY load in flight, and pass 1 would start behind the neighbour planes
    // (it did, for the first build of this layout: aveg x 952 0.4430 -> 0.4259 ms once the branch was gone).
    uint32_t row0 = 0;
    if (WITH_GRAD) row0 = __builtin_nontemporal_load(&g_rowtab[tid < 65 ? tid : 64]);
    __builtin_amdgcn_sched_barrier(0);

    // The smoothness coefficient is applied ONCE per vertex at the very end instead of nine times per slot:
    // dE/dF = c1 (Q + (c2 / c1) pen' cof F), so pass 3 works with s_pen = c2 / c1 and the per-vertex sums are scaled
    // by c1 (and by grad_output) when they are written.
    // Only while c2 / c1 is a well-behaved fp32 number: a tiny c1 would make the ratio overflow (inf * 0 = NaN on every
    // owned, non-inverted tet) or swamp Q's bits.  Outside that range -- and for c1 == 0, which drops Q -- pass 3 applies
    // c1 to Q and c2 to the penalty separately (q_scale) and the vertex sums are written unscaled.
    const float ratio = a.coef ? k_c2 / k_c1 : a.ratio;
    const bool factored = k_c1 != 0.f && __builtin_fabsf(ratio) <= 0x1p+40f;   // (false for inf and NaN as well)
    const float s_pen = factored ? ratio : k_c2, out_scale = factored ? k_c1 : 1.f, q_scale = factored ? 1.f : k_c1;
    // byte address of each own record's ninth entry (see load_slot), once per slot instead of once per access
    uint32_t t_own_reg[SPT];
#pragma unroll
    for (int p = 0; p < SPT; ++p) t_own_reg[p] = RB + own_token_addr(uint32_t(p * nq + tid));
    // The explicit-operator build has its ten weight registers on top of everything else: there the address is recomputed at
    // each of its five uses (from an opaque copy of the lane id, or the compiler keeps the result in a register all the same)
    // -- with it the vertex fields stay in registers for pass 3 like in the built-in kernel, without a spill.
    auto t_own_at = [&](int p) -> uint32_t {
        if (!WEIGHTED) return t_own_reg[p];
        uint32_t t = uint32_t(tid);
        asm volatile("" : "+v"(t));
        return RB + own_token_addr(uint32_t(p * nq) + t);
    };
#define t_own(p) t_own_at(p)
    // item number of the lane's p-th slot against the descriptor: owned below n_owned, a real slot below n_slots
    auto is_owned = [&](int p) { return p * nq + tid < td.n_owned; };

    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----
    STAGE_PRIORITY(0);
    float scal[SPT];  // (c2 / c1) * d(penalty)/d(det F), 0 unless owned and inverted
#pragma unroll
    for (int p = 0; p < SPT; ++p) scal[p] = 0.f;
    float e_b = 0.f, e_s = 0.f;
    float Fk[kKeepF > 0 ? kKeepF : 1][9];
#pragma unroll
    for (int p = 0; p < kKeepF; ++p)
#pragma unroll
        for (int c = 0; c < 9; ++c) UNDEF(Fk[p][c]);
    if (active) {
#pragma unroll
        for (int p = 0; p < SPT; ++p) {
            const uint32_t w0 = q_lv01[p], w1 = q_lv23[p];
            if (REBUILD) rebuild_dminv(RS, pos_lo(w0), pos_hi(w0), pos_lo(w1), pos_hi(w1), dm, p);
            float F[9];
            slot_F(kXS, pos_lo(w0), pos_hi(w0), pos_lo(w1), pos_hi(w1), dm, p, F);
            if (p < kKeepF) {
#pragma unroll
                for (int c = 0; c < 9; ++c) Fk[p][c] = F[c];
            }
            if (is_owned(p)) {   // halo slots only contribute F: a real branch (owned / halo is all but wave-uniform in
                                 // the balanced slot order), tile kernel 0.4643 -> 0.4611 ms (profiles/r03_experiments.md)
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
                e_b += pen;
                scal[p] = s_pen * dpen;
            }
            store_slot(t_own(p), F);
            SLOT_FENCE();
        }
    }
    // the row table as LDS byte addresses of the rows (wave 0 and lane 64; read from the scatter phase on)
    if (WITH_GRAD) {
        if (tid < 72) *lds_at<uint32_t>(4u * uint32_t(tid)) = tid < 65 ? RB + 12u * row0 : 0u;   // (+ the zero entry)
        if (nthr < 72 && tid < 8)   // a 64-thread workgroup: its first lanes write the table's tail as well
            *lds_at<uint32_t>(256u + 4u * uint32_t(tid)) = tid == 0 ? RB + 12u * uint32_t(g_rowtab[64]) : 0u;
    }
    __syncthreads();

    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----
    STAGE_PRIORITY(1);
    // `owned` is uniform across all but one wave per position: halo slots skip their gathers with a real branch.
    float H[SPT][9];
    auto neighbours = [](uint32_t n01, uint32_t n23, uint32_t *nb) {   // byte addresses of the four records' ninth entries
        nb[0] = (n01 & 0xffffu) << 2;
        nb[1] = (n01 >> 14) & 0x3fffcu;
        nb[2] = (n23 & 0xffffu) << 2;
        nb[3] = (n23 >> 14) & 0x3fffcu;
    };
    if (active) {
#pragma unroll
        for (int p = 0; p < SPT; ++p) {
            const uint32_t n01 = q_nb01[p], n23 = q_nb23[p];
            if (is_owned(p)) {
                uint32_t nb[4];
                neighbours(n01, n23, nb);
                Mat9 h;
                if (WEIGHTED) {
                    const float w4[4] = {wk[0][p], wk[1][p], wk[2][p], wk[3][p]};
                    h = operator_gather(load_slot(t_own(p)), wd[p], w4, nb);
                } else {
                    h = laplace_gather(p < kKeepF ? mat9_of(Fk[p]) : load_slot(t_own(p)), nb);
                }
                v2f sq = h.p01 * h.p01;
                sq = __builtin_elementwise_fma(h.p23, h.p23, sq);
                sq = __builtin_elementwise_fma(h.p45, h.p45, sq);
                sq = __builtin_elementwise_fma(h.p67, h.p67, sq);
                e_s += 0.5f * (sq.x + sq.y + h.p8 * h.p8);
                H[p][0] = h.p01.x; H[p][1] = h.p01.y; H[p][2] = h.p23.x; H[p][3] = h.p23.y;
                H[p][4] = h.p45.x; H[p][5] = h.p45.y; H[p][6] = h.p67.x; H[p][7] = h.p67.y;
                H[p][8] = h.p8;
            } else {   // halo / padding slot: H = 0.  (Defined on this path only: owned lanes do not pay for it, and an
                       // array left undefined on a path is given registers from the kernel entry on.)
#pragma unroll
                for (int c = 0; c < 9; ++c) H[p][c] = 0.f;
            }
            SLOT_FENCE();  // keep one slot's gathers in flight, not all of them (VGPR budget)
        }
    } else {
#pragma unroll
        for (int p = 0; p < SPT; ++p)
#pragma unroll
            for (int c = 0; c < 9; ++c) UNDEF(H[p][c]);
    }
    // ---- the two energy terms are complete: deterministic block reduction (fixed order; doubles across waves) ----
    // Done HERE, around a barrier the tile needs anyway, not at the end of the kernel: there the reduction was a serial
    // tail (wave sums, a barrier, one lane summing the waves) between this workgroup and its successor on the CU,
    // and the kernel time follows that tail at better than 1 : 1.
    const int wave = tid / kWave, lane = tid % kWave, nw = (nthr + kWave - 1) / kWave;
    auto publish_wave_sums = [&]() {
        e_s = wave_sum(e_s);
        e_b = wave_sum(e_b);
        if (lane == 0) {
            red[2 * wave] = double(e_s);
            red[2 * wave + 1] = double(e_b);
        }
    };
    auto sum_waves = [&]() {   // lane w fetches wave w's pair, DPP butterfly over <= 16 lanes; the last wave has the fewest slots
        if (wave == nw - 1) {
            double s = lane < nw ? red[2 * lane] : 0.0, b = lane < nw ? red[2 * lane + 1] : 0.0;
            s = row_sum_f64(s);
            b = row_sum_f64(b);
            if (lane == 0) {
                g_partials[2 * size_t(tile)] = s;
                g_partials[2 * size_t(tile) + 1] = b;
            }
        }
    };
    // With a gradient the wave sums go out behind the H stores (their DPP chain overlaps the stores' drain) and are added
    // up after the next barrier, inside pass 3: tile kernel -0.5 % against doing both around the barrier after pass 2.
    constexpr bool kSumLate = WITH_GRAD;
    if (!kSumLate) publish_wave_sums();
    __syncthreads();  // every read of F is done; overwrite it with H in place
    if (!kSumLate) sum_waves();

    if (WITH_GRAD) {
        if (active) {
            // (Dm^-1, needed again by pass 3, stays in registers)
#pragma unroll
            for (int p = 0; p < SPT; ++p) store_slot(t_own(p), H[p]);   // (zeros on halo and padding slots)
            if (WEIGHTED && a.n_planes == kPlanesWeighted) {   // pass 3 applies L^T: the column weights L[n_k, e] (a symmetric
                                                               // operator has none: the row weights serve both passes)
                pair_f(18, wk[0], wk[1]);
                pair_f(20, wk[2], wk[3]);
            }
        }
        publish_wave_sums();
        __syncthreads();
        sum_waves();

        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----
        STAGE_PRIORITY(2);
        // d[k] = P Dminv[k,:]^T is the force on local vertex k+1; vertex 0 gets -(d1+d2+d3).
        // Held in registers across the barrier, then scattered over H (all reads of it done by then).
        float D[SPT][9];
        if (active) {
#pragma unroll
            for (int p = 0; p < SPT; ++p) {
                uint32_t n01 = q_nb01[p], n23 = q_nb23[p];
                // (opaque copies: otherwise the eight record addresses decoded in pass 2 are kept in registers across
                // two barriers for reuse here -- 16 VGPRs that push the kernel over its 80-register budget)
                asm volatile("" : "+v"(n01), "+v"(n23));
                uint32_t nb[4];
                neighbours(n01, n23, nb);
                Mat9 q;
                if (WEIGHTED) {
                    const float w4[4] = {wk[0][p], wk[1][p], wk[2][p], wk[3][p]};
                    q = operator_gather(load_slot(t_own(p)), wd[p], w4, nb);
                } else {
                    q = laplace_gather(p < kKeepH ? mat9_of(H[p]) : load_slot(t_own(p)), nb);
                }
                float P[9] = {q.p01.x, q.p01.y, q.p23.x, q.p23.y, q.p45.x, q.p45.y, q.p67.x, q.p67.y, q.p8};
                if (!factored) {   // rare: c1 == 0 (only the penalty term is left) or c2 / c1 out of range
#pragma unroll
                    for (int c = 0; c < 9; ++c) P[c] *= q_scale;
                }
                if (scal[p] != 0.f) {  // inverted owned tet: rebuild F (it was overwritten by H).  The vertex fields stay in
                                       // registers from the stream phase on: the path is rare per TET, not per WAVE (with 1.5 % of the
                                       // headline scene's tets inverted, 63 % of all wave-slots take it), and a re-fetch from memory
                                       // here cost 7 % of the kernel (profiles/r03_experiments.md)
                    const uint32_t w0 = q_lv01[p], w1 = q_lv23[p];
                    float F[9], C[9];
                    slot_F(kXS, pos_lo(w0), pos_hi(w0), pos_lo(w1), pos_hi(w1), dm, p, F);
                    cof3(F, C);
#pragma unroll
                    for (int c = 0; c < 9; ++c) P[c] += scal[p] * C[c];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        D[p][3 * k + i] = P[3 * i + 0] * dm[3 * k + 0][p] + P[3 * i + 1] * dm[3 * k + 1][p] +
                                          P[3 * i + 2] * dm[3 * k + 2][p];
                SLOT_FENCE();
            }
        } else {
#pragma unroll
            for (int p = 0; p < SPT; ++p)
#pragma unroll
                for (int c = 0; c < 9; ++c) UNDEF(D[p][c]);
        }
        // Where each corner's force goes: row start (LDS byte address, from the table at LDS address 0, indexed by the corner's
        // rank) + 12 * vertex.  The table is read-only from here on, so its reads are issued ahead of the barrier.
        uint32_t fdst[SPT][4];
#pragma unroll
        for (int p = 0; p < SPT; ++p) {
            const uint32_t w0 = q_lv01[p], w1 = q_lv23[p];
            const uint32_t r0 = *lds_at<const uint32_t>((w0 >> 8) & 0xfcu), r1 = *lds_at<const uint32_t>((w0 >> 24) & 0xfcu),
                           r2 = *lds_at<const uint32_t>((w1 >> 8) & 0xfcu), r3 = *lds_at<const uint32_t>((w1 >> 24) & 0xfcu);
            fdst[p][0] = (w0 & kVertMask) * 12u + r0;
            fdst[p][1] = ((w0 >> 16) & kVertMask) * 12u + r1;
            fdst[p][2] = (w1 & kVertMask) * 12u + r2;
            fdst[p][3] = ((w1 >> 16) & kVertMask) * 12u + r3;
        }
        // the vertex blocks of this wave in the per-vertex sum: 64 consecutive tile vertices each; waves 4-7 take their four
        // blocks in reverse so that every SIMD (waves w, w + 4, w + 8) gets one long and one short block of the sorted vertices
        int vb0 = wave;
        if (wave & 4) {   // (a permutation of the group's waves whatever the block size: the last group may hold fewer than four)
            const int g4 = wave & ~3, top = g4 + 3 < nw - 1 ? g4 + 3 : nw - 1;
            vb0 = g4 + (top - wave);
        }
        int32_t dst_row = 0;  // where this lane's first vertex goes: fetched here, a whole phase ahead of its use
        if (64 * vb0 + lane < td.n_verts) dst_row = __builtin_nontemporal_load(&g_vdst[td.vert_off + 64 * vb0 + lane]);
        __syncthreads();   // all waves done with H and with the staged positions
        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----
        STAGE_PRIORITY(3);
        if (active) {
#pragma unroll
            for (int p = 0; p < SPT; ++p) {
                if (p * nq + tid < td.n_slots) {   // (padding slots -- at most three, the last lanes -- have no entries)
                    const float *d = D[p];
                    LDS_AS float *f0 = lds_at<float>(fdst[p][0]), *f1 = lds_at<float>(fdst[p][1]), *f2 = lds_at<float>(fdst[p][2]),
                                 *f3 = lds_at<float>(fdst[p][3]);
                    f0[0] = (-d[0] - d[3]) - d[6], f0[1] = (-d[1] - d[4]) - d[7], f0[2] = (-d[2] - d[5]) - d[8];   // (= -(f1 + f2 + f3), bit for bit)
                    f1[0] = d[0], f1[1] = d[1], f1[2] = d[2];
                    f2[0] = d[3], f2[1] = d[4], f2[2] = d[5];
                    f3[0] = d[6], f3[1] = d[7], f3[2] = d[8];
                }
            }
        }
        __syncthreads();

        // ---- per-vertex sums: lane = vertex, rows in order (= slot order: fixed, no atomics) ----
        // Lane r of every wave holds row r's start address and width: the row loop is wave-uniform (v_readlane), its trip
        // count the slot count of the wave's first -- fullest -- vertex.  Exclusive vertices go straight to grad, vertices
        // shared with other tiles to the staging rows, which the finish kernel sums in plan order.
        const float gscale = grad_out_scale * out_scale;
        const uint32_t tab = *lds_at<const uint32_t>(4u * uint32_t(lane));
        const uint32_t wid = *lds_at<const uint32_t>(4u * uint32_t(lane) + 4u) - tab;   // 12 * (vertices in row `lane`)
        auto vertex_sums = [&](int vb, int32_t row) {
            const int v = 64 * vb + lane;
            const uint32_t v12 = 12u * uint32_t(v);
            const int rows = __builtin_popcountll(__builtin_amdgcn_ballot_w64(wid > 768u * uint32_t(vb)));
            float gx = 0.f, gy = 0.f, gz = 0.f;
            // (four rows per trip, unrolled by hand -- v_readlane is convergent, the compiler does not unroll around it; rows
            // beyond the tile's last one are empty: width 0.  Branch-free: a lane beyond a row's width reads the three zero
            // dwords behind the row table instead, so that the four rows' reads are in flight together.  Eight rows per trip --
            // a.veg's fullest vertices carry 56 -- were measured in round 6: -4.6 % on a.veg before the read-once arrays went
            // non-temporal, nothing after (0.3814 against 0.3820 ms), +0.3 % on the lattice: tools/lab_variants.py rows8.)
            for (int r = 0; r < rows; r += 4) {
                const LDS_AS float *f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t base = uint32_t(__builtin_amdgcn_readlane(int(tab), r + u)), w = uint32_t(__builtin_amdgcn_readlane(int(wid), r + u));
                    f[u] = lds_at<const float>(v12 < w ? base + v12 : kZeroEntry);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    gx += f[u][0];
                    gy += f[u][1];
                    gz += f[u][2];
                }
            }
            if (v < td.n_verts) {
                const bool excl = row >= 0;
                GLOBAL_AS float *dst = (excl ? g_grad : g_stage) + size_t(excl ? row : ~row) * 3;
                const float sc = excl ? gscale : out_scale;
                // (non-temporal stores here were measured: tile kernel +28 %; the wave's end waits for their acknowledgement)
                dst[0] = gx * sc;
                dst[1] = gy * sc;
                dst[2] = gz * sc;
            }
        };
        if (64 * vb0 < td.n_verts) vertex_sums(vb0, dst_row);
        if (64 * nw < td.n_verts) {   // more vertices than lanes (rare; a wave-uniform branch: the loads of the extra rounds and the
                                      // waits the compiler puts around a loop with loads stay out of the common path)
            for (int vb = vb0 + nw; 64 * vb < td.n_verts; vb += nw)
                vertex_sums(vb, 64 * vb + lane < td.n_verts ? g_vdst[td.vert_off + 64 * vb + lane] : 0);
        }
    }
#undef t_own
}

template <bool WITH_GRAD, int BLOCK, int WPE, bool WEIGHTED = false, bool REBUILD = false, int SPT = kSlotsPerLane>
__global__ __launch_bounds__(BLOCK, WPE) void tile_energy_kernel(const KernelArgs a)
{
    // XCD-aware tile order: workgroup b lands on XCD b % 8 (observed, speed only), so give each
    // XCD a contiguous run of tiles -- the tiles of one sphere then share one L2.
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile_end = (xcd + 1) * a.tiles_per_xcd < a.n_tiles ? (xcd + 1) * a.tiles_per_xcd : a.n_tiles;
    const int tile = xcd * a.tiles_per_xcd + jb;
    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;
    // (lds_at() assumes the dynamic LDS array starts at LDS address 0: true for a kernel without static LDS objects,
    // which configure_kernels() verifies on the host.)
    // The position gather is the longest dependent chain at the head of a tile: vertex id -> position -> LDS.  The id's
    // address needs only the tile number and a kernel argument (vertex ids sit at tile * vert_stride, plan.cpp), so it is
    // requested HERE, before the tile descriptor is even asked for: the chain is two memory latencies instead of three.
    // Entries beyond n_verts name vertex 0 and are never used.
    // (non-temporal, like every read-once array of a tile -- planes, ids, destinations, row table: what they would evict is what IS
    // re-read, the positions shared with neighbouring tiles and the staging rows; round 6: a.veg x 952 tile kernel -6 %, finish -8 %)
    const int32_t gv0 = __builtin_nontemporal_load(&as_global(a.gvid)[size_t(tile) * size_t(a.vert_stride) + size_t(int(threadIdx.x) < a.vert_stride ? threadIdx.x : 0)]);
    __builtin_amdgcn_sched_barrier(0);
    tile_body<WITH_GRAD, WEIGHTED, REBUILD, SPT>(a, tile, gv0);
}

struct FinishArgs {
    const int32_t *fin_vid, *fin_off;
    int64_t n_finish;
    const float *stage;
    float *grad;
    const float *grad_out;
    const double *partials;
    int64_t n_tiles;
    float c1, c2;
    const float *coef;  // optional device (c1, c2), see KernelArgs
    float *energy;
    double *terms;
    float *energy_copy;   // optional second destination of the energy (a replay's per-launch argument: the ring slot of an energy exchange)
};

// Per-tile energy partials -> E = c1 E_s + c2 E_b (tet_spheres_cuda.cu:191), one workgroup of T threads, fixed
// order (bitwise repeatable).  Each thread owns tiles tid, tid + T, ... and issues its loads in batches of eight
// before summing: a plain dependent loop costs one HBM latency per tile (60 us for 19 k tiles).
template <int T>
__device__ __forceinline__ void energy_reduce(const FinishArgs &a, double *red)
{
    const int tid = threadIdx.x;
    const double2 *part = reinterpret_cast<const double2 *>(a.partials);
    double s = 0.0, b = 0.0;
    int64_t t = tid;
    for (; t + 7 * T < a.n_tiles; t += 8 * T) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[t + u * T];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s += v[u].x;
            b += v[u].y;
        }
    }
    for (; t < a.n_tiles; t += T) {
        const double2 v = part[t];
        s += v.x;
        b += v.y;
    }
    red[tid] = s;
    red[T + tid] = b;
    __syncthreads();
    for (int off = T / 2; off > 0; off >>= 1) {
        if (tid < off) {
            red[tid] += red[tid + off];
            red[T + tid] += red[T + tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.terms[0] = red[0];
        a.terms[1] = red[T];
        const float c1 = a.coef ? a.coef[0] : a.c1, c2 = a.coef ? a.coef[1] : a.c2;
        const float e = float(double(c1) * red[0] + double(c2) * red[T]);
        a.energy[0] = e;
        if (a.energy_copy) a.energy_copy[0] = e;
    }
}

// Forward-only evaluations: just the energy.
__global__ __launch_bounds__(1024) void energy_reduce_kernel(const FinishArgs a)
{
    __shared__ double red[2 * 1024];
    energy_reduce<1024>(a, red);
}

// Workgroup 0 reduces the energy partials (when asked to) while the others sum, for every vertex touched by
// several tiles, its staged partial gradients in plan order (deterministic): one launch, and the 8 us of the
// single-workgroup reduction hide behind the vertex work.
// (Ordinary loads: non-temporal ones for the staging rows and lists -- read for the last time / once per evaluation -- were measured,
// round 6: finish kernel 0.0269 -> 0.0409 ms on 512 x kuhn19, 0.0297 -> 0.0414 on a.veg x 952.  The rows are still in the memory-side
// cache from the tile kernel's stores; a non-temporal read gives that up.)
#define FIN_LOAD(p) (*(p))
__global__ __launch_bounds__(256) void finish_kernel(const FinishArgs a)
{
    __shared__ double red[2 * 256];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        if (a.energy) energy_reduce<256>(a, red);
        return;
    }
    const float gscale = a.grad_out ? *a.grad_out : 1.f;
    // one thread per shared vertex (measured faster than one thread per output float: 0.065 vs 0.093 ms)
    const int64_t stride = int64_t(gridDim.x - 1) * 256;
    for (int64_t k = int64_t(blockIdx.x - 1) * 256 + tid; k < a.n_finish; k += stride) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        const int32_t e0 = FIN_LOAD(&a.fin_off[k]), e1 = FIN_LOAD(&a.fin_off[k + 1]);   // consecutive rows, tile order
        // the first four rows go out together (most shared vertices have 2-4 copies): one memory latency instead of one
        // per row; rows beyond the vertex's own are re-reads of its last row and are not added.  Finish kernel 0.037 ->
        // 0.031 ms on the 512-sphere scene (2 / 3 / 8 rows: 0.034 / 0.031 / 0.031); same order of additions.
        constexpr int kRowsAhead = 4;
        {
            float3 r[kRowsAhead];
#pragma unroll
            for (int j = 0; j < kRowsAhead; ++j) {
                // (a vertex no tet references is in the list with zero rows: e1 - 1 may be -1 -- read row 0, add nothing)
                const int32_t e = e0 + j < e1 ? e0 + j : (e1 > 0 ? e1 - 1 : 0);
                const float *p = a.stage + size_t(e) * 3;
                r[j] = make_float3(FIN_LOAD(p), FIN_LOAD(p + 1), FIN_LOAD(p + 2));
            }
#pragma unroll
            for (int j = 0; j < kRowsAhead; ++j)
                if (e0 + j < e1) {
                    gx += r[j].x;
                    gy += r[j].y;
                    gz += r[j].z;
                }
        }
        for (int32_t e = e0 + kRowsAhead; e < e1; ++e) {
            const float *r = a.stage + size_t(e) * 3;
            gx += FIN_LOAD(r);
            gy += FIN_LOAD(r + 1);
            gz += FIN_LOAD(r + 2);
        }
        float *g = a.grad + size_t(FIN_LOAD(&a.fin_vid[k])) * 3;
        g[0] = gx * gscale;
        g[1] = gy * gscale;
        g[2] = gz * gscale;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(const float *in, const float *scalar, float *out, int64_t n)
{
    const float s = *scalar;
    if (in == out && s == 1.f) return;  // in-place by exactly 1 (the usual autograd grad_output): nothing to do
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

// ---- AdamUniform (reference utils/optimizer.py:38-89), two passes, no host sync ----
// pass 1: moments in place (optimizer.py:61-62) + max g2 and max |g1| (the two global maxima the reference
//         takes with .max(), :74 and :84, are monotone in the bias-corrected values)
__global__ __launch_bounds__(256) void adam_moments_kernel(const float *grad, float *g1, float *g2, int64_t n, float b1,
                                                           float b2, unsigned int *ws)
{
    __shared__ float red[2 * (256 / kWave)];
    float mx2 = 0.f, mx1 = 0.f;
    const int64_t stride = int64_t(gridDim.x) * 256, gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
    auto upd = [&](float g, float &a, float &b) {
        a = a * b1 + g * (1.f - b1);
        b = b * b2 + (g * g) * (1.f - b2);
        mx1 = fmaxf(mx1, fabsf(a));
        mx2 = fmaxf(mx2, b);
    };
    // 16 B per lane (torch allocations are 16-B aligned; the tail and odd alignments take the scalar loop)
    const bool vec = ((reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(g1) | reinterpret_cast<uintptr_t>(g2)) & 15) == 0;
    const int64_t n4 = vec ? n >> 2 : 0;
    for (int64_t i = gid; i < n4; i += stride) {
        const float4 g = reinterpret_cast<const float4 *>(grad)[i];
        float4 a = reinterpret_cast<float4 *>(g1)[i], b = reinterpret_cast<float4 *>(g2)[i];
        upd(g.x, a.x, b.x);
        upd(g.y, a.y, b.y);
        upd(g.z, a.z, b.z);
        upd(g.w, a.w, b.w);
        reinterpret_cast<float4 *>(g1)[i] = a;
        reinterpret_cast<float4 *>(g2)[i] = b;
    }
    for (int64_t i = 4 * n4 + gid; i < n; i += stride) {
        float a = g1[i], b = g2[i];
        upd(grad[i], a, b);
        g1[i] = a;
        g2[i] = b;
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        mx1 = fmaxf(mx1, __shfl_down(mx1, off, kWave));
        mx2 = fmaxf(mx2, __shfl_down(mx2, off, kWave));
    }
    const int wave = threadIdx.x / kWave;
    if (threadIdx.x % kWave == 0) {
        red[2 * wave] = mx1;
        red[2 * wave + 1] = mx2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; ++w) {
            mx1 = fmaxf(mx1, red[2 * w]);
            mx2 = fmaxf(mx2, red[2 * w + 1]);
        }
        atomicMax(ws, __float_as_uint(mx1));      // non-negative floats order like their bit patterns
        atomicMax(ws + 1, __float_as_uint(mx2));
    }
}

// pass 2: gr = m1 / (1e-8 + max sqrt(m2)) (:74), optional max-abs clamp to `limit` (:84-86), p -= lr * gr (:88)
__global__ __launch_bounds__(256) void adam_apply_kernel(float *p, const float *g1, int64_t n, float lr, float bias1,
                                                         float bias2, float limit, const unsigned int *ws)
{
    const float mx1 = __uint_as_float(ws[0]) / bias1;             // max |m1|
    const float denom = 1e-8f + sqrtf(__uint_as_float(ws[1]) / bias2);
    float scale = 1.f / denom;
    const float s = mx1 / denom;                                  // max |gr|
    if (limit > 0.f && s > limit) scale *= limit / s;
    const float k = lr * scale / bias1;
    const int64_t stride = int64_t(gridDim.x) * 256, gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g1)) & 15) == 0;
    const int64_t n4 = vec ? n >> 2 : 0;
    for (int64_t i = gid; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4 *>(p)[i];
        const float4 a = reinterpret_cast<const float4 *>(g1)[i];
        v.x -= k * a.x;
        v.y -= k * a.y;
        v.z -= k * a.z;
        v.w -= k * a.w;
        reinterpret_cast<float4 *>(p)[i] = v;
    }
    for (int64_t i = 4 * n4 + gid; i < n; i += stride) p[i] -= k * g1[i];
}

// Grid of the kernels that end in one atomicMax per workgroup (adam_moments_kernel, absmax_kernel): few, long-running workgroups.
// The atomics of all workgroups hit the same two words and serialise: 4 096 workgroups cost AdamUniform.step 0.127 ms at 12.3 M
// elements, 512 cost 0.065 (6.0 TB/s algorithmic), 256 0.063, 16 384 0.305; 0.705 against 0.802 ms at 120 M elements.
constexpr int kReduceGridCap = 512;

int grid_for(int64_t n, int per_block, int cap)
{
    int64_t b = (n + per_block - 1) / per_block;
    return int(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

hipError_t launch_eval_kernels(const EvalArgs &e, hipStream_t stream, hipEvent_t *ev);

namespace {

// The lane layouts tile kernels are built for: (slots per lane, block-size cap, waves per SIMD of the launch bounds).
//   2 x 768 @ 6 : 80 VGPRs, two workgroups of <= 80 KiB per CU -- the default, every operator variant
//   3 x 512 @ 4 : 128 VGPRs, two workgroups per CU of fewer, fatter waves (built-in operator)
//   4 x 768 @ 3 : 168 VGPRs, one workgroup per CU with up to 160 KiB: whole small tet-spheres as ONE tile, no halo, no
//   3 x 1024 @ 4: 128 VGPRs, the same with more waves                   shared vertices (built-in operator)
#define TSAMD_FN(...) reinterpret_cast<const void *>(&tile_energy_kernel<__VA_ARGS__>)
const void *tile_kernel_for(int spt, int block_threads, bool grad, bool weighted, bool rebuild)
{
    if (spt == 2 && block_threads <= 768) {
        if (weighted) return grad ? TSAMD_FN(true, 768, 6, true, false, 2) : TSAMD_FN(false, 768, 6, true, false, 2);
        if (rebuild) return grad ? TSAMD_FN(true, 768, 6, false, true, 2) : TSAMD_FN(false, 768, 6, false, true, 2);
        return grad ? TSAMD_FN(true, 768, 6, false, false, 2) : TSAMD_FN(false, 768, 6, false, false, 2);
    }
    if (weighted || rebuild) return nullptr;
    if (spt == 3 && block_threads <= 512) return grad ? TSAMD_FN(true, 512, 4, false, false, 3) : TSAMD_FN(false, 512, 4, false, false, 3);
    if (spt == 4 && block_threads <= 768) return grad ? TSAMD_FN(true, 768, 3, false, false, 4) : TSAMD_FN(false, 768, 3, false, false, 4);
    if (spt == 3 && block_threads <= 1024) return grad ? TSAMD_FN(true, 1024, 4, false, false, 3) : TSAMD_FN(false, 1024, 4, false, false, 3);
    return nullptr;
}
#undef TSAMD_FN

}  // namespace

bool lane_layout_supported(int spt, int max_threads) { return tile_kernel_for(spt, max_threads, true, false, false) != nullptr; }

hipError_t configure_kernels(int lds_bytes)
{
    // hipFuncAttributeMaxDynamicSharedMemorySize is per function, i.e. per process and device -- not per handle: only
    // ever raise it, or a handle with small tiles would lower the limit under an earlier handle with large ones
    static int configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds_bytes <= configured[dev]) return hipSuccess;
    // every tile kernel a plan can reach
    static const int layouts[][2] = {{2, 768}, {3, 512}, {4, 768}, {3, 1024}};
    for (const auto &lay : layouts)
        for (int variant = 0; variant < 6; ++variant) {
            const void *fn = tile_kernel_for(lay[0], lay[1], (variant & 1) != 0, variant / 2 == 1, variant / 2 == 2);
            if (!fn) continue;
            // lds_at() addresses the dynamic LDS array by absolute byte address: that is only right while the array starts
            // at LDS address 0, i.e. while the kernel has no static LDS object in front of it (a static __shared__ variable,
            // an LDS-using helper that was not inlined, a compiler-generated module-LDS block)
            hipFuncAttributes attr;
            hipError_t e = hipFuncGetAttributes(&attr, fn);
            if (e != hipSuccess) return e;
            if (attr.sharedSizeBytes != 0) return hipErrorInvalidDeviceFunction;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return e;
        }
    configured[dev] = lds_bytes;
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

namespace {

// What one evaluation launches: the tile kernel (if the plan has tiles) and then either the finish kernel (shared-vertex
// sums + the energy reduction in its workgroup 0) or, without a gradient, the energy reduction alone.  One description
// serves the stream launches below and the nodes of an EvalGraph.
struct LaunchRecipe {
    KernelArgs k;
    FinishArgs f;
    const void *tile_fn = nullptr, *finish_fn = nullptr;   // nullptr = not launched
    dim3 tile_grid, tile_block, finish_grid, finish_block;
    size_t tile_lds = 0;
};

void set_coefficients(KernelArgs &k, float c1, float c2)
{
    k.c1 = c1;
    k.c2 = c2;
    k.ratio = c2 / c1;   // (IEEE single-precision division, like the kernel's own: inf / NaN for c1 == 0 are handled there)
}

hipError_t make_recipe(const EvalArgs &e, LaunchRecipe &r)
{
    if (e.n_tiles > 0) {
        KernelArgs &k = r.k;
        k.tiles = e.tiles;
        k.blob = e.blob;
        k.gvid = e.gvid;
        k.vdst = e.vdst;
        k.x = e.x;
        k.grad_out = e.grad_out;
        k.grad = e.grad;
        k.stage = e.stage;
        k.partials = e.partials;
        set_coefficients(k, e.c1, e.c2);
        k.coef = e.coef;
        k.order = e.order;
        k.n_tiles = int(e.n_tiles);
        k.tiles_per_xcd = int((e.n_tiles + 7) / 8);
        k.vert_stride = e.vert_stride;
        k.n_planes = e.n_planes;
        r.tile_fn = tile_kernel_for(e.spt, e.block_threads, e.grad != nullptr, e.weighted, e.rebuild);
        if (!r.tile_fn) return hipErrorInvalidConfiguration;
        r.tile_block = dim3(unsigned(e.block_threads));
        r.tile_grid = dim3(unsigned(8 * k.tiles_per_xcd));
        r.tile_lds = size_t(e.lds_bytes);
    }
    FinishArgs &f = r.f;
    f.fin_vid = e.fin_vid;
    f.fin_off = e.fin_off;
    f.n_finish = e.grad ? e.n_finish : 0;
    f.stage = e.stage;
    f.grad = e.grad;
    f.grad_out = e.grad_out;
    f.partials = e.partials;
    f.n_tiles = e.n_tiles;
    f.c1 = e.c1;
    f.c2 = e.c2;
    f.coef = e.coef;
    f.energy = e.energy;
    f.terms = e.terms;
    f.energy_copy = nullptr;
    if (f.n_finish > 0) {
        // one vertex per thread: the per-vertex chain off[k] -> rows -> store is pure latency, so expose all of it
        r.finish_fn = reinterpret_cast<const void *>(&finish_kernel);
        r.finish_grid = dim3(1u + unsigned(grid_for(f.n_finish, 256, 1 << 20)));
        r.finish_block = dim3(256);
    } else if (f.energy) {
        r.finish_fn = reinterpret_cast<const void *>(&energy_reduce_kernel);
        r.finish_grid = dim3(1);
        r.finish_block = dim3(1024);
    }
    return hipSuccess;
}

}  // namespace

hipError_t launch_eval_kernels(const EvalArgs &e, hipStream_t stream, hipEvent_t *ev)
{
    LaunchRecipe r;
    hipError_t err = make_recipe(e, r);
    if (err != hipSuccess) return err;
    if (r.tile_fn) {
        void *argv[] = {&r.k};
        err = hipLaunchKernel(r.tile_fn, r.tile_grid, r.tile_block, argv, r.tile_lds, stream);
        if (err != hipSuccess) return err;
    }
    if (ev) {
        err = hipEventRecord(ev[1], stream);
        if (err != hipSuccess) return err;
    }
    if (r.finish_fn) {
        void *argv[] = {&r.f};
        err = hipLaunchKernel(r.finish_fn, r.finish_grid, r.finish_block, argv, 0, stream);
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}

// ---- the evaluation as an explicit HIP graph (tile node -> finish node) ----
// The coefficients are kernel ARGUMENTS of both nodes: a launch updates them with hipGraphExecKernelNodeSetParams --
// host-side bookkeeping, nothing on the GPU's timeline -- so a replay follows the reference's coefficient schedule
// (energies/smooth_barrier.py:47-58) without a device-side coefficient buffer and its per-step 8-byte copy (measured
// round 3: 64 x kuhn8 24.3 us per replayed step with that copy against 15.6 us with constant coefficients).
struct EvalGraph {
    LaunchRecipe r;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t tile_node = nullptr, finish_node = nullptr;
    hipKernelNodeParams tile_p, finish_p;
    void *tile_argv[1], *finish_argv[1];
    float *own_grad = nullptr;   // the gradient buffer the graph was created with (a launch may name another one)
};

hipError_t eval_graph_create(const EvalArgs &e, EvalGraph **out)
{
    *out = nullptr;
    EvalGraph *g = new EvalGraph();
    hipError_t err = make_recipe(e, g->r);
    g->own_grad = e.grad;
    auto fail = [&](hipError_t code) {
        eval_graph_destroy(g);
        return code;
    };
    if (err != hipSuccess) return fail(err);
    if ((err = hipGraphCreate(&g->graph, 0)) != hipSuccess) return fail(err);
    if (g->r.tile_fn) {
        g->tile_argv[0] = &g->r.k;
        hipKernelNodeParams &p = g->tile_p;
        p = hipKernelNodeParams{};
        p.func = const_cast<void *>(g->r.tile_fn);
        p.gridDim = g->r.tile_grid;
        p.blockDim = g->r.tile_block;
        p.sharedMemBytes = unsigned(g->r.tile_lds);
        p.kernelParams = g->tile_argv;
        p.extra = nullptr;
        if ((err = hipGraphAddKernelNode(&g->tile_node, g->graph, nullptr, 0, &p)) != hipSuccess) return fail(err);
    }
    if (g->r.finish_fn) {
        g->finish_argv[0] = &g->r.f;
        hipKernelNodeParams &p = g->finish_p;
        p = hipKernelNodeParams{};
        p.func = const_cast<void *>(g->r.finish_fn);
        p.gridDim = g->r.finish_grid;
        p.blockDim = g->r.finish_block;
        p.sharedMemBytes = 0;
        p.kernelParams = g->finish_argv;
        p.extra = nullptr;
        if ((err = hipGraphAddKernelNode(&g->finish_node, g->graph, g->tile_node ? &g->tile_node : nullptr, g->tile_node ? 1 : 0, &p)) !=
            hipSuccess)
            return fail(err);
    }
    if ((err = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0)) != hipSuccess) return fail(err);
    *out = g;
    return hipSuccess;
}

hipError_t eval_graph_launch(EvalGraph *g, float c1, float c2, hipStream_t stream, float *energy_copy, float *grad)
{
    hipError_t err;
    const bool coef = g->r.k.c1 != c1 || g->r.k.c2 != c2 || g->r.f.c1 != c1 || g->r.f.c2 != c2;
    // (the gradient's address is an argument of both nodes, the energy copy's of the finish node -- per-launch, like the coefficients)
    if (!grad) grad = g->own_grad;
    if (grad != g->r.f.grad && !g->own_grad) return hipErrorInvalidValue;   // (a graph created without a gradient has no backward kernels)
    const bool moved = grad != g->r.f.grad;
    if (coef || moved) {
        set_coefficients(g->r.k, c1, c2);
        g->r.f.c1 = c1;
        g->r.f.c2 = c2;
        g->r.k.grad = grad;
        if (g->tile_node && (err = hipGraphExecKernelNodeSetParams(g->exec, g->tile_node, &g->tile_p)) != hipSuccess) return err;
    }
    if (coef || moved || g->r.f.energy_copy != energy_copy) {
        if (energy_copy && (!g->finish_node || !g->r.f.energy)) return hipErrorInvalidValue;   // (no node reduces the energy in this graph)
        g->r.f.energy_copy = energy_copy;
        g->r.f.grad = grad;
        if (g->finish_node && (err = hipGraphExecKernelNodeSetParams(g->exec, g->finish_node, &g->finish_p)) != hipSuccess) return err;
    }
    return hipGraphLaunch(g->exec, stream);
}

void eval_graph_destroy(EvalGraph *g)
{
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

// ---- n optimisation steps as one graph (trainer.py:128-133 folded: energy + backward, optimizer.step) ----
// Every step is four kernel nodes in a chain -- tile, finish, adam_moments, adam_apply -- and the steps are chained to each
// other through the parameter (apply of step k writes what tile of step k + 1 reads).  Nothing per step on the host: one launch
// per n steps, the schedule (coefficients, order, bias corrections, step limit) goes in as node arguments beforehand.
struct TrainLoopGraph {
    struct Adam {
        const float *grad;
        float *g1, *g2, *p;
        int64_t n;
        float b1, b2, lr, bias1, bias2, limit;
        unsigned int *ws;
    };
    int n = 0;
    std::vector<LaunchRecipe> r;
    std::vector<Adam> adam;
    std::vector<hipGraphNode_t> tile, finish, moments, apply;
    std::vector<hipKernelNodeParams> tile_p, finish_p, moments_p, apply_p;
    std::vector<void *> argv;     // [n][1 + 1 + 7 + 8]
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int adam_grid = 0;
};

hipError_t train_loop_create(const EvalArgs &e, float *param, float *g1, float *g2, int64_t n_param, void *ws, int n_iters, TrainLoopGraph **out)
{
    *out = nullptr;
    if (n_iters < 1 || n_iters > 4096 || n_param < 1 || !e.grad || !e.energy) return hipErrorInvalidValue;
    TrainLoopGraph *g = new TrainLoopGraph();
    auto fail = [&](hipError_t code) {
        train_loop_destroy(g);
        return code;
    };
    g->n = n_iters;
    g->r.resize(n_iters);
    g->adam.resize(n_iters);
    g->tile.assign(n_iters, nullptr);
    g->finish.assign(n_iters, nullptr);
    g->moments.assign(n_iters, nullptr);
    g->apply.assign(n_iters, nullptr);
    g->tile_p.resize(n_iters);
    g->finish_p.resize(n_iters);
    g->moments_p.resize(n_iters);
    g->apply_p.resize(n_iters);
    g->argv.resize(size_t(n_iters) * 17);
    g->adam_grid = grid_for(n_param, 1024, kReduceGridCap);
    hipError_t err;
    if ((err = hipGraphCreate(&g->graph, 0)) != hipSuccess) return fail(err);
    // the optimiser's two-dword scratch of every step, zeroed once per launch
    hipGraphNode_t prev = nullptr;
    {
        hipMemsetParams m = {};
        m.dst = ws;
        m.elementSize = 4;
        m.width = size_t(2 * n_iters);
        m.height = 1;
        m.value = 0;
        if ((err = hipGraphAddMemsetNode(&prev, g->graph, nullptr, 0, &m)) != hipSuccess) return fail(err);
    }
    for (int k = 0; k < n_iters; ++k) {
        EvalArgs ek = e;
        ek.x = param;
        ek.energy = e.energy + k;
        if ((err = make_recipe(ek, g->r[k])) != hipSuccess) return fail(err);
        LaunchRecipe &r = g->r[k];
        void **av = g->argv.data() + size_t(k) * 17;
        auto add = [&](hipGraphNode_t *node, hipKernelNodeParams &p, const void *fn, dim3 grid, dim3 block, size_t lds, void **args) {
            p = hipKernelNodeParams{};
            p.func = const_cast<void *>(fn);
            p.gridDim = grid;
            p.blockDim = block;
            p.sharedMemBytes = unsigned(lds);
            p.kernelParams = args;
            p.extra = nullptr;
            const hipError_t rc = hipGraphAddKernelNode(node, g->graph, prev ? &prev : nullptr, prev ? 1 : 0, &p);
            if (rc == hipSuccess) prev = *node;
            return rc;
        };
        if (r.tile_fn) {
            av[0] = &r.k;
            if ((err = add(&g->tile[k], g->tile_p[k], r.tile_fn, r.tile_grid, r.tile_block, r.tile_lds, av)) != hipSuccess) return fail(err);
        }
        if (r.finish_fn) {
            av[1] = &r.f;
            if ((err = add(&g->finish[k], g->finish_p[k], r.finish_fn, r.finish_grid, r.finish_block, 0, av + 1)) != hipSuccess) return fail(err);
        }
        TrainLoopGraph::Adam &a = g->adam[k];
        a = TrainLoopGraph::Adam{e.grad, g1, g2, param, n_param, 0.9f, 0.999f, 0.f, 1.f, 1.f, -1.f, static_cast<unsigned int *>(ws) + 2 * k};
        void **am = av + 2, **aa = av + 9;
        am[0] = &a.grad, am[1] = &a.g1, am[2] = &a.g2, am[3] = &a.n, am[4] = &a.b1, am[5] = &a.b2, am[6] = &a.ws;
        aa[0] = &a.p, aa[1] = &a.g1, aa[2] = &a.n, aa[3] = &a.lr, aa[4] = &a.bias1, aa[5] = &a.bias2, aa[6] = &a.limit, aa[7] = &a.ws;
        if ((err = add(&g->moments[k], g->moments_p[k], reinterpret_cast<const void *>(&adam_moments_kernel), dim3(unsigned(g->adam_grid)), dim3(256), 0,
                       am)) != hipSuccess)
            return fail(err);
        if ((err = add(&g->apply[k], g->apply_p[k], reinterpret_cast<const void *>(&adam_apply_kernel), dim3(unsigned(g->adam_grid)), dim3(256), 0, aa)) !=
            hipSuccess)
            return fail(err);
    }
    if ((err = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0)) != hipSuccess) return fail(err);
    *out = g;
    return hipSuccess;
}

hipError_t train_loop_launch(TrainLoopGraph *g, const TrainLoopStep *steps, float lr, float b1, float b2, hipStream_t stream)
{
    hipError_t err;
    for (int k = 0; k < g->n; ++k) {
        LaunchRecipe &r = g->r[k];
        const TrainLoopStep &s = steps[k];
        if (r.k.c1 != s.c1 || r.k.c2 != s.c2 || r.k.order != s.order || r.f.c1 != s.c1 || r.f.c2 != s.c2) {
            set_coefficients(r.k, s.c1, s.c2);
            r.f.c1 = s.c1;
            r.f.c2 = s.c2;
            r.k.order = s.order;
            if (g->tile[k] && (err = hipGraphExecKernelNodeSetParams(g->exec, g->tile[k], &g->tile_p[k])) != hipSuccess) return err;
            if (g->finish[k] && (err = hipGraphExecKernelNodeSetParams(g->exec, g->finish[k], &g->finish_p[k])) != hipSuccess) return err;
        }
        TrainLoopGraph::Adam &a = g->adam[k];
        if (a.b1 != b1 || a.b2 != b2) {
            a.b1 = b1;
            a.b2 = b2;
            if ((err = hipGraphExecKernelNodeSetParams(g->exec, g->moments[k], &g->moments_p[k])) != hipSuccess) return err;
        }
        if (a.lr != lr || a.bias1 != s.bias1 || a.bias2 != s.bias2 || a.limit != s.limit) {
            a.lr = lr;
            a.bias1 = s.bias1;
            a.bias2 = s.bias2;
            a.limit = s.limit;
            if ((err = hipGraphExecKernelNodeSetParams(g->exec, g->apply[k], &g->apply_p[k])) != hipSuccess) return err;
        }
    }
    return hipGraphLaunch(g->exec, stream);
}

void train_loop_destroy(TrainLoopGraph *g)
{
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

hipError_t launch_scale(const float *in, const float *scalar, float *out, int64_t n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scale_kernel, dim3(unsigned(grid_for(n, 1024, 2048))), dim3(256), 0, stream, in, scalar, out, n);
    return hipGetLastError();
}

hipError_t launch_adam_uniform(float *p, const float *grad, float *g1, float *g2, int64_t n, float lr, float b1, float b2,
                               float bias1, float bias2, float limit, void *workspace, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    unsigned int *ws = static_cast<unsigned int *>(workspace);
    hipError_t e = hipMemsetAsync(ws, 0, 2 * sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    const int g = grid_for(n, 1024, kReduceGridCap);
    hipLaunchKernelGGL(adam_moments_kernel, dim3(unsigned(g)), dim3(256), 0, stream, grad, g1, g2, n, b1, b2, ws);
    hipLaunchKernelGGL(adam_apply_kernel, dim3(unsigned(g)), dim3(256), 0, stream, p, g1, n, lr, bias1, bias2, limit, ws);
    return hipGetLastError();
}

hipError_t launch_grad_limit(float *grad, int64_t n, float thr, float s, void *workspace, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    unsigned int *mx = static_cast<unsigned int *>(workspace);
    hipError_t e = hipMemsetAsync(mx, 0, sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    const int g = grid_for(n, 1024, kReduceGridCap);
    hipLaunchKernelGGL(absmax_kernel, dim3(unsigned(g)), dim3(256), 0, stream, grad, n, mx);
    hipLaunchKernelGGL(clamp_kernel, dim3(unsigned(g)), dim3(256), 0, stream, grad, n, mx, thr, s);
    return hipGetLastError();
}

}  // namespace tsamd
