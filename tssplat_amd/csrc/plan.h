// Host-side tiling plan for the fused tet-energy kernels (no HIP in this header).
//
// The reference builds two global sparse matrices on the CPU at construction
// (libpgo, /root/reference/tssplat_ext/tet_spheres/tet_spheres.cpp:140-159) and
// streams ~2 KB/tet of COO data through five SpMVs per evaluation.  We instead
// cut the tet mesh into LDS-sized *tiles* once, and per evaluation stream 52 B
// per tile slot exactly once; everything else (F, L F, L^T L F, the vertex
// accumulation) lives in LDS and registers.  See DESIGN.md.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

// (the one function the kernels share with the planner is marked for both sides when hipcc compiles this header)
#ifdef __HIPCC__
#define TSAMD_HOST_DEVICE __host__ __device__
#else
#define TSAMD_HOST_DEVICE
#endif

namespace tsamd {

struct PlanOptions {
    // Defaults = the measured optimum on MI355X for large batches: two 768-thread workgroups per CU, 2 tets per lane, 80 KiB of
    // LDS each.
    int lds_budget = 0;           // bytes of LDS one workgroup may use; 0 = 80 KiB
    int max_threads = 0;          // workgroup size cap (multiple of 64, <= kTileThreads); 0 = 768
    int target_owned = 0;         // 0 = auto
    int num_threads = 0;          // 0 = hardware concurrency
    int conflict_aware = 1;       // order the neighbour entries to dodge LDS bank conflicts
    int lane_search_sweeps = 2;   // with conflict_aware: sweeps of the lane-assignment search (conflict_opt.cpp), 0 = items stay in Morton order
    int rebuild_dminv = 0;        // 1 = do not stream Dm^-1 (36 of the 52 bytes per slot): keep each tile's REST positions
                                  // (16 B per tile vertex) and invert Dm in registers, in fp32 (see kPlanesRebuild)
    int slots_per_lane = 0;       // 0 = kSlotsPerLane (2); 3 and 4 select the kernels built for fewer, fatter waves
};

// Device-visible tile descriptor (48 bytes, uniform loads in the kernel).
struct TileDesc {
    uint64_t blob_off;   // byte offset of the tile's dword planes in the blob
    int32_t n_slots;     // owned + halo tets
    int32_t n_owned;
    int32_t s_pad;       // n_slots rounded up to a multiple of the lane layout (= plane length in dwords)
    int32_t n_verts;     // local vertices (a vertex met by more than kMaxRank slots of the tile counts once per kMaxRank)
    int32_t n_excl;      // how many of them belong to this tile alone (statistics: the kernels read the sign of vdst[])
    int32_t vert_off;    // offset into gvid[] / vdst[] (= tile * vert_stride)
    int64_t stage_off;   // host bookkeeping: the tile's first entry in fin_idx[]
    int32_t n_rows;      // rows of the tile's force array = the largest number of slots any of its vertices meets (<= kMaxRank)
    int32_t rec_base;    // LDS byte address of record 0 (tile_rec_base)
};
static_assert(sizeof(TileDesc) == 48, "TileDesc layout is part of the kernel ABI");

// Per tile, in this order, inside the blob:
//   13 planes of s_pad dwords : lv01, lv23, nb01, nb23, dminv[9]        (tet-slot major, 52 B / slot)
//   kRowTabEntries x u16      : row_start[r], r = 0 .. n_rows (the rest repeats the total): where row r of the tile's
//                               per-vertex force array begins, in 12-byte entries
//   (rebuild_dminv plans)     : one float4 per tile vertex, 16-byte aligned, behind the row table
// That is ALL an evaluation streams per slot: since round 5 there are no per-vertex incidence lists (8.6 bytes per slot until
// round 4).  The forces of a tile are summed per vertex through a "jagged diagonal" array instead: the tile's vertices are
// numbered by falling slot count, entry (v, r) -- the r-th slot that meets vertex v -- sits at row_start[r] + v, so row r is
// the contiguous run of the vertices met by more than r slots.  A slot SCATTERS its four corner forces (the rank r of each
// corner rides in the six spare bits of its 16-bit vertex field), a lane per vertex then walks the rows: consecutive lanes read
// consecutive 12-byte entries -- no list, no index arithmetic beyond one add, no bank conflict.
constexpr int kPlanes = 13;
constexpr int kRowTabEntries = 72;                 // u16 each; 144 bytes keep what follows 16-byte aligned
constexpr int kMaxRank = 64;                       // slots per (virtual) tile vertex: 6-bit rank
// Lane layout the default kernels are compiled for: two consecutive slots per lane (one 8-byte load per plane), workgroups of
// at most 768 threads (80-VGPR builds, also with an explicit operator): two workgroups per CU when a tile's LDS is <= 80 KiB.
constexpr int kSlotsPerLane = 2;
constexpr int kTileThreads = 768;
// Plans built with an explicit element operator (build_plan's `op`) carry 9 more fp32 planes per slot:
//   [13]      L[e, e]
//   [14..17]  L[e, n_k]   row weights, in the slot's (possibly re-ordered) neighbour order -- pass 2, H = L F
//   [18..21]  L[n_k, e]   column weights, same order                                     -- pass 3, Q = L^T H
// (0 where a face has no neighbour and the token points at the slot itself).  Without an operator the kernels use the uniform face-adjacency
// umbrella (diagonal = number of face neighbours, off-diagonals = -1) and these planes do not exist.
constexpr int kPlanesWeighted = 22;
// A SYMMETRIC operator (L[e, n_k] == L[n_k, e] for every face, after the rounding to fp32 -- e.g. the uniform umbrella passed as
// data, or any weighted graph Laplacian) needs the weights once: planes [13] and [14..17] only, 72 instead of 88 bytes per slot;
// pass 3 applies L^T = L with the row weights it already holds.
constexpr int kPlanesWeightedSym = 18;
// Plans built with rebuild_dminv carry only the four index planes (16 bytes per slot instead of 52) and, behind the
// row table, one float4 per tile vertex with its REST position; the kernels stage those next
// to the current positions and rebuild Dm^-1 = cofactor^T / det per slot in fp32 registers.  This is NOT bit-identical
// to the streamed operator (built in double and rounded to fp32 like the reference's matrices, tet_spheres.cpp:43-45):
// entries differ by <= 2.8e-7 relative (mean 2.7e-8) against 5.9e-8 (mean 1.1e-8) for the rounding itself, and the
// gradient moves by ~7e-8 relative (profiles/r02_experiments.md).  Not combined with an explicit operator.
constexpr int kPlanesRebuild = 4;
// Index planes, two 16-bit fields per dword:
//   lv01 = (v0 | r0 << 10) | (v1 | r1 << 10) << 16     v = local vertex (< 1024), r = rank of this slot among the slots that
//   lv23 = (v2 | r2 << 10) | (v3 | r3 << 10) << 16     meet v (< 64): the corner's force goes to entry row_start[r] + v
//   nb01 =  f0 | f1 << 16                              f = record_token(LDS record index of the face neighbour): a quarter of
//   nb23 =  f2 | f3 << 16                              the byte address of the record's ninth entry (16 bits: any LDS address)
// A face without a neighbour in the tile (mesh boundary; for halo slots: any neighbour that is not owned by the
// tile) points at the slot's OWN record: the kernels evaluate 4 * own - sum of the four, so such a face adds
// own - own = 0 and neither a degree nor a dummy record is needed.
// There is no "owned" bit any more: lane t's p-th slot is item p * nq + t of the tile (nq = s_pad / slots per lane) and the
// items below n_owned are the owned ones, so a lane compares its item number with the descriptor's n_owned.
constexpr uint32_t kVertMask = 0x3ffu;
constexpr int kRankShift = 10;
constexpr int kMaxTileVerts = 1023;
// LDS map of a tile (absolute byte addresses; the kernels' only LDS object is the dynamic array at address 0):
//   [0, 320)                 row table as u32 byte addresses (65 entries used)
//   [320, 320 + 16 VP)       staged positions, one float4 per local vertex (VP = vertices rounded up to 4)
//   (+ 16 VP)                rebuild_dminv plans: the staged rest positions
//   + 256                    reduction scratch
//   rec_base ...             one 48-byte record per slot: F, later H; then, over the same bytes, the force array
//                            (12 bytes per (vertex, slot) incidence = 48 per slot)
constexpr int kRowTabBytes = 320;
inline int64_t tile_rec_base(int64_t n_verts, bool rebuild_dminv = false)
{
    const int64_t vp = (n_verts + 3) & ~int64_t(3);
    return kRowTabBytes + (rebuild_dminv ? 32 : 16) * vp + 256;
}
// LDS record of index idx: 48 bytes at rec_base + 48 * idx = [tail quad | entries 0..3 | entries 4..7]; the ninth matrix entry
// sits in the tail quad at dword (idx >> 3) & 3 -- rotating it with bits 3-4 of the index spreads the 4-byte
// gathers of it over all 32 banks (at a fixed position the 48-byte stride folds them onto 8).  The token
// rec_base / 4 + 12 * idx + rot is what the planes store: token << 2 is the byte address of the ninth entry, and that address
// with its low four bits cleared is the record base -- one shift and one mask per gathered record.
inline uint32_t record_token(uint32_t idx, uint32_t rec_base) { return rec_base / 4u + 12u * idx + ((idx >> 3) & 3u); }
inline uint32_t token_record(uint32_t token, uint32_t rec_base) { return (token - rec_base / 4u) / 12u; }

// DEVICE IMAGE of a tile's planes.  Plan::blob (and tsamd_tile_view) hold the planes one after the other, s_pad dwords each, lane
// t's slots at dwords [spt t, spt t + spt).  The copy in HBM interleaves the planes that the kernels always load together --
// plane q with plane q ^ 1, for every q except the two odd ones out, 12 (the ninth entry of Dm^-1) and 13 (an explicit
// operator's diagonal) -- per lane: dwords [2 spt t, 2 spt t + spt) of the pair's range are plane q's, the next spt plane q + 1's.
// One 16-byte load per lane and pair instead of two 8-byte loads: the same bytes in half the requests (a CU's ingest rate is
// requests in flight / latency, tools/ubench_ingest.hip: 13.2 against 11.4 bytes per cycle and CU with two workgroups streaming).
// Byte ranges of planes, row table and rest positions inside the blob are the same in both layouts.
// Measured (round 6, tools/ab_variants.py base pair16, same box): tile kernel 512 x kuhn19 0.3688 -> 0.3639 ms, a.veg x 952 0.4102 ->
// 0.3974 ms, bit-identical results.  (-DTSAMD_PAIRED_PLANES=0 builds the plane-after-plane image of rounds 1-5 for an A/B: the
// one laboratory hook in the product sources, a build-time layout constant; tools/lab_variants.py `unpaired`.)
#ifndef TSAMD_PAIRED_PLANES
#define TSAMD_PAIRED_PLANES 1
#endif
TSAMD_HOST_DEVICE constexpr bool planes_paired(int spt) { return TSAMD_PAIRED_PLANES != 0 && (spt == 2 || spt == 4); }
TSAMD_HOST_DEVICE constexpr bool plane_has_partner(int q, int n_planes) { return q != 12 && q != 13 && (q | 1) < n_planes; }
// dst = device image of one tile's planes (n_planes * s_pad dwords) from the host layout src
inline void interleave_tile_planes(const uint32_t *src, uint32_t *dst, int n_planes, int64_t s_pad, int spt)
{
    for (int q = 0; q < n_planes; ++q) {
        if (!plane_has_partner(q, n_planes)) {
            for (int64_t i = 0; i < s_pad; ++i) dst[q * s_pad + i] = src[q * s_pad + i];
            continue;
        }
        const int q0 = q & ~1, half = q & 1;
        for (int64_t i = 0; i < s_pad; ++i) dst[q0 * s_pad + 2 * spt * (i / spt) + half * spt + i % spt] = src[q * s_pad + i];
    }
}

// Where slot s (HBM plane order: thread t streams slots spt*t .. spt*t+spt-1 as one load per plane) lives in
// the LDS planes.  Lane t keeps its p-th slot at p * nq + t, so the 64 lanes of a wave touch 64
// consecutive float4 -- conflict-free -- instead of a 64 B stride (4-way bank conflict, measured:
// 70 % of all LDS cycles).  Neighbour entries in the blob hold these LDS indices (as tokens).
inline int32_t lds_index(int32_t slot, int32_t nq, int32_t spt) { return (slot % spt) * nq + slot / spt; }

// LDS bytes the kernels carve for a tile with padded slot count s_pad and n_verts vertices.
inline int64_t tile_lds_bytes(int64_t s_pad, int64_t n_verts, bool rebuild_dminv = false)
{
    return tile_rec_base(n_verts, rebuild_dminv) + 48 * s_pad;
}

// Byte offsets, inside a tile's blob, of the row table and of the float4 rest positions of a rebuild_dminv plan.
inline int64_t tile_rowtab_offset(int64_t n_planes, int64_t s_pad) { return n_planes * s_pad * 4; }
inline int64_t tile_rest_offset(int64_t n_planes, int64_t s_pad) { return tile_rowtab_offset(n_planes, s_pad) + 2 * kRowTabEntries; }

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the plan's big arrays (hundreds of MB
// at 21 M tets) are then first touched -- and zero-filled where needed -- by the worker threads that fill them, not by
// one serial memset inside build_plan (1.8 s of a 5 s build at 5 M tets).
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    template <class U, class... Args>
    void construct(U *p, Args &&...args)
    {
        if constexpr (sizeof...(Args) == 0)
            ::new (static_cast<void *>(p)) U;   // default-init: no zero fill
        else
            ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};
template <class T>
using RawVector = std::vector<T, DefaultInitAllocator<T>>;

struct Plan {
    int64_t n = 0, m = 0;
    int64_t n_components = 0;
    std::vector<int32_t> nbr;        // 4 per tet, -1 = boundary
    std::vector<TileDesc> tiles;
    RawVector<uint32_t> blob;        // all tiles' planes
    RawVector<int32_t> gvid;         // all tiles' local->global vertex ids
    RawVector<int32_t> vdst;         // same layout: where the tile's sum for that vertex goes -- >= 0: row of grad (the vertex
                                     // belongs to this tile alone), < 0: staging row ~vdst (summed by the finish kernel)
    RawVector<int32_t> slot_tet;     // per tile s_pad entries, global tet id or -1 (host only)
    std::vector<int64_t> slot_base;  // per tile offset into slot_tet
    // finish vertex k (global id fin_vid[k]) = sum of staging rows [fin_off[k], fin_off[k+1]);
    // fin_idx[tile.stage_off + j] = staging row the tile's j-th shared vertex writes to
    std::vector<int32_t> fin_vid, fin_off, fin_idx;
    int64_t n_stage = 0;             // rows in the staging buffer
    int64_t total_slots = 0, total_tile_verts = 0;
    int32_t max_slots = 0, max_verts = 0, block_threads = 64, lds_bytes = 0, spt = kSlotsPerLane;
    int32_t n_planes = kPlanes;      // kPlanes, or kPlanesWeighted when an explicit operator was given
    int32_t vert_stride = 64;        // gvid entries per tile (tile t's ids start at t * vert_stride = its TileDesc::vert_off)
    std::vector<float> op_diag;      // explicit operator only: L[e,e] per tet
    std::vector<float> op_w;         // explicit operator only: L[e, nbr[4e+k]] per tet face (0 on boundary faces)
};

// Element operator L (m x m) in CSR over tets.  Its sparsity must lie inside "diagonal + face adjacency":
// any other nonzero entry is rejected.  This is how the reference's true operator -- libpgo's
// pgo_create_tet_biharmonic_gradient_matrix(geo, faceNeighbor=1, scale=0), tet_spheres.cpp:148, whose
// source is not part of the reference -- or any variant of it (e.g. the row-scaled scale=1 form) is
// substituted for the uniform umbrella this library assumes by default.
struct ElementOperatorCSR {
    const int64_t *rowptr;  // m + 1
    const int32_t *col;
    const double *val;
};

// Returns 0 on success, otherwise a tsamd_status value with `err` filled in.
int build_plan(const float *rest, int64_t n, const int32_t *tets, int64_t m, const PlanOptions &opt,
               Plan &plan, std::string &err, const ElementOperatorCSR *op = nullptr);

// nbr[4e+k] = tet across the face of e opposite local vertex k, -1 on the boundary; a face shared by more
// than two tets is an error.  nthreads <= 1 runs serially.
int build_adjacency(const int32_t *tets, int64_t n, int64_t m, std::vector<int32_t> &nbr, int nthreads,
                    std::string &err);

// Minimal Vega .veg reader (*VERTICES / *ELEMENTS TET), 0-based output.
int read_veg(const char *path, std::vector<float> &rest, std::vector<int32_t> &tets, std::string &err);

}  // namespace tsamd
