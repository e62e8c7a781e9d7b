// Host-side plan of the STREAMING tile kernels (round 3, experimental; DESIGN.md section 8, profiles/r03_experiments.md).
//
// The blob tiling of plan.h keeps a whole tile (owned tets + one-ring halo) in LDS; 22 % of its slots are halo.  Here a
// tet-sphere is cut into a few TUBES, and a tube is swept level by level: levels = breadth-first distance over face
// adjacency inside the tube (owned tets + the side halo), so a tet's neighbours are always within +-1 level.
// Consecutive levels are merged into BANDS of <= kBand slots.  One workgroup sweeps one tube with a rolling window of band
// records in LDS -- F of the last four bands, H of four, the per-tet vertex forces of two -- and four wave groups that
// run four stages on four different bands between two barriers:
//
//   step s:   group 0  stream + pass 1 (F, det, penalty)        on band s       (+ prefetch of band s+1)
//             group 1  pass 2 (H = L F, 1/2 |H|^2)              on band s - 2
//             group 2  pass 3 (P = L^T H + penalty, d = P Dm^-T) on band s - 4
//             group 3  per-vertex partial sums of the forces     of band s - 5
//
// The sweep direction needs no halo; only a tube's sides do: 1.078 slots per tet on kuhn19 instead of 1.284.  Vertices live
// in a position ring whose slots are assigned here (a slot is reused once the last band of its previous vertex has left
// stage 3); every live vertex has a 12-byte accumulator next to its position, written out when its last band is summed.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "plan.h"

namespace tsamd {

constexpr int kBand = 256;        // slots per band = lanes of one stage group (4 waves)
constexpr int kFRing = 4;         // bands of F alive at once (pass 2 of band s-2 reads s-3 .. s-1 while pass 1 writes s)
constexpr int kHRing = 4;         // bands of H alive at once
constexpr int kDRing = 2;         // bands of per-tet vertex forces alive at once
constexpr int kScalRing = 6;      // bands of the per-slot penalty factor (written in pass 1, read in pass 3, 4 steps later)
constexpr int kLagP2 = 2, kLagP3 = 4, kLagSum = 5;   // stage lags in steps behind pass 1
constexpr int kVertexReuseGap = kLagSum + 2;          // a ring slot is reused by a vertex entering >= this many bands after its predecessor's last band
constexpr int kMaxVertexSlots = 1024;

// per band, inside the tube's blob (all offsets in bytes from the tube's blob start)
struct StreamBandDesc {
    uint32_t planes_off;   // 13 planes of kBand dwords: lv01, lv23, nb01, nb23, dminv[9]
    uint32_t enter_off;    // n_enter x {u32 ring slot, i32 global vertex}: vertices whose first band this is
    uint32_t pairs_off;    // n_pairs x StreamPair: the vertices this band's tets touch
    uint32_t chunks_off;   // incidence chunks of 4 x u16 entries (lane << 2 | local vertex; padding = kBand << 2)
    uint16_t n_slots;      // real slots (owned first, then halo); the rest of the band is inert padding
    uint16_t n_owned;
    uint16_t n_enter;
    uint16_t n_pairs;
};
static_assert(sizeof(StreamBandDesc) == 24, "StreamBandDesc layout is part of the kernel ABI");

struct StreamPair {
    uint16_t vslot;        // position / accumulator ring slot of the vertex
    uint16_t n_chunks;     // bit 15: this is the vertex' LAST band (write the sum out); bit 14: shared with another tube (-> staging row)
    uint32_t first_chunk;  // index into the band's chunk array
    int32_t out_row;       // last band only: global vertex id (exclusive) or staging row (shared)
};
static_assert(sizeof(StreamPair) == 12, "StreamPair layout is part of the kernel ABI");
constexpr uint16_t kPairLast = 0x8000u, kPairShared = 0x4000u;

struct StreamTubeDesc {
    uint64_t blob_off;     // byte offset of the tube's blob: n_bands StreamBandDesc, then the bands' data
    int32_t n_bands;
    int32_t n_vslots;      // position-ring slots this tube uses
    int32_t n_owned, n_slots;
    int32_t reserved[2];
};
static_assert(sizeof(StreamTubeDesc) == 32, "StreamTubeDesc layout is part of the kernel ABI");

// Neighbour token (16 bits, two per dword; bit 15 of nb01's low half = owned): (band delta + 1) << 8 | lane
inline uint32_t stream_token(int delta, uint32_t lane) { return (uint32_t(delta + 1) << 8) | lane; }

struct StreamPlan {
    int64_t n = 0, m = 0, n_components = 0;
    std::vector<int32_t> nbr;
    std::vector<StreamTubeDesc> tubes;
    RawVector<uint32_t> blob;
    std::vector<int32_t> fin_vid, fin_off;   // shared vertex k = sum of staging rows [fin_off[k], fin_off[k+1])
    int64_t n_stage = 0;
    int64_t total_slots = 0, total_bands = 0, total_pairs = 0, total_chunks = 0;
    int32_t max_vslots = 0, max_bands = 0;
    // host-only: global tet id of every band slot (tube-major, band-major, kBand entries per band; -1 = padding)
    std::vector<int32_t> slot_tet;
    std::vector<int64_t> tube_band_base;     // per tube: index of its first band in slot_tet / kBand
};

// Returns 0 on success, otherwise a tsamd_status value with `err` filled in (ERR_TILING when a component cannot be cut
// into tubes whose widest level fits a band: callers fall back to the blob tiling).
int build_stream_plan(const float *rest, int64_t n, const int32_t *tets, int64_t m, int num_threads, StreamPlan &plan,
                      std::string &err);

}  // namespace tsamd
