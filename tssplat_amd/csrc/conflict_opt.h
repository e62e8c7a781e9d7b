// Neighbour-gather step assignment against LDS bank conflicts (see conflict_opt.cpp).
#pragma once

#include <cstdint>
#include <vector>

namespace tsamd {

// Step assignment of one ds_read_b128 lane group: lane li (nl <= 16 of them) reads the records cand[li][0..3]
// (zs = the all-zero slot: any number of lanes may read it in one step), one per step, in an order that is free.
// Two DIFFERENT records of one step collide when they agree modulo 16.  Exact where possible: a bipartite
// multigraph lanes x columns whose nodes have at most 4 edges has a proper 4-edge-colouring (Koenig), found with
// alternating paths; columns with more than 4 reads are split into virtual columns first, which puts the
// unavoidable d - 4 extra cycles on them and nothing else.  from[li][step] = index into cand[li].
void colour_group_reads(int nl, const uint32_t cand[][4], uint32_t zs, int from[][4]);

// Local repair of the step assignment of one half-wave (n <= 32 lanes, group[li] in {0, 1} = its ds_read_b128 group) on the true
// price.  The colouring is exact for the graph it sees, but (i) a column read d > 4 times costs d - 4 extra cycles wherever its
// surplus reads land -- two columns' surplus reads in the SAME step cost one cycle together, in different steps two; (ii) lanes that
// read the same record in the same step are served by one access (broadcast); (iii) the ninth dword of every record is read by a
// ds_read_b32 whose 32 lanes -- both groups -- collide on the record index modulo 32.  Hill climbing over "lane li swaps the reads
// of two of its steps" (always legal) on the LDS cycles of the steps (2 x fullest column per group + fullest bank of the half-wave,
// distinct records), ties broken towards even loads.  from[li][step] is updated in place.
void repair_half_wave_steps(int n, const uint8_t *group, const uint32_t cand[][4], int from[][4]);

// Lane assignment of a tile's items against the same conflicts (round 5; the colouring above then orders each lane's four reads).
// Item i of a tile is LDS record i and is read by lane (i % nq) of its workgroup at position i / nq; the 16 lanes of a ds_read_b128
// group hold records of 16 different columns (i mod 16) and every item reads four records (nb[4 * i + k]; its own where a face has no
// usable neighbour), so a group's 64 reads meet the 16 columns four times each AT BEST: the colouring serves a column read d times
// in max(d, 4) cycles.  Which item sits on which lane of its group is free.  Local search over swaps of two items of one group
// (both owned or both halo: items < n_owned stay below n_owned) on the overflow sum((d - 4)^2 over columns with d > 4), counted for
// pass 2 (owned readers) and pass 3 (all readers), b128 columns (mod 16, per 16-lane group) twice as heavy as the banks of the
// rotated ninth dword (mod 32, per half-wave).  Deterministic.  item_at[i] = the item that moves to position i.
void search_lane_assignment(int n, int n_owned, int nq, const int32_t *nb, int sweeps, std::vector<int32_t> &item_at);

}  // namespace tsamd
