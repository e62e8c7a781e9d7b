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

}  // namespace tsamd
