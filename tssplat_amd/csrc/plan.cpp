// Host-side construction of the tiling plan: face adjacency, connected
// components (= tet-spheres), recursive coordinate bisection into LDS-sized
// tiles with a one-ring face halo, per-tile local indexing, and the staging /
// finish lists for vertices that more than one tile touches.
//
// Replaces the role of libpgo in the reference's constructor
// (/root/reference/tssplat_ext/tet_spheres/tet_spheres.cpp:140-159): there the
// rest mesh becomes two global COO matrices, here it becomes per-tile planes
// of Dm^-1 (double -> fp32, as tet_spheres.cpp:43-45 rounds the matrix values)
// plus 16-bit local vertex / neighbour indices.  Pure C++17, no HIP.
#include "plan.h"

#include "conflict_opt.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <numeric>
#include <sstream>
#include <chrono>
#include <thread>

namespace tsamd {
namespace {

// TSAMD_PLAN_TIMING=1 prints the wall time of every stage of build_plan to stderr (tuning aid)
struct StageTimer {
    bool on = std::getenv("TSAMD_PLAN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[plan] %-28s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

enum { OK = 0, ERR_INVALID = 1, ERR_BAD_MESH = 2, ERR_IO = 5, ERR_TILING = 6 };

// ---- tiny work-sharing helper: fn(begin, end, worker) over [0, n) in dynamic chunks ----
template <class Fn>
void parallel_chunks(int64_t n, int64_t chunk, int nthreads, Fn fn)
{
    if (n <= 0) return;
    nthreads = std::max(1, nthreads);
    if (nthreads == 1 || n <= chunk) {
        fn(int64_t(0), n, 0);
        return;
    }
    std::atomic<int64_t> next{0};
    auto body = [&](int worker) {
        for (;;) {
            int64_t b = next.fetch_add(chunk, std::memory_order_relaxed);
            if (b >= n) break;
            fn(b, std::min(n, b + chunk), worker);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(body, t);
    body(0);
    for (auto &th : pool) th.join();
}

inline void sort3(uint32_t &a, uint32_t &b, uint32_t &c)
{
    if (a > b) std::swap(a, b);
    if (b > c) std::swap(b, c);
    if (a > b) std::swap(a, b);
}

struct BucketFace {
    uint32_t b, c, slot;  // the smallest vertex is the bucket id; slot = 4*tet + opposite local vertex
};

}  // namespace

// nbr[4e+k] = tet across the face of e opposite local vertex k, -1 on the boundary.
int build_adjacency(const int32_t *tets, int64_t n, int64_t m, std::vector<int32_t> &nbr, int nthreads,
                    std::string &err)
{
    static const int opp[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
    nbr.assign(size_t(4 * m), -1);
    // bucket the 4 m faces by their smallest vertex: count, prefix sum, fill -- counting and filling in parallel with
    // atomic per-bucket cursors (the order inside a bucket is arbitrary here; every bucket is sorted below)
    std::vector<std::atomic<int32_t>> count(static_cast<size_t>(n));
    parallel_chunks(n, 1 << 16, nthreads, [&](int64_t b, int64_t e, int) {
        for (int64_t v = b; v < e; ++v) count[size_t(v)].store(0, std::memory_order_relaxed);
    });
    parallel_chunks(m, 1 << 15, nthreads, [&](int64_t eb, int64_t ee, int) {
        for (int64_t e = eb; e < ee; ++e) {
            const int32_t *t = tets + 4 * e;
            for (int k = 0; k < 4; ++k) {
                uint32_t a = t[opp[k][0]], b = t[opp[k][1]], c = t[opp[k][2]];
                sort3(a, b, c);
                count[a].fetch_add(1, std::memory_order_relaxed);
            }
        }
    });
    std::vector<int64_t> start(size_t(n + 1), 0);
    for (int64_t v = 0; v < n; ++v) start[v + 1] = start[v] + count[size_t(v)].load(std::memory_order_relaxed);
    RawVector<BucketFace> faces(size_t(4 * m));
    parallel_chunks(n, 1 << 16, nthreads, [&](int64_t b, int64_t e, int) {
        for (int64_t v = b; v < e; ++v) count[size_t(v)].store(0, std::memory_order_relaxed);
    });
    parallel_chunks(m, 1 << 15, nthreads, [&](int64_t eb, int64_t ee, int) {
        for (int64_t e = eb; e < ee; ++e) {
            const int32_t *t = tets + 4 * e;
            for (int k = 0; k < 4; ++k) {
                uint32_t a = t[opp[k][0]], b = t[opp[k][1]], c = t[opp[k][2]];
                sort3(a, b, c);
                const int64_t pos = start[a] + count[a].fetch_add(1, std::memory_order_relaxed);
                faces[size_t(pos)] = BucketFace{b, c, uint32_t(4 * e + k)};
            }
        }
    });
    std::atomic<int> bad{0};
    parallel_chunks(n, 4096, nthreads, [&](int64_t vb, int64_t ve, int) {
        for (int64_t v = vb; v < ve; ++v) {
            BucketFace *f0 = faces.data() + start[v], *f1 = faces.data() + start[v + 1];
            std::sort(f0, f1, [](const BucketFace &x, const BucketFace &y) {
                return x.b != y.b ? x.b < y.b : (x.c != y.c ? x.c < y.c : x.slot < y.slot);
            });
            for (BucketFace *f = f0; f + 1 < f1; ++f) {
                if (f->b == f[1].b && f->c == f[1].c) {
                    if (f + 2 < f1 && f[2].b == f->b && f[2].c == f->c) {
                        bad.store(1);
                        return;
                    }
                    nbr[f->slot] = int32_t(f[1].slot >> 2);
                    nbr[f[1].slot] = int32_t(f->slot >> 2);
                    ++f;
                }
            }
        }
    });
    if (bad.load()) {
        err = "non-manifold tet mesh: a face is shared by more than two tets";
        return ERR_BAD_MESH;
    }
    return OK;
}

namespace {

// per-worker scratch with O(1) reset through stamps
struct Scratch {
    std::vector<int32_t> tet_stamp, tet_slot, vert_stamp, vert_local;
    int32_t stamp = 0;
    void init(int64_t m, int64_t n)
    {
        if (int64_t(tet_stamp.size()) != m) {
            tet_stamp.assign(size_t(m), 0);
            tet_slot.assign(size_t(m), 0);
        }
        if (int64_t(vert_stamp.size()) != n) {
            vert_stamp.assign(size_t(n), 0);
            vert_local.assign(size_t(n), 0);
        }
    }
    int32_t next()
    {
        if (stamp > std::numeric_limits<int32_t>::max() - 8) {
            std::fill(tet_stamp.begin(), tet_stamp.end(), 0);
            std::fill(vert_stamp.begin(), vert_stamp.end(), 0);
            stamp = 0;
        }
        stamp += 2;
        return stamp;  // `stamp` marks owned, `stamp+1` marks halo
    }
};

struct Limits {
    int64_t budget;
    int64_t max_spad;
    int64_t pad_unit = 4;
    bool rebuild = false;
    bool fits(int64_t n_slots, int64_t n_verts) const
    {
        const int64_t sp = (n_slots + pad_unit - 1) / pad_unit * pad_unit;
        return sp <= max_spad && n_verts <= kMaxTileVerts && tile_lds_bytes(sp, n_verts, rebuild) <= budget;
    }
};

struct Mesh {
    const float *rest;
    const int32_t *tets;
    const int32_t *nbr;
    int64_t n, m;
};

// owned + one-ring halo size and the number of tile vertices they touch (a vertex met by more than kMaxRank slots of
// the tile is split into several tile vertices of at most kMaxRank slots each, see build_plan)
void measure(const Mesh &M, const int32_t *owned, int64_t cnt, Scratch &S, int64_t &n_slots, int64_t &n_verts,
             std::vector<int32_t> *halo_out = nullptr)
{
    const int32_t so = S.next(), sh = so + 1;
    for (int64_t i = 0; i < cnt; ++i) S.tet_stamp[owned[i]] = so;
    int64_t halo = 0, verts = 0;
    if (halo_out) halo_out->clear();
    auto touch = [&](int32_t e) {
        for (int a = 0; a < 4; ++a) {
            int32_t v = M.tets[4 * int64_t(e) + a];
            if (S.vert_stamp[v] != so) {
                S.vert_stamp[v] = so;
                S.vert_local[v] = 1;
                ++verts;
            } else if (S.vert_local[v]++ % kMaxRank == 0) {
                ++verts;
            }
        }
    };
    for (int64_t i = 0; i < cnt; ++i) {
        const int32_t e = owned[i];
        touch(e);
        for (int k = 0; k < 4; ++k) {
            int32_t q = M.nbr[4 * int64_t(e) + k];
            if (q < 0) continue;
            int32_t st = S.tet_stamp[q];
            if (st == so || st == sh) continue;
            S.tet_stamp[q] = sh;
            ++halo;
            if (halo_out) halo_out->push_back(q);
            touch(q);
        }
    }
    n_slots = cnt + halo;
    n_verts = verts;
}

inline uint32_t spread10(uint32_t v)
{
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ff;
    v = (v | (v << 8)) & 0x0300f00f;
    v = (v | (v << 4)) & 0x030c30c3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

struct Splitter {
    const Mesh &M;
    const Limits &lim;
    const std::vector<float> &cen;  // 3 per tet
    Scratch &S;
    std::vector<std::vector<int32_t>> &out;
    std::string &err;
    int rc = OK;
    bool strict = false;  // strict: a leaf that does not fit aborts the attempt (caller retries with more parts)
    bool failed = false;

    void bbox(const int32_t *ids, int64_t cnt, float lo[3], float hi[3]) const
    {
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::numeric_limits<float>::max();
            hi[d] = -std::numeric_limits<float>::max();
        }
        for (int64_t i = 0; i < cnt; ++i)
            for (int d = 0; d < 3; ++d) {
                float c = cen[3 * size_t(ids[i]) + d];
                lo[d] = std::min(lo[d], c);
                hi[d] = std::max(hi[d], c);
            }
    }

    void emit(int32_t *ids, int64_t cnt)
    {
        // order the leaf along a Morton curve so that lanes of a wave hold nearby tets
        float lo[3], hi[3];
        bbox(ids, cnt, lo, hi);
        std::vector<std::pair<uint32_t, int32_t>> key(static_cast<size_t>(cnt));
        for (int64_t i = 0; i < cnt; ++i) {
            uint32_t code = 0;
            for (int d = 0; d < 3; ++d) {
                float ext = hi[d] - lo[d];
                float u = ext > 0 ? (cen[3 * size_t(ids[i]) + d] - lo[d]) / ext : 0.f;
                code |= spread10(uint32_t(std::min(1023.f, u * 1023.f))) << d;
            }
            key[size_t(i)] = {code, ids[i]};
        }
        std::sort(key.begin(), key.end());
        std::vector<int32_t> t(static_cast<size_t>(cnt));
        for (int64_t i = 0; i < cnt; ++i) t[size_t(i)] = key[size_t(i)].second;
        out.push_back(std::move(t));
    }

    void split(int32_t *ids, int64_t cnt, int64_t k)
    {
        if (rc || failed) return;
        if (k <= 1) {
            int64_t ns, nv;
            measure(M, ids, cnt, S, ns, nv);
            if (lim.fits(ns, nv)) {
                emit(ids, cnt);
                return;
            }
            if (strict) {
                failed = true;
                return;
            }
            if (cnt <= 1) {
                err = "a single tet with its face neighbours exceeds the LDS budget";
                rc = ERR_TILING;
                return;
            }
            k = 2;
        }
        float lo[3], hi[3];
        bbox(ids, cnt, lo, hi);
        int ax = 0;
        for (int d = 1; d < 3; ++d)
            if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
        const int64_t k1 = k / 2;
        int64_t mid = cnt * k1 / k;
        mid = std::max<int64_t>(1, std::min(cnt - 1, mid));
        std::nth_element(ids, ids + mid, ids + cnt, [&](int32_t a, int32_t b) {
            float ca = cen[3 * size_t(a) + ax], cb = cen[3 * size_t(b) + ax];
            return ca != cb ? ca < cb : a < b;
        });
        split(ids, mid, k1);
        split(ids + mid, cnt - mid, k - k1);
    }
};

}  // namespace

int build_plan(const float *rest, int64_t n, const int32_t *tets, int64_t m, const PlanOptions &opt, Plan &P,
               std::string &err, const ElementOperatorCSR *op)
{
    StageTimer timer;
    if (n < 0 || m < 0 || (n > 0 && !rest) || (m > 0 && !tets)) {
        err = "null pointer or negative size";
        return ERR_INVALID;
    }
    if (n >= (int64_t(1) << 31) / 3 || m >= (int64_t(1) << 29)) {
        err = "mesh too large for 32-bit indexing";
        return ERR_INVALID;
    }
    for (int64_t i = 0; i < 4 * m; ++i)
        if (tets[i] < 0 || tets[i] >= n) {
            err = "tet index out of range at flat position " + std::to_string(i);
            return ERR_INVALID;
        }
    timer.lap("index check");
    // Default: all cores up to 32.  More did not help where it was measured (256-core EPYC 9575F host of the MI355X box, 21 M
    // tets): 32 threads 2.8 s, 64 threads 3.4 s, 128 asked (= 64) 3.4 s -- the bisection and pass A get slower, pass B (planes,
    // colouring, incidence matching: 1.0-1.1 s) does not get faster.  An explicit num_threads is honoured up to 64.
    int nthreads = opt.num_threads > 0 ? std::min(opt.num_threads, 64) : std::min(int(std::thread::hardware_concurrency()), 32);
    nthreads = std::max(1, nthreads);
    const int spt = opt.slots_per_lane > 0 ? opt.slots_per_lane : kSlotsPerLane;
    if (spt < 2 || spt > 4) {
        err = "slots_per_lane must be 2, 3 or 4";
        return ERR_INVALID;
    }
    // Defaults: 768 threads x 2 slots and 80 KiB, two workgroups per CU -- with or without an explicit operator (its nine
    // extra planes live in registers, not in LDS, and the kernel still fits 80 VGPRs).  (Which block sizes go with which
    // lane layout is the launcher's business: capi.cpp.)
    if (opt.max_threads > 1024) {
        err = "max_threads exceeds 1024";
        return ERR_INVALID;
    }
    int max_threads = opt.max_threads > 0 ? opt.max_threads : kTileThreads;
    max_threads = std::max(64, (max_threads / 64) * 64);
    Limits lim;
    lim.budget = opt.lds_budget > 0 ? opt.lds_budget : 80 * 1024;
    lim.pad_unit = spt == 3 ? 12 : 4;
    lim.max_spad = int64_t(spt) * int64_t(max_threads) / lim.pad_unit * lim.pad_unit;
    lim.rebuild = opt.rebuild_dminv != 0 && op == nullptr;
    if (lim.budget < tile_lds_bytes(12, 8, lim.rebuild)) {
        err = "lds_budget_bytes too small";
        return ERR_INVALID;
    }

    P = Plan();
    P.n = n;
    P.m = m;
    P.spt = spt;
    int rc = build_adjacency(tets, n, m, P.nbr, nthreads, err);
    if (rc) return rc;
    Mesh M{rest, tets, P.nbr.data(), n, m};
    timer.lap("face adjacency");

    // ---- explicit element operator: CSR -> (diagonal, one weight per tet face) ----
    if (op) {
        if (!op->rowptr || (m > 0 && op->rowptr[m] > 0 && (!op->col || !op->val))) {
            err = "element operator: null CSR array";
            return ERR_INVALID;
        }
        if (op->rowptr[0] != 0 || op->rowptr[m] < 0) {
            err = "element operator: rowptr[0] must be 0 and rowptr[m] non-negative";
            return ERR_INVALID;
        }
        if (opt.rebuild_dminv) {
            err = "element operator: not combined with rebuild_dminv (the explicit-operator kernels stream Dm^-1)";
            return ERR_INVALID;
        }
        P.n_planes = kPlanesWeighted;
        P.op_diag.assign(size_t(m), 0.f);
        P.op_w.assign(size_t(4 * m), 0.f);
        std::vector<double> dg(static_cast<size_t>(m), 0.0), w(static_cast<size_t>(4 * m), 0.0);
        for (int64_t e = 0; e < m; ++e) {
            if (op->rowptr[e + 1] < op->rowptr[e]) {
                err = "element operator: rowptr is not monotone";
                return ERR_INVALID;
            }
            for (int64_t q = op->rowptr[e]; q < op->rowptr[e + 1]; ++q) {
                const int64_t j = op->col[q];
                const double v = op->val[q];
                if (!std::isfinite(v)) {
                    err = "element operator: non-finite value in row " + std::to_string(e);
                    return ERR_INVALID;
                }
                if (j == e) {
                    dg[size_t(e)] += v;
                    continue;
                }
                int k = -1;
                for (int f = 0; f < 4; ++f)
                    if (j >= 0 && P.nbr[4 * size_t(e) + f] == j) {
                        k = f;
                        break;
                    }
                if (k >= 0) {
                    w[4 * size_t(e) + k] += v;
                } else if (v != 0.0) {
                    err = "element operator: entry (" + std::to_string(e) + ", " + std::to_string(j) +
                          ") is neither on the diagonal nor a face adjacency of the mesh";
                    return ERR_INVALID;
                }
            }
        }
        for (int64_t e = 0; e < m; ++e) P.op_diag[size_t(e)] = float(dg[size_t(e)]);   // double -> fp32, as the
        for (int64_t i = 0; i < 4 * m; ++i) P.op_w[size_t(i)] = float(w[size_t(i)]);   // reference rounds its matrices
        // symmetric in fp32?  then the column weights are the row weights and their four planes are not stored (kPlanesWeightedSym)
        bool symmetric = true;
        for (int64_t e = 0; e < m && symmetric; ++e)
            for (int k = 0; k < 4; ++k) {
                const int32_t q = P.nbr[4 * size_t(e) + k];
                if (q < 0) continue;
                float back = 0.f;
                for (int f = 0; f < 4; ++f)
                    if (P.nbr[4 * size_t(q) + f] == e) back = P.op_w[4 * size_t(q) + f];
                if (back != P.op_w[4 * size_t(e) + k]) {
                    symmetric = false;
                    break;
                }
            }
        if (symmetric) P.n_planes = kPlanesWeightedSym;
    }
    const bool weighted = op != nullptr;
    const bool rebuild = lim.rebuild;
    if (rebuild) P.n_planes = kPlanesRebuild;
    const int n_planes = P.n_planes;

    // ---- connected components over face adjacency (each tet-sphere is one) ----
    // Lock-free union-find over the face adjacency: the larger root is always linked under the smaller one, so a
    // component's root is its smallest tet id whatever the thread interleaving -- components are then numbered by that
    // id and list their tets in increasing order, exactly what the serial flood fill of rounds 1-2 produced.
    std::vector<int32_t> comp(size_t(m), -1);
    std::vector<int64_t> comp_start;
    RawVector<int32_t> comp_tets(static_cast<size_t>(m));
    {
        std::vector<std::atomic<int32_t>> parent(static_cast<size_t>(m));
        parallel_chunks(m, 1 << 16, nthreads, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i) parent[size_t(i)].store(int32_t(i), std::memory_order_relaxed);
        });
        auto find = [&](int32_t x) {
            for (;;) {
                const int32_t p = parent[size_t(x)].load(std::memory_order_relaxed);
                if (p == x) return x;
                const int32_t gp = parent[size_t(p)].load(std::memory_order_relaxed);
                if (gp != p) {   // path halving (a lost race only skips the shortcut)
                    int32_t expect = p;
                    parent[size_t(x)].compare_exchange_weak(expect, gp, std::memory_order_relaxed);
                }
                x = p;
            }
        };
        parallel_chunks(m, 1 << 15, nthreads, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i)
                for (int k = 0; k < 4; ++k) {
                    const int32_t q = P.nbr[4 * size_t(i) + k];
                    if (q < 0 || q > i) continue;            // every interior face once, from its larger tet
                    int32_t ra = find(int32_t(i)), rb = find(q);
                    while (ra != rb) {
                        int32_t hi = std::max(ra, rb), lo = std::min(ra, rb);
                        int32_t expect = hi;
                        if (parent[size_t(hi)].compare_exchange_strong(expect, lo, std::memory_order_relaxed)) break;
                        ra = find(hi);
                        rb = find(lo);
                    }
                }
        });
        // roots in increasing order = component numbers; tets of a component in increasing order (counting sort)
        std::vector<int32_t> root_comp(static_cast<size_t>(m), -1);
        RawVector<int32_t> root_of(static_cast<size_t>(m));
        parallel_chunks(m, 1 << 15, nthreads, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i) root_of[size_t(i)] = find(int32_t(i));
        });
        int64_t ncomp = 0;
        for (int64_t i = 0; i < m; ++i)
            if (root_of[size_t(i)] == i) root_comp[size_t(i)] = int32_t(ncomp++);
        std::vector<int64_t> fill(static_cast<size_t>(ncomp) + 1, 0);
        for (int64_t i = 0; i < m; ++i) {
            const int32_t c = root_comp[size_t(root_of[size_t(i)])];
            comp[size_t(i)] = c;
            ++fill[size_t(c) + 1];
        }
        for (int64_t c = 0; c < ncomp; ++c) fill[size_t(c) + 1] += fill[size_t(c)];
        comp_start.assign(fill.begin(), fill.end());
        for (int64_t i = 0; i < m; ++i) comp_tets[size_t(fill[size_t(comp[size_t(i)])]++)] = int32_t(i);
    }
    const int64_t C = int64_t(comp_start.size()) - 1;
    timer.lap("components");
    P.n_components = C;

    // ---- tet centroids (rest state) ----
    std::vector<float> cen(static_cast<size_t>(3 * m));
    parallel_chunks(m, 1 << 16, nthreads, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i)
            for (int d = 0; d < 3; ++d) {
                float s = 0.f;
                for (int a = 0; a < 4; ++a) s += rest[3 * size_t(tets[4 * i + a]) + d];
                cen[3 * size_t(i) + d] = 0.25f * s;
            }
    });

    std::vector<Scratch> scratch(static_cast<size_t>(nthreads));
    auto get_scratch = [&](int w) -> Scratch & {
        scratch[size_t(w)].init(m, n);
        return scratch[size_t(w)];
    };

    // ---- which components fit into one tile as they are (no halo)? ----
    std::vector<int64_t> comp_verts(static_cast<size_t>(C), 0);
    std::vector<uint8_t> comp_fits(static_cast<size_t>(C), 0);
    parallel_chunks(C, 16, nthreads, [&](int64_t b, int64_t e, int w) {
        Scratch &S = get_scratch(w);
        for (int64_t c = b; c < e; ++c) {
            int64_t cnt = comp_start[c + 1] - comp_start[c], ns, nv;
            if (cnt > lim.max_spad) continue;
            measure(M, comp_tets.data() + comp_start[c], cnt, S, ns, nv);
            comp_verts[size_t(c)] = nv;
            comp_fits[size_t(c)] = lim.fits(ns, nv) ? 1 : 0;
        }
    });

    timer.lap("centroids + fit check");
    // ---- group: pack small components together, bisect large ones ----
    // group g covers components [gb, ge); a group of one non-fitting component is bisected.
    struct Group {
        int64_t cb, ce;
        bool bisect;
    };
    std::vector<Group> groups;
    {
        int64_t c = 0;
        while (c < C) {
            if (!comp_fits[size_t(c)]) {
                groups.push_back({c, c + 1, true});
                ++c;
                continue;
            }
            int64_t tets_sum = 0, verts_sum = 0, ce = c;
            while (ce < C && comp_fits[size_t(ce)]) {
                int64_t t2 = tets_sum + (comp_start[ce + 1] - comp_start[ce]);
                int64_t v2 = verts_sum + comp_verts[size_t(ce)];  // upper bound on the union
                if (ce > c && !lim.fits(t2, v2)) break;
                tets_sum = t2;
                verts_sum = v2;
                ++ce;
            }
            groups.push_back({c, ce, false});
            c = ce;
        }
    }
    // LDS: 48 B per slot + 16 B per vertex at ~0.27 vertices per slot
    int64_t s_cap = std::min<int64_t>(lim.max_spad, (lim.budget - kRowTabBytes - 256) / (rebuild ? 58 : 53));
    int64_t target = opt.target_owned > 0 ? opt.target_owned : int64_t(0.70 * double(s_cap));
    target = std::max<int64_t>(1, target);

    std::vector<std::vector<std::vector<int32_t>>> group_tiles(groups.size());
    std::atomic<int> first_rc{0};
    std::string split_err;
    std::atomic<bool> err_set{false};
    parallel_chunks(int64_t(groups.size()), 1, nthreads, [&](int64_t b, int64_t e, int w) {
        Scratch &S = get_scratch(w);
        for (int64_t g = b; g < e; ++g) {
            const Group &G = groups[size_t(g)];
            int32_t *ids = comp_tets.data() + comp_start[G.cb];
            int64_t cnt = comp_start[G.ce] - comp_start[G.cb];
            if (!G.bisect) {
                group_tiles[size_t(g)].emplace_back(ids, ids + cnt);
                continue;
            }
            std::string local_err;
            Splitter sp{M, lim, cen, S, group_tiles[size_t(g)], local_err};
            // Fewest parts whose tiles all fit: start optimistic (owned ~ 0.85 of the slot capacity) and add
            // parts until no leaf overflows -- splitting an overflowing leaf in two would leave half-empty tiles.
            bool done = false;
            if (opt.target_owned <= 0) {
                int64_t k = std::max<int64_t>(2, (cnt + int64_t(0.85 * double(s_cap)) - 1) / int64_t(0.85 * double(s_cap)));
                for (int attempt = 0; attempt < 24 && !done; ++attempt) {
                    group_tiles[size_t(g)].clear();
                    sp.strict = true;
                    sp.failed = false;
                    sp.split(ids, cnt, k);
                    done = !sp.failed && !sp.rc;
                    k = std::max<int64_t>(k + 1, (k * 103 + 99) / 100);
                }
                if (!done) group_tiles[size_t(g)].clear();
            }
            sp.strict = false;
            sp.failed = false;
            if (!done) sp.split(ids, cnt, (cnt + target - 1) / target);
            if (sp.rc) {
                first_rc.store(sp.rc);
                bool expected = false;
                if (err_set.compare_exchange_strong(expected, true)) split_err = local_err;
            }
        }
    });
    if (first_rc.load()) {
        err = split_err;
        return first_rc.load();
    }

    timer.lap("bisection");
    std::vector<std::vector<int32_t>> tiles_owned;
    for (auto &gt : group_tiles)
        for (auto &t : gt) tiles_owned.push_back(std::move(t));
    group_tiles.clear();
    const int64_t T = int64_t(tiles_owned.size());

    // ---- pass A: per-tile halo + vertex lists, global per-vertex copy count ----
    // A tile vertex is (global vertex, copy): a vertex met by more than kMaxRank slots of the tile (a hub: the cone fixture's
    // centre meets ~1 500 slots of every tile) is split into copies of at most kMaxRank slots each, so that a slot's rank at a
    // corner fits the six spare bits of its vertex field; the copies are staged from the same position, their partial sums
    // go through the staging rows like any vertex shared by several tiles, and the finish kernel adds them up.
    // Tile vertices are numbered by FALLING slot count: row r of the tile's force array (plan.h) is then the prefix of the
    // vertices met by more than r slots, and the lanes of one wave of the per-vertex sum carry about the same number of rows.
    std::vector<std::vector<int32_t>> tile_halo(static_cast<size_t>(T)), tile_verts(static_cast<size_t>(T)), tile_vdeg(static_cast<size_t>(T));
    std::vector<std::atomic<int32_t>> vcount(static_cast<size_t>(n));
    for (auto &a : vcount) a.store(0, std::memory_order_relaxed);
    parallel_chunks(T, 4, nthreads, [&](int64_t b, int64_t e, int w) {
        Scratch &S = get_scratch(w);
        std::vector<int32_t> uniq, cnt;
        std::vector<std::pair<int32_t, int32_t>> key;   // (-slots, global vertex), in first-touch order before the sort
        for (int64_t t = b; t < e; ++t) {
            const auto &own = tiles_owned[size_t(t)];
            int64_t ns, nv;
            measure(M, own.data(), int64_t(own.size()), S, ns, nv, &tile_halo[size_t(t)]);
            uniq.clear();
            cnt.clear();
            const int32_t st = S.next();
            auto touch = [&](int32_t el) {
                for (int a = 0; a < 4; ++a) {
                    const int32_t v = tets[4 * int64_t(el) + a];
                    if (S.vert_stamp[v] != st) {
                        S.vert_stamp[v] = st;
                        S.vert_local[v] = int32_t(uniq.size());
                        uniq.push_back(v);
                        cnt.push_back(0);
                    }
                    ++cnt[size_t(S.vert_local[v])];
                }
            };
            for (int32_t el : own) touch(el);
            for (int32_t el : tile_halo[size_t(t)]) touch(el);
            key.clear();
            for (size_t i = 0; i < uniq.size(); ++i) {
                int32_t left = cnt[i], copies = 0;
                while (left > 0) {
                    const int32_t c = std::min<int32_t>(left, kMaxRank);
                    key.push_back({-c, uniq[i]});
                    left -= c;
                    ++copies;
                }
                vcount[size_t(uniq[i])].fetch_add(copies, std::memory_order_relaxed);
            }
            // (ties by global vertex id: the lanes that gather a tile's positions and store its gradient rows then walk runs of
            // consecutive rows of x / grad -- fewer memory transactions per wave instruction than in first-touch order)
            std::sort(key.begin(), key.end());
            auto &tv = tile_verts[size_t(t)];
            auto &td = tile_vdeg[size_t(t)];
            tv.resize(key.size());
            td.resize(key.size());
            for (size_t i = 0; i < key.size(); ++i) {
                tv[i] = key[i].second;
                td[i] = -key[i].first;
            }
        }
    });

    timer.lap("pass A (halo, vertex lists, vertex order)");
    // ---- offsets ----
    P.tiles.resize(size_t(T));
    P.slot_base.resize(size_t(T) + 1);
    int64_t blob_bytes = 0, vert_off = 0, stage_off = 0, slot_off = 0;
    int32_t max_quads = 1;
    // Every tile's vertex ids sit at tile * vert_stride: the kernel can issue the id load of the position gather -- the
    // head of its longest dependent chain (ids -> positions -> LDS) -- from the workgroup index alone, in parallel with
    // the tile descriptor's fetch instead of behind it (unused entries name vertex 0).
    int64_t vert_stride = 64;
    for (int64_t t = 0; t < T; ++t) vert_stride = std::max<int64_t>(vert_stride, (int64_t(tile_verts[size_t(t)].size()) + 63) & ~int64_t(63));
    P.vert_stride = int32_t(vert_stride);
    for (int64_t t = 0; t < T; ++t) {
        TileDesc &d = P.tiles[size_t(t)];
        std::memset(&d, 0, sizeof(d));
        auto &tv = tile_verts[size_t(t)];
        int32_t n_excl = 0;
        for (int32_t v : tv) n_excl += vcount[size_t(v)].load(std::memory_order_relaxed) == 1;
        d.n_owned = int32_t(tiles_owned[size_t(t)].size());
        d.n_slots = d.n_owned + int32_t(tile_halo[size_t(t)].size());
        d.s_pad = int32_t((d.n_slots + lim.pad_unit - 1) / lim.pad_unit * lim.pad_unit);
        d.n_verts = int32_t(tv.size());
        d.n_excl = n_excl;
        d.blob_off = uint64_t(blob_bytes);
        d.vert_off = int32_t(vert_off);
        d.stage_off = stage_off;
        d.n_rows = tv.empty() ? 0 : tile_vdeg[size_t(t)][0];
        d.rec_base = int32_t(tile_rec_base(d.n_verts, rebuild));
        if (d.n_verts > kMaxTileVerts || 4 * int64_t(d.s_pad) > 65535) {
            err = "tile exceeds the 10-bit vertex / 16-bit entry fields of the plan";
            return ERR_TILING;
        }
        blob_bytes += (tile_rest_offset(n_planes, d.s_pad) + (rebuild ? 16 * int64_t(d.n_verts) : 0) + 127) & ~int64_t(127);
        vert_off += vert_stride;
        P.total_tile_verts += d.n_verts;
        stage_off += d.n_verts - d.n_excl;
        P.slot_base[size_t(t)] = slot_off;
        slot_off += d.s_pad;
        P.total_slots += d.n_slots;
        P.max_slots = std::max(P.max_slots, d.n_slots);
        P.max_verts = std::max(P.max_verts, d.n_verts);
        P.lds_bytes = std::max<int32_t>(P.lds_bytes, int32_t(tile_lds_bytes(d.s_pad, d.n_verts, rebuild)));
        max_quads = std::max(max_quads, d.s_pad / spt);
        if (vert_off >= (int64_t(1) << 31)) {
            err = "too many tile vertices for 32-bit offsets";
            return ERR_TILING;
        }
    }
    P.slot_base[size_t(T)] = slot_off;
    P.n_stage = stage_off;
    P.block_threads = std::min(max_threads, ((max_quads + 63) / 64) * 64);
    P.blob.resize(size_t(blob_bytes / 4));   // (uninitialised: every tile zero-fills its own range in pass B)
    P.gvid.resize(size_t(vert_off));
    P.vdst.resize(size_t(vert_off));
    P.slot_tet.resize(size_t(slot_off));

    // ---- finish lists: every vertex with more than one tile-vertex copy; staging rows vertex-major, copies in tile order ----
    // (tile-major rows + a gather in the finish kernel was measured: tile kernel unchanged, finish kernel 0.046 -> 0.084 ms)
    std::vector<int32_t> fin_of(static_cast<size_t>(n), -1);
    {
        int64_t entries = 0;
        for (int64_t v = 0; v < n; ++v) {
            int32_t c = vcount[size_t(v)].load(std::memory_order_relaxed);
            if (c == 1) continue;
            fin_of[size_t(v)] = int32_t(P.fin_vid.size());
            P.fin_vid.push_back(int32_t(v));
            P.fin_off.push_back(int32_t(entries));
            entries += c;
            if (entries >= (int64_t(1) << 31)) {
                err = "too many shared vertex copies for 32-bit offsets";
                return ERR_TILING;
            }
        }
        P.fin_off.push_back(int32_t(entries));
        P.fin_idx.assign(size_t(entries), 0);
        std::vector<int32_t> cur(P.fin_off.begin(), P.fin_off.end() - 1);
        for (int64_t t = 0; t < T; ++t) {
            const TileDesc &d = P.tiles[size_t(t)];
            const auto &tv = tile_verts[size_t(t)];
            int64_t j = 0;
            int32_t *vd = P.vdst.data() + d.vert_off;
            for (int32_t i = 0; i < d.n_verts; ++i) {
                const int32_t v = tv[size_t(i)];
                const int32_t k = fin_of[size_t(v)];
                if (k < 0) {
                    vd[i] = v;
                } else {
                    const int32_t row = cur[size_t(k)]++;
                    vd[i] = ~row;
                    P.fin_idx[size_t(d.stage_off + j++)] = row;
                }
            }
            std::fill_n(vd + d.n_verts, size_t(P.vert_stride - d.n_verts), int32_t(0));
        }
    }
    timer.lap("offsets + allocation + finish lists");
    // ---- pass B: fill planes ----
    std::atomic<int> singular{0};
    parallel_chunks(T, 2, nthreads, [&](int64_t b, int64_t e, int w) {
        Scratch &S = get_scratch(w);
        std::vector<int32_t> next_rank, copy_of;      // per tile vertex: ranks handed out so far; next copy of the same vertex (-1: none)
        std::vector<int32_t> lane_nb, lane_item_at, lane_tets;
        for (int64_t t = b; t < e; ++t) {
            const TileDesc &d = P.tiles[size_t(t)];
            auto &own = tiles_owned[size_t(t)];
            auto &halo = tile_halo[size_t(t)];
            const auto &tv = tile_verts[size_t(t)];
            const auto &tdeg = tile_vdeg[size_t(t)];
            const int32_t st = S.next();
            // global vertex -> its first copy (the copies of a hub follow each other through copy_of, fullest first)
            copy_of.assign(size_t(d.n_verts), -1);
            next_rank.assign(size_t(d.n_verts), 0);
            for (int32_t i = d.n_verts - 1; i >= 0; --i) {
                const int32_t v = tv[size_t(i)];
                if (S.vert_stamp[v] == st) copy_of[size_t(i)] = S.vert_local[v];
                S.vert_stamp[v] = st;
                S.vert_local[v] = i;
                P.gvid[size_t(d.vert_off) + size_t(i)] = v;
            }
            std::fill_n(P.gvid.data() + d.vert_off + d.n_verts, size_t(P.vert_stride - d.n_verts), int32_t(0));   // unused entries: vertex 0
            const int32_t nq = d.s_pad / spt;
            const uint32_t RB = uint32_t(d.rec_base);
            // item L (owned tets first, Morton order each) -> slot: lane L % nq takes it as its (L / nq)-th slot, so that the owned
            // and the halo items are spread evenly over the lanes and `owned` is all but wave-uniform per position
            auto slot_of_item = [&](int32_t L) { return spt * (L % nq) + L / nq; };
            auto item_tet = [&](int32_t L) { return L < d.n_owned ? own[size_t(L)] : halo[size_t(L - d.n_owned)]; };
            for (int32_t L = 0; L < d.n_slots; ++L) {
                int32_t el = item_tet(L);
                S.tet_stamp[el] = st + (L < d.n_owned ? 0 : 1);
                S.tet_slot[el] = slot_of_item(L);
            }
            // neighbour k of item L as an item (= LDS record) of this tile; a face without a usable neighbour points at the item itself
            // (halo tets only look at owned neighbours; an owned tet's neighbours are owned or halo by construction)
            auto neighbour_item = [&](int32_t L, int k) {
                const int32_t q = P.nbr[4 * size_t(item_tet(L)) + k];
                if (q >= 0 && (L < d.n_owned || S.tet_stamp[q] == st)) return lds_index(S.tet_slot[q], nq, spt);
                return L;
            };
            if (opt.conflict_aware && opt.lane_search_sweeps > 0) {
                // ---- which item sits on which lane of its ds_read_b128 group: local search against bank conflicts (conflict_opt.cpp) ----
                lane_nb.resize(4 * size_t(d.n_slots));
                for (int32_t L = 0; L < d.n_slots; ++L)
                    for (int k = 0; k < 4; ++k) lane_nb[4 * size_t(L) + k] = neighbour_item(L, k);
                search_lane_assignment(d.n_slots, d.n_owned, nq, lane_nb.data(), opt.lane_search_sweeps, lane_item_at);
                lane_tets.resize(size_t(d.n_slots));
                for (int32_t L = 0; L < d.n_slots; ++L) lane_tets[size_t(L)] = item_tet(lane_item_at[size_t(L)]);
                std::copy(lane_tets.begin(), lane_tets.begin() + d.n_owned, own.begin());
                std::copy(lane_tets.begin() + d.n_owned, lane_tets.end(), halo.begin());
                for (int32_t L = 0; L < d.n_slots; ++L) S.tet_slot[item_tet(L)] = slot_of_item(L);   // (same tets, same stamps)
            }
            uint32_t *pl = P.blob.data() + d.blob_off / 4;
            {   // this tile's part of the (uninitialised) plan arrays
                const uint64_t blob_end = t + 1 < T ? P.tiles[size_t(t) + 1].blob_off : uint64_t(P.blob.size()) * 4;
                std::memset(pl, 0, size_t(blob_end - d.blob_off));
                std::fill_n(P.slot_tet.data() + P.slot_base[size_t(t)], size_t(d.s_pad), int32_t(-1));
            }
            // padding slots: lv = 0, neighbours = the slot itself, dminv = 0 (F = 0; they write no forces: the kernels stop at n_slots)
            for (int32_t s = 0; s < d.s_pad; ++s) {
                const uint32_t f = record_token(uint32_t(lds_index(s, nq, spt)), RB);
                pl[2 * size_t(d.s_pad) + s] = f | (f << 16);
                pl[3 * size_t(d.s_pad) + s] = f | (f << 16);
            }
            int32_t *stet = P.slot_tet.data() + P.slot_base[size_t(t)];
            for (int32_t L = 0; L < d.n_slots; ++L) {
                const int32_t el = item_tet(L);
                const int32_t s = slot_of_item(L);
                stet[s] = el;
                uint32_t nb[4];
                const uint32_t self = uint32_t(L);   // (= lds_index(s, nq, spt))
                for (int k = 0; k < 4; ++k) nb[k] = uint32_t(neighbour_item(L, k));
                if (weighted) {   // planes 13..21: L[e,e], L[e,n_k], L[n_k,e] in the (not yet re-ordered) neighbour order
                    auto putf = [&](int plane, float v) { std::memcpy(&pl[size_t(plane) * size_t(d.s_pad) + s], &v, 4); };
                    putf(13, P.op_diag[size_t(el)]);
                    for (int k = 0; k < 4; ++k) {
                        const int32_t q = P.nbr[4 * size_t(el) + k];
                        float wr = 0.f, wc = 0.f;
                        if (q >= 0 && nb[k] != self) {
                            wr = P.op_w[4 * size_t(el) + k];
                            for (int f = 0; f < 4; ++f)
                                if (P.nbr[4 * size_t(q) + f] == el) wc = P.op_w[4 * size_t(q) + f];
                        }
                        putf(14 + k, wr);
                        if (n_planes == kPlanesWeighted) putf(18 + k, wc);
                    }
                }
                pl[2 * size_t(d.s_pad) + s] = record_token(nb[0], RB) | (record_token(nb[1], RB) << 16);
                pl[3 * size_t(d.s_pad) + s] = record_token(nb[2], RB) | (record_token(nb[3], RB) << 16);
                // Dm^-1 in double from the fp32 rest positions, rounded to fp32
                const int32_t *tt = tets + 4 * int64_t(el);
                double D[9];
                for (int i = 0; i < 3; ++i)
                    for (int k = 0; k < 3; ++k)
                        D[3 * i + k] = double(rest[3 * size_t(tt[k + 1]) + i]) - double(rest[3 * size_t(tt[0]) + i]);
                double Cf[9];
                Cf[0] = D[4] * D[8] - D[5] * D[7];
                Cf[1] = D[5] * D[6] - D[3] * D[8];
                Cf[2] = D[3] * D[7] - D[4] * D[6];
                Cf[3] = D[2] * D[7] - D[1] * D[8];
                Cf[4] = D[0] * D[8] - D[2] * D[6];
                Cf[5] = D[1] * D[6] - D[0] * D[7];
                Cf[6] = D[1] * D[5] - D[2] * D[4];
                Cf[7] = D[2] * D[3] - D[0] * D[5];
                Cf[8] = D[0] * D[4] - D[1] * D[3];
                const double det = D[0] * Cf[0] + D[1] * Cf[1] + D[2] * Cf[2];
                if (det == 0.0 || !std::isfinite(det)) {
                    singular.store(1);
                    continue;
                }
                if (!rebuild)
                    for (int i = 0; i < 3; ++i)
                        for (int k = 0; k < 3; ++k) {
                            float v = float(Cf[3 * k + i] / det);  // inverse = cofactor^T / det
                            std::memcpy(&pl[size_t(4 + 3 * i + k) * size_t(d.s_pad) + s], &v, 4);
                        }
            }
            // ---- vertex fields: local vertex + the slot's rank at it ----
            // Which of its vertex's rows a (slot, corner) writes to is free -- it only fixes the order of the per-vertex sum -- and
            // decides the LDS bank of the scattered 12-byte entry: entry = row_start[rank] + vertex, bank of its first dword =
            // 3 * entry mod 32.  The 32 lanes of a half-wave that scatter corner k of their p-th slots in one instruction are
            // served conflict-free when their entries differ mod 32 (3 is invertible mod 32: the dwords 3e, 3e + 1 of a
            // ds_write2_b32 then load every bank exactly twice).  Handed out in slot order the entries collide 2.8x as often as
            // that (kuhn19 and a.veg alike) and the scatter is bound by exactly these conflicts (profiles/r05_experiments.md); so
            // per half-wave instruction a maximum matching lanes x residues (augmenting paths) picks, for every lane, one of the
            // still unused ranks of its vertex; a lane left over takes the unused rank whose residue is least loaded.
            {
                std::vector<uint16_t> row_start(kMaxRank + 1, 0);     // (the row table proper is written below, from the same degrees)
                {
                    int32_t start = 0, width = d.n_verts;
                    for (int32_t r = 0; r <= kMaxRank; ++r) {
                        row_start[size_t(r)] = uint16_t(start);
                        while (width > 0 && tdeg[size_t(width) - 1] <= r) --width;
                        start += width;
                    }
                }
                // (slot, corner) -> tile vertex (a hub's copies are filled in slot order), ranks to be chosen
                std::vector<int32_t> corner_vert(4 * size_t(d.s_pad), -1);
                std::vector<uint64_t> unused(size_t(d.n_verts), 0);
                for (int32_t i = 0; i < d.n_verts; ++i) unused[size_t(i)] = tdeg[size_t(i)] >= 64 ? ~uint64_t(0) : ((uint64_t(1) << tdeg[size_t(i)]) - 1);
                for (int32_t s = 0; s < d.s_pad; ++s) {
                    if (stet[s] < 0) continue;
                    for (int a = 0; a < 4; ++a) {
                        int32_t i = S.vert_local[tets[4 * int64_t(stet[s]) + a]];
                        while (next_rank[size_t(i)] >= tdeg[size_t(i)]) i = copy_of[size_t(i)];   // this copy is full: the hub's next one
                        ++next_rank[size_t(i)];
                        corner_vert[4 * size_t(s) + a] = i;
                    }
                }
                std::vector<uint8_t> corner_rank(4 * size_t(d.s_pad), 0);
                if (!opt.conflict_aware) {
                    for (int32_t s = 0; s < d.s_pad; ++s)
                        for (int a = 0; a < 4; ++a) {
                            const int32_t i = corner_vert[4 * size_t(s) + a];
                            if (i < 0) continue;
                            const int r = __builtin_ctzll(unused[size_t(i)]);
                            unused[size_t(i)] &= unused[size_t(i)] - 1;
                            corner_rank[4 * size_t(s) + a] = uint8_t(r);
                        }
                } else {
                    for (int32_t pp = 0; pp < spt; ++pp)
                        for (int a = 0; a < 4; ++a)
                            for (int32_t base = 0; base < nq; base += 32) {
                                int32_t nl = 0, lane_c[32];
                                for (int32_t tl = base; tl < std::min(base + 32, nq); ++tl) {
                                    const size_t c = 4 * size_t(spt * tl + pp) + size_t(a);
                                    if (corner_vert[c] >= 0) lane_c[nl++] = int32_t(c);
                                }
                                int32_t owner[32], pick[32];          // residue -> lane, lane -> rank
                                for (auto &o : owner) o = -1;
                                for (int32_t l = 0; l < nl; ++l) pick[l] = -1;
                                auto residue = [&](int32_t l, int r) { return (int32_t(row_start[size_t(r)]) + corner_vert[size_t(lane_c[l])]) & 31; };
                                struct Matcher {
                                    int32_t *owner, *pick;
                                    const int32_t *lane_c;
                                    const std::vector<int32_t> &corner_vert;
                                    const std::vector<uint64_t> &unused;
                                    const std::vector<uint16_t> &row_start;
                                    bool seen[32];
                                    bool aug(int32_t l)   // augmenting path from lane l (DFS over at most 32 residues)
                                    {
                                        const int32_t v = corner_vert[size_t(lane_c[l])];
                                        for (uint64_t m = unused[size_t(v)]; m; m &= m - 1) {
                                            const int r = __builtin_ctzll(m);
                                            const int32_t res = (int32_t(row_start[size_t(r)]) + v) & 31;
                                            if (seen[res]) continue;
                                            seen[res] = true;
                                            if (owner[res] < 0 || aug(owner[res])) {
                                                owner[res] = l;
                                                pick[l] = r;
                                                return true;
                                            }
                                        }
                                        return false;
                                    }
                                } M{owner, pick, lane_c, corner_vert, unused, row_start, {}};
                                for (int32_t l = 0; l < nl; ++l) {
                                    std::memset(M.seen, 0, sizeof(M.seen));
                                    M.aug(l);
                                }
                                // (two lanes of the same vertex matched to different residues hold different ranks: same vertex + same
                                // rank = same residue.)  Commit the matched lanes, then serve the others from what is left.
                                int32_t load[32] = {};
                                for (int32_t l = 0; l < nl; ++l)
                                    if (pick[l] >= 0) {
                                        unused[size_t(corner_vert[size_t(lane_c[l])])] &= ~(uint64_t(1) << pick[l]);
                                        ++load[residue(l, pick[l])];
                                    }
                                for (int32_t l = 0; l < nl; ++l) {
                                    if (pick[l] >= 0) continue;
                                    const int32_t v = corner_vert[size_t(lane_c[l])];
                                    int best = -1;
                                    for (uint64_t m = unused[size_t(v)]; m; m &= m - 1) {
                                        const int r = __builtin_ctzll(m);
                                        if (best < 0 || load[residue(l, r)] < load[residue(l, best)]) best = r;
                                    }
                                    pick[l] = best;
                                    unused[size_t(v)] &= ~(uint64_t(1) << best);
                                    ++load[residue(l, best)];
                                }
                                for (int32_t l = 0; l < nl; ++l) corner_rank[size_t(lane_c[l])] = uint8_t(pick[l]);
                            }
                }
                for (int32_t s = 0; s < d.s_pad; ++s) {
                    if (stet[s] < 0) continue;
                    uint32_t lv[4];
                    for (int a = 0; a < 4; ++a)
                        lv[a] = uint32_t(corner_vert[4 * size_t(s) + a]) | (uint32_t(corner_rank[4 * size_t(s) + a]) << kRankShift);
                    pl[0 * size_t(d.s_pad) + s] = lv[0] | (lv[1] << 16);
                    pl[1 * size_t(d.s_pad) + s] = lv[2] | (lv[3] << 16);
                }
            }
            // ---- row table: row r = the vertices met by more than r slots, a prefix of the (sorted) tile vertices ----
            {
                uint16_t *rt = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(pl) + tile_rowtab_offset(n_planes, d.s_pad));
                int32_t start = 0, width = d.n_verts;
                for (int32_t r = 0; r < kRowTabEntries; ++r) {
                    rt[r] = uint16_t(start);
                    while (width > 0 && tdeg[size_t(width) - 1] <= r) --width;
                    start += width;
                }
                if (rebuild) {   // the tile's rest positions, tile vertex order, one float4 each
                    float *rp = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(pl) + tile_rest_offset(n_planes, d.s_pad));
                    for (int32_t v = 0; v < d.n_verts; ++v) {
                        const int32_t gv = tv[size_t(v)];
                        rp[4 * v + 0] = rest[3 * size_t(gv) + 0];
                        rp[4 * v + 1] = rest[3 * size_t(gv) + 1];
                        rp[4 * v + 2] = rest[3 * size_t(gv) + 2];
                        rp[4 * v + 3] = 0.f;
                    }
                }
            }
            // ---- LDS bank-conflict-aware neighbour order ----
            // A wave reads neighbour k of 16 lanes' tets with one ds_read_b128 per 16-lane group; two lanes
            // collide when their records share a 16-byte bank column, i.e. when the record indices agree
            // mod 16 (48 B stride: column = 3 * idx mod 16).  The order of a tet's four neighbours is free:
            // every lane group gets a proper 4-edge-colouring of its lanes x columns read graph (conflict_opt.cpp).
            if (opt.conflict_aware) {
                static const int kGroups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                                   {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                                   {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
                uint32_t *p2 = pl + 2 * size_t(d.s_pad), *p3 = pl + 3 * size_t(d.s_pad);
                for (int32_t pp = 0; pp < spt; ++pp)
                    for (int32_t base = 0; base < nq; base += 64)
                        for (int hw = 0; hw < 2; ++hw) {
                            int32_t lane_slot[32];
                            uint32_t cand[32][4];
                            uint8_t group[32];
                            int from[32][4];   // step `step` of lane li reads candidate from[li][step] (the weights follow)
                            int nl = 0;
                            for (int gi = 2 * hw; gi < 2 * hw + 2; ++gi) {
                                const int first = nl;
                                for (int li = 0; li < 16; ++li) {
                                    const int32_t tl = base + kGroups[gi][li];
                                    if (tl >= nq) continue;
                                    const int32_t sl = spt * tl + pp;
                                    lane_slot[nl] = sl;
                                    group[nl] = uint8_t(gi & 1);
                                    cand[nl][0] = token_record(p2[sl] & 0xffffu, RB);
                                    cand[nl][1] = token_record(p2[sl] >> 16, RB);
                                    cand[nl][2] = token_record(p3[sl] & 0xffffu, RB);
                                    cand[nl][3] = token_record(p3[sl] >> 16, RB);
                                    ++nl;
                                }
                                colour_group_reads(nl - first, cand + first, 0xffffffffu, from + first);   // (no free reads: a missing face reads the slot itself)
                            }
                            repair_half_wave_steps(nl, group, cand, from);
                            for (int li = 0; li < nl; ++li) {
                                const int32_t sl = lane_slot[li];
                                uint32_t chosen[4];
                                for (int step = 0; step < 4; ++step) chosen[step] = cand[li][from[li][step]];
                                p2[sl] = record_token(chosen[0], RB) | (record_token(chosen[1], RB) << 16);
                                p3[sl] = record_token(chosen[2], RB) | (record_token(chosen[3], RB) << 16);
                                if (weighted)
                                    for (int base_plane : {14, 18}) {
                                        if (base_plane + 4 > n_planes) continue;   // (symmetric operator: no column-weight planes)
                                        uint32_t old[4];
                                        for (int k = 0; k < 4; ++k) old[k] = pl[size_t(base_plane + k) * size_t(d.s_pad) + sl];
                                        for (int k = 0; k < 4; ++k)
                                            pl[size_t(base_plane + k) * size_t(d.s_pad) + sl] = old[from[li][k]];
                                    }
                            }
                        }
            }
        }
    });
    timer.lap("pass B (planes, ranks, colouring)");
    if (singular.load()) {
        err = "singular (zero-volume) rest tetrahedron";
        return ERR_BAD_MESH;
    }
    return OK;
}

int read_veg(const char *path, std::vector<float> &rest, std::vector<int32_t> &tets, std::string &err)
{
    std::ifstream in(path);
    if (!in) {
        err = std::string("cannot open ") + (path ? path : "(null)");
        return ERR_IO;
    }
    rest.clear();
    tets.clear();
    std::vector<int64_t> ids;
    std::string line;
    int mode = 0, header = 0;  // 1 = vertices, 2 = elements
    int64_t min_id = std::numeric_limits<int64_t>::max();
    while (std::getline(in, line)) {
        size_t p = line.find_first_not_of(" \t\r\n");
        if (p == std::string::npos || line[p] == '#') continue;
        if (line[p] == '*') {
            std::string key = line.substr(p);
            for (auto &ch : key) ch = char(std::toupper(static_cast<unsigned char>(ch)));
            if (key.rfind("*VERTICES", 0) == 0) {
                mode = 1;
                header = 1;
            } else if (key.rfind("*ELEMENTS", 0) == 0) {
                mode = 2;
                header = 2;
            } else {
                mode = 0;
            }
            continue;
        }
        if (!mode) continue;
        if (header) {
            --header;
            if (mode == 2 && header == 1) {
                std::string ty = line.substr(p);
                while (!ty.empty() && std::isspace(static_cast<unsigned char>(ty.back()))) ty.pop_back();
                if (ty != "TET" && ty != "TETS") {
                    err = "only TET elements are supported, got '" + ty + "'";
                    return ERR_IO;
                }
            }
            continue;
        }
        for (auto &ch : line)
            if (ch == ',') ch = ' ';
        std::istringstream ss(line);
        if (mode == 1) {
            int64_t id;
            double x, y, z;
            if (!(ss >> id >> x >> y >> z)) {
                err = "malformed vertex line: " + line;
                return ERR_IO;
            }
            min_id = std::min(min_id, id);
            rest.push_back(float(x));
            rest.push_back(float(y));
            rest.push_back(float(z));
        } else {
            int64_t id, a, b, c, d;
            if (!(ss >> id >> a >> b >> c >> d)) {
                err = "malformed element line: " + line;
                return ERR_IO;
            }
            ids.insert(ids.end(), {a, b, c, d});
        }
    }
    if (rest.empty() || ids.empty()) {
        err = "no vertices or no tet elements found";
        return ERR_IO;
    }
    const int64_t base = min_id == std::numeric_limits<int64_t>::max() ? 1 : min_id;
    tets.resize(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) tets[i] = int32_t(ids[i] - base);
    return OK;
}

}  // namespace tsamd
