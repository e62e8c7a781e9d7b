// Host-side construction of the streaming-tile plan (stream_plan.h): components -> tubes -> breadth-first levels ->
// bands -> per-band planes, vertex ring slots, entering-vertex lists and per-(vertex, band) incidence chunks.
// Pure C++17, no HIP.  Replaces, like plan.cpp, the role of libpgo in the reference's constructor
// (/root/reference/tssplat_ext/tet_spheres/tet_spheres.cpp:140-159).
#include "stream_plan.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <functional>
#include <queue>
#include <thread>

namespace tsamd {
namespace {

enum { OK = 0, ERR_INVALID = 1, ERR_BAD_MESH = 2, ERR_TILING = 6 };

struct TubeBuild {
    std::vector<int32_t> members;   // owned tets first, then the side halo
    int32_t n_owned = 0;
    std::vector<int32_t> level;     // per member
    int32_t n_levels = 0;
};

template <class Fn>
void parallel_for(int64_t n, int nthreads, Fn fn)
{
    nthreads = int(std::max<int64_t>(1, std::min<int64_t>(nthreads, n)));
    if (nthreads == 1) {
        for (int64_t i = 0; i < n; ++i) fn(i, 0);
        return;
    }
    std::atomic<int64_t> next{0};
    auto body = [&](int w) {
        for (;;) {
            const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) break;
            fn(i, w);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(body, t);
    body(0);
    for (auto &th : pool) th.join();
}

// per-worker scratch over the whole mesh with O(1) reset through stamps
struct Scratch {
    std::vector<int32_t> stamp, local;
    int32_t cur = 0;
    void init(int64_t m)
    {
        if (int64_t(stamp.size()) != m) {
            stamp.assign(size_t(m), 0);
            local.assign(size_t(m), 0);
            cur = 0;
        }
    }
    int32_t next()
    {
        if (cur > (1 << 30)) {
            std::fill(stamp.begin(), stamp.end(), 0);
            cur = 0;
        }
        return ++cur;
    }
};

// Levels of one tube part: multi-source breadth-first search over owned + halo tets from the low end of the sweep axis.
// Returns false when a level is wider than a band.
bool level_tube_variant(const int32_t *nbr, const std::vector<float> &cen, int sweep, TubeBuild &T, Scratch &S, int variant);

// The widest level depends on where the search starts: try a few end caps (either end of the sweep axis, 2 % / 1 % / 4 %
// of the tube) before giving up on this cut.
bool level_tube(const int32_t *nbr, const std::vector<float> &cen, int sweep, TubeBuild &T, Scratch &S)
{
    for (int variant = 0; variant < 6; ++variant)
        if (level_tube_variant(nbr, cen, sweep, T, S, variant)) return true;
    return false;
}

bool level_tube_variant(const int32_t *nbr, const std::vector<float> &cen, int sweep, TubeBuild &T, Scratch &S, int variant)
{
    const float dir = (variant & 1) ? -1.f : 1.f;
    const int frac = (variant >> 1) == 0 ? 50 : ((variant >> 1) == 1 ? 100 : 25);
    const int32_t st = S.next();
    const int32_t k = int32_t(T.members.size());
    for (int32_t i = 0; i < k; ++i) {
        S.stamp[size_t(T.members[size_t(i)])] = st;
        S.local[size_t(T.members[size_t(i)])] = i;
    }
    T.level.assign(size_t(k), -1);
    // seeds: the tets with the smallest sweep coordinate (an end cap, so the first levels are not a point's tiny shells)
    std::vector<int32_t> order(static_cast<size_t>(k));
    for (int32_t i = 0; i < k; ++i) order[size_t(i)] = i;
    const int32_t n_seed = std::max<int32_t>(1, std::min<int32_t>(kBand - 16, k / frac));
    std::partial_sort(order.begin(), order.begin() + n_seed, order.end(), [&](int32_t a, int32_t b) {
        const float ca = dir * cen[3 * size_t(T.members[size_t(a)]) + sweep], cb = dir * cen[3 * size_t(T.members[size_t(b)]) + sweep];
        return ca != cb ? ca < cb : T.members[size_t(a)] < T.members[size_t(b)];
    });
    std::vector<int32_t> frontier(order.begin(), order.begin() + n_seed), next;
    int32_t lvl = 0, reached = 0;
    size_t scan = 0;   // next candidate for a restart (pieces the search has not reached)
    std::vector<int32_t> by_coord;
    for (;;) {
        for (int32_t i : frontier) T.level[size_t(i)] = lvl;
        while (!frontier.empty()) {
            if (int32_t(frontier.size()) > kBand) return false;
            reached += int32_t(frontier.size());
            next.clear();
            for (int32_t i : frontier) {
                const int32_t e = T.members[size_t(i)];
                for (int f = 0; f < 4; ++f) {
                    const int32_t q = nbr[4 * size_t(e) + f];
                    if (q < 0 || S.stamp[size_t(q)] != st) continue;
                    const int32_t j = S.local[size_t(q)];
                    if (T.level[size_t(j)] >= 0) continue;
                    T.level[size_t(j)] = lvl + 1;
                    next.push_back(j);
                }
            }
            std::sort(next.begin(), next.end());
            frontier.swap(next);
            ++lvl;
        }
        if (reached == k) break;
        // a piece the search has not reached (no face path inside the tube): continue with it on fresh levels
        if (by_coord.empty()) {
            by_coord = order;
            std::sort(by_coord.begin(), by_coord.end(), [&](int32_t a, int32_t b) {
                const float ca = dir * cen[3 * size_t(T.members[size_t(a)]) + sweep], cb = dir * cen[3 * size_t(T.members[size_t(b)]) + sweep];
                return ca != cb ? ca < cb : T.members[size_t(a)] < T.members[size_t(b)];
            });
        }
        while (scan < by_coord.size() && T.level[size_t(by_coord[scan])] >= 0) ++scan;
        frontier.assign(1, by_coord[scan]);
    }
    T.n_levels = lvl;
    return true;
}

// owned part + its side halo
void make_members(const int32_t *nbr, const int32_t *own, int64_t cnt, TubeBuild &T, Scratch &S)
{
    const int32_t st = S.next();
    T.members.assign(own, own + cnt);
    T.n_owned = int32_t(cnt);
    for (int64_t i = 0; i < cnt; ++i) S.stamp[size_t(own[i])] = st;
    for (int64_t i = 0; i < cnt; ++i)
        for (int f = 0; f < 4; ++f) {
            const int32_t q = nbr[4 * size_t(own[i]) + f];
            if (q < 0 || S.stamp[size_t(q)] == st) continue;
            S.stamp[size_t(q)] = st;
            T.members.push_back(q);
        }
    std::sort(T.members.begin() + cnt, T.members.end());
}

struct TubeBlob {
    std::vector<uint32_t> data;
    StreamTubeDesc desc{};
    std::vector<int32_t> slot_tet;    // n_bands * kBand
    int64_t pairs = 0, chunks = 0;
};

}  // namespace

int build_stream_plan(const float *rest, int64_t n, const int32_t *tets, int64_t m, int num_threads, StreamPlan &P, std::string &err)
{
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
    int nthreads = num_threads > 0 ? num_threads : int(std::thread::hardware_concurrency());
    nthreads = std::max(1, std::min(nthreads, 64));
    P = StreamPlan();
    P.n = n;
    P.m = m;
    int rc = build_adjacency(tets, n, m, P.nbr, nthreads, err);
    if (rc) return rc;
    const int32_t *nbr = P.nbr.data();

    // ---- components (flood fill; a component's tets in increasing order) ----
    std::vector<int32_t> comp(static_cast<size_t>(m), -1), comp_tets(static_cast<size_t>(m));
    std::vector<int64_t> comp_start;
    {
        int64_t filled = 0;
        std::vector<int32_t> stack;
        for (int64_t s = 0; s < m; ++s) {
            if (comp[size_t(s)] >= 0) continue;
            const int32_t c = int32_t(comp_start.size());
            comp_start.push_back(filled);
            comp[size_t(s)] = c;
            stack.push_back(int32_t(s));
            const int64_t first = filled;
            while (!stack.empty()) {
                const int32_t e = stack.back();
                stack.pop_back();
                comp_tets[size_t(filled++)] = e;
                for (int f = 0; f < 4; ++f) {
                    const int32_t q = nbr[4 * size_t(e) + f];
                    if (q >= 0 && comp[size_t(q)] < 0) {
                        comp[size_t(q)] = c;
                        stack.push_back(q);
                    }
                }
            }
            std::sort(comp_tets.begin() + first, comp_tets.begin() + filled);
        }
        comp_start.push_back(filled);
    }
    const int64_t C = int64_t(comp_start.size()) - 1;
    P.n_components = C;

    std::vector<float> cen(static_cast<size_t>(3 * m));
    for (int64_t i = 0; i < m; ++i)
        for (int d = 0; d < 3; ++d) {
            float s = 0.f;
            for (int a = 0; a < 4; ++a) s += rest[3 * size_t(tets[4 * i + a]) + d];
            cen[3 * size_t(i) + d] = 0.25f * s;
        }

    // ---- tubes of every component: the fewest K x K parts whose widest level fits a band ----
    std::vector<std::vector<TubeBuild>> comp_tubes(static_cast<size_t>(C));
    std::vector<Scratch> scratch(static_cast<size_t>(nthreads));
    std::atomic<int> failed{0};
    parallel_for(C, nthreads, [&](int64_t c, int w) {
        Scratch &S = scratch[size_t(w)];
        S.init(m);
        int32_t *ids = comp_tets.data() + comp_start[size_t(c)];
        const int64_t cnt = comp_start[size_t(c) + 1] - comp_start[size_t(c)];
        float lo[3], hi[3];
        for (int d = 0; d < 3; ++d) lo[d] = 3.4e38f, hi[d] = -3.4e38f;
        for (int64_t i = 0; i < cnt; ++i)
            for (int d = 0; d < 3; ++d) {
                lo[d] = std::min(lo[d], cen[3 * size_t(ids[i]) + d]);
                hi[d] = std::max(hi[d], cen[3 * size_t(ids[i]) + d]);
            }
        int sweep = 0;
        for (int d = 1; d < 3; ++d)
            if (hi[d] - lo[d] > hi[sweep] - lo[sweep]) sweep = d;
        const int ax0 = (sweep + 1) % 3, ax1 = (sweep + 2) % 3;
        std::vector<int32_t> work(ids, ids + cnt);
        for (int K = 1; K <= 24; ++K) {
            if (int64_t(K) * K > cnt) break;
            // K strips along ax0 (equal counts), each cut into K along ax1
            auto by = [&](int ax) {
                return [&, ax](int32_t a, int32_t b) {
                    const float ca = cen[3 * size_t(a) + ax], cb = cen[3 * size_t(b) + ax];
                    return ca != cb ? ca < cb : a < b;
                };
            };
            std::sort(work.begin(), work.end(), by(ax0));
            std::vector<TubeBuild> tubes;
            bool ok = true;
            for (int i = 0; i < K && ok; ++i) {
                const int64_t a0 = cnt * i / K, a1 = cnt * (i + 1) / K;
                std::sort(work.begin() + a0, work.begin() + a1, by(ax1));
                for (int j = 0; j < K && ok; ++j) {
                    const int64_t b0 = a0 + (a1 - a0) * j / K, b1 = a0 + (a1 - a0) * (j + 1) / K;
                    if (b1 <= b0) continue;
                    std::vector<int32_t> own(work.begin() + b0, work.begin() + b1);
                    std::sort(own.begin(), own.end());
                    TubeBuild T;
                    make_members(nbr, own.data(), int64_t(own.size()), T, S);
                    ok = level_tube(nbr, cen, sweep, T, S);
                    if (ok) tubes.push_back(std::move(T));
                }
            }
            if (ok) {
                comp_tubes[size_t(c)] = std::move(tubes);
                return;
            }
        }
        failed.store(1);
    });
    if (failed.load()) {
        err = "a component cannot be cut into tubes whose widest breadth-first level fits a band of " + std::to_string(kBand) + " slots";
        return ERR_TILING;
    }
    std::vector<TubeBuild> tubes;
    for (auto &ct : comp_tubes)
        for (auto &t : ct) tubes.push_back(std::move(t));
    comp_tubes.clear();
    const int64_t NT = int64_t(tubes.size());

    // ---- vertices touched by more than one tube go through staging rows (vertex-major, tube order) ----
    std::vector<int32_t> vcount(static_cast<size_t>(n), 0);
    {
        std::vector<int32_t> last_tube(static_cast<size_t>(n), -1);
        for (int64_t t = 0; t < NT; ++t)
            for (int32_t e : tubes[size_t(t)].members)
                for (int a = 0; a < 4; ++a) {
                    const int32_t v = tets[4 * size_t(e) + a];
                    if (last_tube[size_t(v)] != int32_t(t)) {
                        last_tube[size_t(v)] = int32_t(t);
                        ++vcount[size_t(v)];
                    }
                }
    }
    std::vector<int32_t> fin_of(static_cast<size_t>(n), -1), fin_cur;
    {
        int64_t rows = 0;
        for (int64_t v = 0; v < n; ++v) {
            if (vcount[size_t(v)] <= 1) continue;
            fin_of[size_t(v)] = int32_t(P.fin_vid.size());
            P.fin_vid.push_back(int32_t(v));
            P.fin_off.push_back(int32_t(rows));
            rows += vcount[size_t(v)];
            if (rows >= (int64_t(1) << 31)) {
                err = "too many shared vertex copies for 32-bit offsets";
                return ERR_TILING;
            }
        }
        P.fin_off.push_back(int32_t(rows));
        P.n_stage = rows;
        fin_cur.assign(P.fin_off.begin(), P.fin_off.end() - 1);
    }
    // staging row of (tube, shared vertex): assigned in tube order (serial: it defines the summation order)
    std::vector<std::vector<std::pair<int32_t, int32_t>>> tube_rows(static_cast<size_t>(NT));
    {
        std::vector<int32_t> last_tube(static_cast<size_t>(n), -1);
        for (int64_t t = 0; t < NT; ++t) {
            auto &rows = tube_rows[size_t(t)];
            for (int32_t e : tubes[size_t(t)].members)
                for (int a = 0; a < 4; ++a) {
                    const int32_t v = tets[4 * size_t(e) + a];
                    if (last_tube[size_t(v)] == int32_t(t)) continue;
                    last_tube[size_t(v)] = int32_t(t);
                    if (fin_of[size_t(v)] >= 0) rows.push_back({v, fin_cur[size_t(fin_of[size_t(v)])]++});
                }
            std::sort(rows.begin(), rows.end());
        }
    }

    // ---- per tube: bands, slots, vertex ring, planes and lists ----
    std::vector<TubeBlob> blobs(static_cast<size_t>(NT));
    std::atomic<int> singular{0}, too_many_vertices{0};
    parallel_for(NT, nthreads, [&](int64_t t, int w) {
        Scratch &S = scratch[size_t(w)];
        S.init(m);
        const TubeBuild &T = tubes[size_t(t)];
        TubeBlob &B = blobs[size_t(t)];
        const int32_t k = int32_t(T.members.size());
        // bands: consecutive levels merged while they fit
        std::vector<int32_t> width(static_cast<size_t>(T.n_levels), 0);
        for (int32_t i = 0; i < k; ++i) ++width[size_t(T.level[size_t(i)])];
        std::vector<int32_t> band_of_level(static_cast<size_t>(T.n_levels), 0);
        int32_t nb = 0;
        {
            int32_t cur = 0;
            for (int32_t l = 0; l < T.n_levels; ++l) {
                if (cur + width[size_t(l)] > kBand && cur > 0) {
                    ++nb;
                    cur = 0;
                }
                band_of_level[size_t(l)] = nb;
                cur += width[size_t(l)];
            }
            ++nb;
        }
        // slot of every member: per band owned first, then halo; inside each class by level, then by tet id
        std::vector<int32_t> order(static_cast<size_t>(k));
        for (int32_t i = 0; i < k; ++i) order[size_t(i)] = i;
        auto band_of = [&](int32_t i) { return band_of_level[size_t(T.level[size_t(i)])]; };
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            const int32_t ba = band_of(a), bb = band_of(b);
            if (ba != bb) return ba < bb;
            const bool oa = a < T.n_owned, ob = b < T.n_owned;
            if (oa != ob) return oa;
            if (T.level[size_t(a)] != T.level[size_t(b)]) return T.level[size_t(a)] < T.level[size_t(b)];
            return T.members[size_t(a)] < T.members[size_t(b)];
        });
        std::vector<int32_t> m_band(static_cast<size_t>(k)), m_lane(static_cast<size_t>(k));
        std::vector<int32_t> band_slots(static_cast<size_t>(nb), 0), band_owned(static_cast<size_t>(nb), 0);
        for (int32_t i : order) {
            const int32_t b = band_of(i);
            m_band[size_t(i)] = b;
            m_lane[size_t(i)] = band_slots[size_t(b)]++;
            if (i < T.n_owned) ++band_owned[size_t(b)];
        }
        const int32_t st = S.next();
        for (int32_t i = 0; i < k; ++i) {
            S.stamp[size_t(T.members[size_t(i)])] = st;
            S.local[size_t(T.members[size_t(i)])] = i;
        }
        // vertices: first / last band, ring slots
        std::vector<int32_t> verts;
        verts.reserve(size_t(k));
        for (int32_t i = 0; i < k; ++i)
            for (int a = 0; a < 4; ++a) verts.push_back(tets[4 * size_t(T.members[size_t(i)]) + a]);
        std::sort(verts.begin(), verts.end());
        verts.erase(std::unique(verts.begin(), verts.end()), verts.end());
        const int32_t nv = int32_t(verts.size());
        auto vidx = [&](int32_t v) { return int32_t(std::lower_bound(verts.begin(), verts.end(), v) - verts.begin()); };
        std::vector<int32_t> vfirst(static_cast<size_t>(nv), 1 << 30), vlast(static_cast<size_t>(nv), -1);
        for (int32_t i = 0; i < k; ++i)
            for (int a = 0; a < 4; ++a) {
                const int32_t j = vidx(tets[4 * size_t(T.members[size_t(i)]) + a]);
                vfirst[size_t(j)] = std::min(vfirst[size_t(j)], m_band[size_t(i)]);
                vlast[size_t(j)] = std::max(vlast[size_t(j)], m_band[size_t(i)]);
            }
        std::vector<int32_t> vorder(static_cast<size_t>(nv)), vslot(static_cast<size_t>(nv), -1);
        for (int32_t j = 0; j < nv; ++j) vorder[size_t(j)] = j;
        std::sort(vorder.begin(), vorder.end(), [&](int32_t a, int32_t b) {
            return vfirst[size_t(a)] != vfirst[size_t(b)] ? vfirst[size_t(a)] < vfirst[size_t(b)] : verts[size_t(a)] < verts[size_t(b)];
        });
        int32_t n_vslots = 0;
        {
            typedef std::pair<int32_t, int32_t> Rel;   // (band from which the slot is free, slot)
            std::priority_queue<Rel, std::vector<Rel>, std::greater<Rel>> busy;
            std::priority_queue<int32_t, std::vector<int32_t>, std::greater<int32_t>> free_slots;
            for (int32_t j : vorder) {
                while (!busy.empty() && busy.top().first <= vfirst[size_t(j)]) {
                    free_slots.push(busy.top().second);
                    busy.pop();
                }
                int32_t s;
                if (!free_slots.empty()) {
                    s = free_slots.top();
                    free_slots.pop();
                } else {
                    s = n_vslots++;
                }
                vslot[size_t(j)] = s;
                busy.push({vlast[size_t(j)] + kVertexReuseGap, s});
            }
        }
        if (n_vslots > kMaxVertexSlots) {
            too_many_vertices.store(1);
            return;
        }
        // per band: entering vertices, pairs with their incidence entries
        std::vector<std::vector<int32_t>> enter(static_cast<size_t>(nb));
        for (int32_t j : vorder) enter[size_t(vfirst[size_t(j)])].push_back(j);
        std::vector<std::vector<std::pair<int32_t, uint16_t>>> inc(static_cast<size_t>(nb));   // (vertex index, entry)
        for (int32_t i : order)
            for (int a = 0; a < 4; ++a)
                inc[size_t(m_band[size_t(i)])].push_back({vidx(tets[4 * size_t(T.members[size_t(i)]) + a]), uint16_t((m_lane[size_t(i)] << 2) | a)});
        // sizes
        std::vector<StreamBandDesc> bd(static_cast<size_t>(nb));
        size_t off = size_t(nb) * sizeof(StreamBandDesc);
        off = (off + 15) & ~size_t(15);
        std::vector<std::vector<StreamPair>> pairs(static_cast<size_t>(nb));
        std::vector<std::vector<uint16_t>> chunks(static_cast<size_t>(nb));
        const auto &rows = tube_rows[size_t(t)];
        for (int32_t b = 0; b < nb; ++b) {
            auto &iv = inc[size_t(b)];
            std::stable_sort(iv.begin(), iv.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
            auto &pp = pairs[size_t(b)];
            auto &ch = chunks[size_t(b)];
            for (size_t p0 = 0; p0 < iv.size();) {
                size_t p1 = p0;
                while (p1 < iv.size() && iv[p1].first == iv[p0].first) ++p1;
                const int32_t j = iv[p0].first;
                StreamPair sp;
                sp.vslot = uint16_t(vslot[size_t(j)]);
                const uint32_t nch = uint32_t((p1 - p0 + 3) / 4);
                sp.n_chunks = uint16_t(nch);
                sp.first_chunk = uint32_t(ch.size() / 4);
                sp.out_row = 0;
                if (vlast[size_t(j)] == b) {
                    sp.n_chunks |= kPairLast;
                    const int32_t v = verts[size_t(j)];
                    if (fin_of[size_t(v)] >= 0) {
                        sp.n_chunks |= kPairShared;
                        auto it = std::lower_bound(rows.begin(), rows.end(), std::make_pair(v, int32_t(-1)));
                        sp.out_row = it->second;
                    } else {
                        sp.out_row = v;
                    }
                }
                for (size_t q = p0; q < p1; ++q) ch.push_back(iv[q].second);
                while (ch.size() % 4) ch.push_back(uint16_t(kBand << 2));
                pp.push_back(sp);
                p0 = p1;
            }
            StreamBandDesc &d = bd[size_t(b)];
            d.n_slots = uint16_t(band_slots[size_t(b)]);
            d.n_owned = uint16_t(band_owned[size_t(b)]);
            d.n_enter = uint16_t(enter[size_t(b)].size());
            d.n_pairs = uint16_t(pp.size());
            d.planes_off = uint32_t(off);
            off += size_t(kPlanes) * kBand * 4;
            d.enter_off = uint32_t(off);
            off += enter[size_t(b)].size() * 8;
            d.pairs_off = uint32_t(off);
            off += pp.size() * sizeof(StreamPair);
            off = (off + 7) & ~size_t(7);
            d.chunks_off = uint32_t(off);
            off += ch.size() * 2;
            off = (off + 15) & ~size_t(15);
            B.pairs += int64_t(pp.size());
            B.chunks += int64_t(ch.size() / 4);
        }
        off = (off + 127) & ~size_t(127);
        B.data.assign(off / 4, 0u);
        uint8_t *base = reinterpret_cast<uint8_t *>(B.data.data());
        std::memcpy(base, bd.data(), size_t(nb) * sizeof(StreamBandDesc));
        B.slot_tet.assign(size_t(nb) * kBand, -1);
        for (int32_t b = 0; b < nb; ++b) {
            const StreamBandDesc &d = bd[size_t(b)];
            uint32_t *pl = reinterpret_cast<uint32_t *>(base + d.planes_off);
            // inert padding: vertices = ring slot 0, neighbours = the slot itself, Dm^-1 = 0
            for (int32_t i = 0; i < kBand; ++i) {
                const uint32_t self = stream_token(0, uint32_t(i));
                pl[2 * kBand + i] = self | (self << 16);
                pl[3 * kBand + i] = self | (self << 16);
            }
            int32_t *en = reinterpret_cast<int32_t *>(base + d.enter_off);
            for (size_t q = 0; q < enter[size_t(b)].size(); ++q) {
                const int32_t j = enter[size_t(b)][q];
                en[2 * q] = vslot[size_t(j)];
                en[2 * q + 1] = verts[size_t(j)];
            }
            if (!pairs[size_t(b)].empty()) std::memcpy(base + d.pairs_off, pairs[size_t(b)].data(), pairs[size_t(b)].size() * sizeof(StreamPair));
            if (!chunks[size_t(b)].empty()) std::memcpy(base + d.chunks_off, chunks[size_t(b)].data(), chunks[size_t(b)].size() * 2);
        }
        for (int32_t i = 0; i < k; ++i) {
            const int32_t e = T.members[size_t(i)], b = m_band[size_t(i)], lane = m_lane[size_t(i)];
            const bool owned = i < T.n_owned;
            uint32_t *pl = reinterpret_cast<uint32_t *>(base + bd[size_t(b)].planes_off);
            B.slot_tet[size_t(b) * kBand + size_t(lane)] = e;
            uint32_t lv[4], tk[4];
            for (int a = 0; a < 4; ++a) lv[a] = uint32_t(vslot[size_t(vidx(tets[4 * size_t(e) + a]))]) << 4;
            if (owned) lv[0] |= kOwnedBit;
            const uint32_t self = stream_token(0, uint32_t(lane));
            for (int f = 0; f < 4; ++f) {
                const int32_t q = nbr[4 * size_t(e) + f];
                uint32_t v = self;
                if (q >= 0 && S.stamp[size_t(q)] == st) {
                    const int32_t j = S.local[size_t(q)];
                    // owned slots see all their neighbours (members by construction); halo slots only their owned ones
                    if (owned || j < T.n_owned) v = stream_token(m_band[size_t(j)] - b, uint32_t(m_lane[size_t(j)]));
                }
                tk[f] = v;
            }
            pl[0 * kBand + lane] = lv[0] | (lv[1] << 16);
            pl[1 * kBand + lane] = lv[2] | (lv[3] << 16);
            pl[2 * kBand + lane] = tk[0] | (owned ? kOwnedBit : 0u) | (tk[1] << 16);
            pl[3 * kBand + lane] = tk[2] | (tk[3] << 16);
            const int32_t *tt = tets + 4 * size_t(e);
            double D[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) D[3 * r + c] = double(rest[3 * size_t(tt[c + 1]) + r]) - double(rest[3 * size_t(tt[0]) + r]);
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
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    const float v = float(Cf[3 * c + r] / det);   // inverse = cofactor^T / det, double -> fp32 (tet_spheres.cpp:43-45)
                    std::memcpy(&pl[size_t(4 + 3 * r + c) * kBand + lane], &v, 4);
                }
        }
        B.desc.n_bands = nb;
        B.desc.n_vslots = n_vslots;
        B.desc.n_owned = T.n_owned;
        B.desc.n_slots = k;
    });
    if (singular.load()) {
        err = "singular (zero-volume) rest tetrahedron";
        return ERR_BAD_MESH;
    }
    if (too_many_vertices.load()) {
        err = "a tube keeps more than " + std::to_string(kMaxVertexSlots) + " vertices alive at once";
        return ERR_TILING;
    }
    // ---- concatenate ----
    int64_t total = 0, bands = 0;
    P.tubes.resize(size_t(NT));
    P.tube_band_base.resize(size_t(NT) + 1);
    for (int64_t t = 0; t < NT; ++t) {
        blobs[size_t(t)].desc.blob_off = uint64_t(total);
        P.tubes[size_t(t)] = blobs[size_t(t)].desc;
        P.tube_band_base[size_t(t)] = bands;
        total += int64_t(blobs[size_t(t)].data.size()) * 4;
        bands += blobs[size_t(t)].desc.n_bands;
        P.total_slots += blobs[size_t(t)].desc.n_slots;
        P.total_pairs += blobs[size_t(t)].pairs;
        P.total_chunks += blobs[size_t(t)].chunks;
        P.max_vslots = std::max(P.max_vslots, blobs[size_t(t)].desc.n_vslots);
        P.max_bands = std::max(P.max_bands, blobs[size_t(t)].desc.n_bands);
    }
    P.tube_band_base[size_t(NT)] = bands;
    P.total_bands = bands;
    P.blob.resize(size_t(total / 4));
    P.slot_tet.resize(size_t(bands) * kBand);
    parallel_for(NT, nthreads, [&](int64_t t, int) {
        const TubeBlob &B = blobs[size_t(t)];
        std::memcpy(P.blob.data() + B.desc.blob_off / 4, B.data.data(), B.data.size() * 4);
        std::memcpy(P.slot_tet.data() + size_t(P.tube_band_base[size_t(t)]) * kBand, B.slot_tet.data(), B.slot_tet.size() * 4);
    });
    return OK;
}

}  // namespace tsamd
