// C ABI of libtssplat_amd.so (declared in include/tssplat_amd.h).
//
// Owns what the reference's `struct TetSpheres` owns
// (/root/reference/tssplat_ext/tet_spheres/tet_spheres.h:9-42): the device
// copy of the rest-state operators and the per-evaluation scratch -- here the
// tiling plan's planes, the staging rows for shared vertices and the per-tile
// energy partials.  Unlike the reference destructor (tet_spheres.cpp:128-138)
// everything allocated is freed.
#include "../../include/tssplat_amd.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "capi_common.h"
#include "kernels.h"
#include "plan.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

}  // namespace

namespace tsamd {
int capi_fail(int code, const std::string &msg) { return fail(code, msg); }
}  // namespace tsamd

using tsamd::DeviceGuard;

struct tsamd_handle {
    tsamd::Plan plan;
    bool host_only = true;
    int device = -1;
    int64_t device_bytes = 0;
    // device
    tsamd::TileDesc *d_tiles = nullptr;
    uint8_t *d_blob = nullptr;
    int32_t *d_gvid = nullptr, *d_vdst = nullptr, *d_fin_vid = nullptr, *d_fin_off = nullptr;
    float *d_stage = nullptr;
    double *d_partials = nullptr;
    double *d_terms = nullptr;
    float *d_energy_scratch = nullptr;
    // optional kernel timing (bench.py roofline leg)
    bool timing = false;
    std::vector<hipEvent_t> events;  // 3 per recorded evaluation
};

struct tsamd_graph {
    tsamd::EvalGraph *g = nullptr;
    int device = -1;
};

namespace {

template <class T>
int upload(T *&dst, const void *src, size_t count, int64_t &bytes)
{
    const size_t nbytes = (count ? count : 1) * sizeof(T);
    TSAMD_HIP(hipMalloc(reinterpret_cast<void **>(&dst), nbytes));
    if (count && src) TSAMD_HIP(hipMemcpy(dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    bytes += int64_t(nbytes);
    return TSAMD_OK;
}

void release(tsamd_handle *h)
{
    if (!h) return;
    if (!h->host_only) {
        DeviceGuard g;
        (void)g.enter(h->device);
        (void)hipFree(h->d_tiles);
        (void)hipFree(h->d_blob);
        (void)hipFree(h->d_gvid);
        (void)hipFree(h->d_fin_vid);
        (void)hipFree(h->d_fin_off);
        (void)hipFree(h->d_vdst);
        (void)hipFree(h->d_stage);
        (void)hipFree(h->d_partials);
        (void)hipFree(h->d_terms);
        (void)hipFree(h->d_energy_scratch);
        for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    }
    delete h;
}

int to_device(tsamd_handle *h, int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(TSAMD_ERR_NO_DEVICE, "no HIP device is visible (this library has no CPU fallback)");
    if (device < 0) TSAMD_HIP(hipGetDevice(&device));
    if (device >= count) return fail(TSAMD_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    hipDeviceProp_t prop;
    TSAMD_HIP(hipGetDeviceProperties(&prop, device));
    if (size_t(h->plan.lds_bytes) > prop.sharedMemPerBlock && size_t(h->plan.lds_bytes) > prop.maxSharedMemoryPerMultiProcessor)
        return fail(TSAMD_ERR_NO_DEVICE, std::string("device ") + prop.name + " (" + prop.gcnArchName +
                                             ") offers less LDS per workgroup than the plan needs; built for gfx950");
    DeviceGuard g;
    TSAMD_HIP(g.enter(device));
    h->device = device;
    h->host_only = false;
    const tsamd::Plan &P = h->plan;
    int rc;
    if ((rc = upload(h->d_tiles, P.tiles.data(), P.tiles.size(), h->device_bytes))) return rc;
    if (tsamd::planes_paired(P.spt) && !P.tiles.empty()) {
        // the device image interleaves the planes that are loaded together (plan.h: planes_paired); row tables, rest positions and
        // the padding between tiles are copied as they are
        tsamd::RawVector<uint32_t> image(P.blob.size());
        const int64_t T = int64_t(P.tiles.size());
        const int nthr = int(std::max<int64_t>(1, std::min<int64_t>({int64_t(std::thread::hardware_concurrency()), 64, T / 64 + 1})));
        std::vector<std::thread> pool;
        for (int w = 0; w < nthr; ++w)
            pool.emplace_back([&, w]() {
                for (int64_t t = T * w / nthr; t < T * (w + 1) / nthr; ++t) {
                    const tsamd::TileDesc &d = P.tiles[size_t(t)];
                    const size_t b0 = size_t(d.blob_off / 4), b1 = t + 1 < T ? size_t(P.tiles[size_t(t) + 1].blob_off / 4) : P.blob.size();
                    const size_t np = size_t(P.n_planes) * size_t(d.s_pad);
                    tsamd::interleave_tile_planes(P.blob.data() + b0, image.data() + b0, P.n_planes, d.s_pad, P.spt);
                    std::memcpy(image.data() + b0 + np, P.blob.data() + b0 + np, (b1 - b0 - np) * 4);
                }
            });
        for (auto &th : pool) th.join();
        const size_t head = size_t(P.tiles[0].blob_off / 4);
        if (head) std::memcpy(image.data(), P.blob.data(), head * 4);
        if ((rc = upload(h->d_blob, reinterpret_cast<const uint8_t *>(image.data()), image.size() * 4, h->device_bytes))) return rc;
    } else if ((rc = upload(h->d_blob, reinterpret_cast<const uint8_t *>(P.blob.data()), P.blob.size() * 4, h->device_bytes))) {
        return rc;
    }
    if ((rc = upload(h->d_gvid, P.gvid.data(), P.gvid.size(), h->device_bytes))) return rc;
    if ((rc = upload(h->d_vdst, P.vdst.data(), P.vdst.size(), h->device_bytes))) return rc;
    if ((rc = upload(h->d_fin_vid, P.fin_vid.data(), P.fin_vid.size(), h->device_bytes))) return rc;
    if ((rc = upload(h->d_fin_off, P.fin_off.data(), P.fin_off.size(), h->device_bytes))) return rc;
    if ((rc = upload(h->d_stage, nullptr, size_t(P.n_stage) * 3 + 3 /* finish_kernel's look-ahead may read row 0 of an empty buffer */, h->device_bytes))) return rc;
    if ((rc = upload(h->d_partials, nullptr, P.tiles.size() * 2, h->device_bytes))) return rc;
    if ((rc = upload(h->d_terms, nullptr, 2, h->device_bytes))) return rc;
    if ((rc = upload(h->d_energy_scratch, nullptr, 1, h->device_bytes))) return rc;
    TSAMD_HIP(hipMemset(h->d_terms, 0, 2 * sizeof(double)));
    {
        const hipError_t ce = tsamd::configure_kernels(P.lds_bytes);
        if (ce == hipErrorInvalidDeviceFunction)
            return fail(TSAMD_ERR_HIP, "a tile kernel of this build has a static LDS object: its dynamic LDS array no longer starts at LDS "
                                       "address 0, which the kernels' absolute LDS addressing relies on (kernels.hip: lds_at)");
        TSAMD_HIP(ce);
    }
    return TSAMD_OK;
}

int create_common(const float *rest, int64_t n, const int32_t *tets, int64_t m, const tsamd_options *o,
                  tsamd_handle **out, const tsamd::ElementOperatorCSR *op = nullptr)
{
    if (!out) return fail(TSAMD_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    tsamd_options opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.device = -1;
    if (o) {
        if (o->struct_size != int32_t(sizeof(tsamd_options)))
            return fail(TSAMD_ERR_INVALID_ARGUMENT, "tsamd_options.struct_size does not match this library");
        if (o->abi_version != TSAMD_ABI_VERSION)
            return fail(TSAMD_ERR_INVALID_ARGUMENT, "tsamd_options.abi_version is " + std::to_string(o->abi_version) + ", this library implements ABI " +
                                                        std::to_string(TSAMD_ABI_VERSION) + " (set it to TSAMD_ABI_VERSION of the header you compile against)");
        opt = *o;
    }
    tsamd::PlanOptions po;
    if (opt.lds_budget_bytes > 0) po.lds_budget = opt.lds_budget_bytes;
    if (opt.max_threads > 0) po.max_threads = opt.max_threads;
    po.target_owned = opt.target_owned;
    po.num_threads = opt.num_threads;
    po.conflict_aware = (opt.debug_flags & 2) ? 0 : 1;  // bit 1 of the debug word switches the LDS-aware neighbour ordering off
    po.lane_search_sweeps = opt.lane_search_sweeps < 0 ? 0 : (opt.lane_search_sweeps == 0 ? 2 : opt.lane_search_sweeps);
    {
        const int spt = opt.slots_per_thread != 0 ? opt.slots_per_thread : tsamd::kSlotsPerLane;
        if (spt < 2 || spt > 4) return fail(TSAMD_ERR_INVALID_ARGUMENT, "slots_per_thread must be 0, 2, 3 or 4");
        if (!tsamd::lane_layout_supported(spt, opt.max_threads > 0 ? opt.max_threads : tsamd::kTileThreads))
            return fail(TSAMD_ERR_INVALID_ARGUMENT, "max_threads exceeds what the tile kernels are compiled for at " + std::to_string(spt) +
                                                        " slots per thread (2: 768, 3: 1024, 4: 768)");
        if (spt != tsamd::kSlotsPerLane && (op || opt.rebuild_dminv == 1))
            return fail(TSAMD_ERR_INVALID_ARGUMENT, "slots_per_thread 3 and 4 are built for the built-in operator with streamed Dm^-1 only");
        po.slots_per_lane = spt;
    }
    if (opt.rebuild_dminv < 0 || opt.rebuild_dminv > 2) return fail(TSAMD_ERR_INVALID_ARGUMENT, "rebuild_dminv must be 0 (auto), 1 (rebuild) or 2 (stream)");
    po.rebuild_dminv = opt.rebuild_dminv == 1 ? 1 : 0;
    tsamd_handle *h = new (std::nothrow) tsamd_handle();
    if (!h) return fail(TSAMD_ERR_INVALID_ARGUMENT, "out of host memory");
    std::string err;
    int rc = 0;
    try {
        rc = tsamd::build_plan(rest, n, tets, m, po, h->plan, err, op);
    } catch (const std::bad_alloc &) {
        rc = TSAMD_ERR_INVALID_ARGUMENT;
        err = "out of host memory while building the plan";
    }
    if (rc) {
        delete h;
        return fail(rc, err);
    }
    // Small batches: with at most two rounds of resident workgroups on the chip (512 tiles in flight on 256 CUs) a tile's latency,
    // not its halo, is what the step costs, and smaller tiles finish sooner: 3 k-tet spheres cut into four tiles of 768 owned tets
    // instead of three of 1 024 -- 64 / 128 / 256 x kuhn8 10.5 / 13.9 / 20.6 -> 9.8 / 12.5 / 19.7 us, 20 x kuhn8 10.2 -> 9.0.  Large
    // batches keep the fullest tiles that fit (fewest slots per tet).  Only with default tiling options.
    if (opt.max_threads == 0 && opt.target_owned == 0 && opt.lds_budget_bytes == 0 && h->plan.tiles.size() > 1 && h->plan.tiles.size() <= 1024) {
        tsamd::PlanOptions po2 = po;
        po2.target_owned = 768;
        tsamd::Plan alt;
        std::string err2;
        int rc2 = 1;
        try {
            rc2 = tsamd::build_plan(rest, n, tets, m, po2, alt, err2, op);
        } catch (const std::bad_alloc &) {
        }
        // ... where the smaller tiles do not cost more than 3 % of halo: a sphere of 3 k tets falls into quarters instead of thirds
        // (1.167 against 1.138 slots per tet), a 22 k- or 41 k-tet sphere would pay 3-5 % more slots for nothing (profiles/r06_experiments.md)
        // (48 x a.veg: 34.8 -> 42.4 us per step with the smaller tiles) -- unless the smaller tiles still fit the chip's ONE round of 512
        // workgroups, where they are simply more parallelism: 8 / 16 x a.veg 17.0 -> 16.5, 21.9 -> 21.1 us, 8 x kuhn19 20.0 -> 19.3,
        // 20 x delaunay3000 22.5 -> 21.3; 12 x kuhn19 (648 tiles) 22.3 -> 24.7: not taken
        if (rc2 == 0 && alt.tiles.size() > h->plan.tiles.size() && alt.tiles.size() <= 2048 &&
            (alt.tiles.size() <= 512 || double(alt.total_slots) <= 1.03 * double(h->plan.total_slots))) {
            h->plan = std::move(alt);
            po = po2;
        }
    }
    // Mid-size batches (rebuild_dminv = 0, "auto"): a plan of a few rounds of workgroups (more than the 512 the chip holds at once,
    // at most 2 048) neither hides a tile's stream behind thousands of others nor is it a single tile's latency -- there the 36 of 52
    // bytes per slot that rebuild_dminv does not stream win, PROVIDED the rest positions it keeps in LDS instead do not shrink the
    // tiles: 256 / 384 / 512 x kuhn8 (tiles = quarter spheres either way) 24.8 -> 22.8, 34.6 -> 32.6, 41.9 -> 40.5 us per step; 32 /
    // 48 x kuhn19 (1.300 against 1.284 slots per tet) 36.0 -> 37.6, 48.1 -> 50.3: stay streamed, like one-round batches (64 x kuhn8
    // 14.3 -> 14.7) and the 21 M-tet scene (+15 %) (profiles/r06_experiments.md).  Built-in operator, default lane layout and tiling
    // options only; 1 / 2 force either form.
    if (opt.rebuild_dminv == 0 && !op && po.slots_per_lane == tsamd::kSlotsPerLane && opt.max_threads == 0 && opt.target_owned == 0 &&
        opt.lds_budget_bytes == 0 && h->plan.tiles.size() > 512 && h->plan.tiles.size() <= 2048) {
        tsamd::PlanOptions po3 = po;
        po3.rebuild_dminv = 1;
        tsamd::Plan alt;
        std::string err3;
        int rc3 = 1;
        try {
            rc3 = tsamd::build_plan(rest, n, tets, m, po3, alt, err3, op);
        } catch (const std::bad_alloc &) {
        }
        if (rc3 == 0 && alt.tiles.size() <= 2048 && double(alt.total_slots) <= 1.002 * double(h->plan.total_slots)) h->plan = std::move(alt);
    }
    if (!opt.host_only) {
        rc = to_device(h, opt.device);
        if (rc) {
            std::string keep = g_err;
            release(h);
            g_err = keep;
            return rc;
        }
    }
    *out = h;
    return TSAMD_OK;
}

int check_eval(tsamd_handle *h, const float *x)
{
    if (!h) return fail(TSAMD_ERR_INVALID_ARGUMENT, "handle is null");
    if (h->host_only) return fail(TSAMD_ERR_HOST_ONLY, "handle was created host_only; no device path");
    if (!x && h->plan.n > 0) return fail(TSAMD_ERR_INVALID_ARGUMENT, "x_dev is null");
    return TSAMD_OK;
}

tsamd::EvalArgs eval_args(tsamd_handle *h, const float *x, const float *grad_out, float c1, float c2, int order, float *energy,
                          float *grad, const float *coef)
{
    tsamd::EvalArgs a;
    a.tiles = h->d_tiles;
    a.blob = h->d_blob;
    a.gvid = h->d_gvid;
    a.vdst = h->d_vdst;
    a.fin_vid = h->d_fin_vid;
    a.fin_off = h->d_fin_off;
    a.n_tiles = int64_t(h->plan.tiles.size());
    a.n_finish = int64_t(h->plan.fin_vid.size());
    a.block_threads = h->plan.block_threads;
    a.lds_bytes = h->plan.lds_bytes;
    a.vert_stride = h->plan.vert_stride;
    a.weighted = h->plan.n_planes == tsamd::kPlanesWeighted || h->plan.n_planes == tsamd::kPlanesWeightedSym;
    a.n_planes = h->plan.n_planes;
    a.rebuild = h->plan.n_planes == tsamd::kPlanesRebuild;
    a.spt = h->plan.spt;
    a.x = x;
    a.grad_out = grad_out;
    a.c1 = c1;
    a.c2 = c2;
    a.coef = coef;
    a.order = order;
    a.grad = grad;
    a.stage = h->d_stage;
    a.partials = h->d_partials;
    a.energy = energy;
    a.terms = h->d_terms;
    return a;
}

int evaluate(tsamd_handle *h, const float *x, const float *grad_out, float c1, float c2, int order, void *stream,
             float *energy, float *grad, const float *coef = nullptr)
{
    int rc = check_eval(h, x);
    if (rc) return rc;
    DeviceGuard g;
    TSAMD_HIP(g.enter(h->device));
    const tsamd::EvalArgs a = eval_args(h, x, grad_out, c1, c2, order, energy, grad, coef);
    if (h->timing) {
        hipEvent_t ev[3];
        for (auto &e : ev) TSAMD_HIP(hipEventCreate(&e));
        h->events.insert(h->events.end(), ev, ev + 3);
        TSAMD_HIP(tsamd::launch_eval(a, static_cast<hipStream_t>(stream), ev));
    } else {
        TSAMD_HIP(tsamd::launch_eval(a, static_cast<hipStream_t>(stream)));
    }
    return TSAMD_OK;
}

}  // namespace

extern "C" {

const char *tsamd_last_error(void) { return g_err.c_str(); }
const char *tsamd_version(void) { return "tssplat_amd 0.2 (gfx950)"; }
int32_t tsamd_abi_version(void) { return TSAMD_ABI_VERSION; }

int tsamd_create(const float *rest_xyz, int64_t n_vertices, const int32_t *tets, int64_t n_tets,
                 const tsamd_options *options, tsamd_handle **out)
{
    return create_common(rest_xyz, n_vertices, tets, n_tets, options, out);
}

int tsamd_create_with_operator(const float *rest_xyz, int64_t n_vertices, const int32_t *tets, int64_t n_tets,
                               const int64_t *op_rowptr, const int32_t *op_col, const double *op_val,
                               const tsamd_options *options, tsamd_handle **out)
{
    if (!op_rowptr) return fail(TSAMD_ERR_INVALID_ARGUMENT, "op_rowptr is null (use tsamd_create for the default operator)");
    const tsamd::ElementOperatorCSR op{op_rowptr, op_col, op_val};
    return create_common(rest_xyz, n_vertices, tets, n_tets, options, out, &op);
}

int tsamd_create_from_veg(const char *path, const tsamd_options *options, tsamd_handle **out)
{
    if (!path) return fail(TSAMD_ERR_INVALID_ARGUMENT, "path is null");
    std::vector<float> rest;
    std::vector<int32_t> tets;
    std::string err;
    int rc = tsamd::read_veg(path, rest, tets, err);
    if (rc) return fail(rc, err);
    return create_common(rest.data(), int64_t(rest.size() / 3), tets.data(), int64_t(tets.size() / 4), options, out);
}

void tsamd_destroy(tsamd_handle *h) { release(h); }

int64_t tsamd_num_vertices(const tsamd_handle *h) { return h ? h->plan.n : -1; }
int64_t tsamd_num_tets(const tsamd_handle *h) { return h ? h->plan.m : -1; }

int tsamd_get_plan_info(const tsamd_handle *h, tsamd_plan_info *out)
{
    if (!h || !out) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    const tsamd::Plan &P = h->plan;
    std::memset(out, 0, sizeof(*out));
    out->n_vertices = P.n;
    out->n_tets = P.m;
    out->n_tiles = int64_t(P.tiles.size());
    out->n_components = P.n_components;
    out->total_slots = P.total_slots;
    out->total_tile_vertices = P.total_tile_verts;
    out->shared_vertex_copies = P.n_stage;
    out->finish_vertices = int64_t(P.fin_vid.size());
    out->device_bytes = h->device_bytes;
    out->max_slots = P.max_slots;
    out->max_tile_vertices = P.max_verts;
    out->block_threads = P.block_threads;
    out->lds_bytes = P.lds_bytes;
    out->slots_per_thread = P.spt;
    out->n_planes = P.n_planes;
    return TSAMD_OK;
}

int tsamd_get_tile(const tsamd_handle *h, int64_t tile, tsamd_tile_view *out)
{
    if (!h || !out) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    const tsamd::Plan &P = h->plan;
    if (tile < 0 || tile >= int64_t(P.tiles.size())) return fail(TSAMD_ERR_INVALID_ARGUMENT, "tile out of range");
    const tsamd::TileDesc &d = P.tiles[size_t(tile)];
    out->n_slots = d.n_slots;
    out->n_owned = d.n_owned;
    out->s_pad = d.s_pad;
    out->n_verts = d.n_verts;
    out->n_excl = d.n_excl;
    out->stage_off = d.stage_off;
    out->n_rows = d.n_rows;
    out->rec_base = d.rec_base;
    out->planes = P.blob.data() + d.blob_off / 4;
    out->row_start = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(out->planes) + tsamd::tile_rowtab_offset(P.n_planes, d.s_pad));
    out->gvid = P.gvid.data() + d.vert_off;
    out->vdst = P.vdst.data() + d.vert_off;
    out->slot_tet = P.slot_tet.data() + P.slot_base[size_t(tile)];
    out->rest = P.n_planes == tsamd::kPlanesRebuild
                    ? reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(out->planes) + tsamd::tile_rest_offset(P.n_planes, d.s_pad))
                    : nullptr;
    return TSAMD_OK;
}

int tsamd_get_finish_lists(const tsamd_handle *h, int64_t *n_finish, const int32_t **vid, const int32_t **off,
                           const int32_t **idx)
{
    if (!h || !n_finish || !vid || !off || !idx) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    *n_finish = int64_t(h->plan.fin_vid.size());
    *vid = h->plan.fin_vid.data();
    *off = h->plan.fin_off.data();
    *idx = h->plan.fin_idx.data();
    return TSAMD_OK;
}

int tsamd_get_adjacency(const tsamd_handle *h, const int32_t **nbr)
{
    if (!h || !nbr) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    *nbr = h->plan.nbr.data();
    return TSAMD_OK;
}

int tsamd_forward(tsamd_handle *h, const float *x_dev, float c1, float c2, int order, void *stream,
                  float *energy_dev)
{
    if (!energy_dev) return fail(TSAMD_ERR_INVALID_ARGUMENT, "energy_dev is null");
    return evaluate(h, x_dev, nullptr, c1, c2, order, stream, energy_dev, nullptr);
}

int tsamd_backward(tsamd_handle *h, const float *x_dev, const float *grad_out_dev, float c1, float c2, int order,
                   void *stream, float *grad_dev)
{
    if (!grad_dev && h && h->plan.n > 0) return fail(TSAMD_ERR_INVALID_ARGUMENT, "grad_dev is null");
    return evaluate(h, x_dev, grad_out_dev, c1, c2, order, stream, nullptr, grad_dev);
}

int tsamd_forward_backward(tsamd_handle *h, const float *x_dev, const float *grad_out_dev, float c1, float c2,
                           int order, void *stream, float *energy_dev, float *grad_dev)
{
    if (!grad_dev && h && h->plan.n > 0) return fail(TSAMD_ERR_INVALID_ARGUMENT, "grad_dev is null");
    return evaluate(h, x_dev, grad_out_dev, c1, c2, order, stream, energy_dev ? energy_dev : (h ? h->d_energy_scratch : nullptr),
                    grad_dev);
}

int tsamd_evaluate_dev_coef(tsamd_handle *h, const float *x_dev, const float *grad_out_dev, const float *coef_dev,
                            int order, void *stream, float *energy_dev, float *grad_dev)
{
    if (!coef_dev) return fail(TSAMD_ERR_INVALID_ARGUMENT, "coef_dev is null");
    if (!energy_dev && !grad_dev) return fail(TSAMD_ERR_INVALID_ARGUMENT, "nothing to compute: energy_dev and grad_dev are both null");
    return evaluate(h, x_dev, grad_out_dev, 0.f, 0.f, order, stream,
                    energy_dev ? energy_dev : (grad_dev && h ? h->d_energy_scratch : nullptr), grad_dev, coef_dev);
}

int tsamd_graph_create(tsamd_handle *h, const float *x_dev, const float *grad_out_dev, int order, float *energy_dev,
                       float *grad_dev, tsamd_graph **out)
{
    if (!out) return fail(TSAMD_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int rc = check_eval(h, x_dev);
    if (rc) return rc;
    if (!energy_dev && !grad_dev) return fail(TSAMD_ERR_INVALID_ARGUMENT, "nothing to compute: energy_dev and grad_dev are both null");
    DeviceGuard g;
    TSAMD_HIP(g.enter(h->device));
    const tsamd::EvalArgs a = eval_args(h, x_dev, grad_out_dev, 0.f, 0.f, order,
                                        energy_dev ? energy_dev : h->d_energy_scratch, grad_dev, nullptr);
    tsamd::EvalGraph *eg = nullptr;
    TSAMD_HIP(tsamd::eval_graph_create(a, &eg));
    tsamd_graph *w = new (std::nothrow) tsamd_graph();
    if (!w) {
        tsamd::eval_graph_destroy(eg);
        return fail(TSAMD_ERR_INVALID_ARGUMENT, "out of host memory");
    }
    w->g = eg;
    w->device = h->device;
    *out = w;
    return TSAMD_OK;
}

int tsamd_graph_launch(tsamd_graph *graph, float c1, float c2, void *stream)
{
    if (!graph || !graph->g) return fail(TSAMD_ERR_INVALID_ARGUMENT, "graph is null");
    DeviceGuard g;
    TSAMD_HIP(g.enter(graph->device));
    TSAMD_HIP(tsamd::eval_graph_launch(graph->g, c1, c2, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_graph_launch_to(tsamd_graph *graph, float c1, float c2, void *stream, float *energy_copy_dev, float *grad_dev)
{
    if (!graph || !graph->g) return fail(TSAMD_ERR_INVALID_ARGUMENT, "graph is null");
    DeviceGuard g;
    TSAMD_HIP(g.enter(graph->device));
    const hipError_t e = tsamd::eval_graph_launch(graph->g, c1, c2, static_cast<hipStream_t>(stream), energy_copy_dev, grad_dev);
    if (e == hipErrorInvalidValue && (energy_copy_dev || grad_dev))
        return fail(TSAMD_ERR_INVALID_ARGUMENT, "this graph cannot redirect that output: it was created without energy_dev / without grad_dev "
                                                "(no node of it computes the value)");
    TSAMD_HIP(e);
    return TSAMD_OK;
}

void tsamd_graph_destroy(tsamd_graph *graph)
{
    if (!graph) return;
    DeviceGuard g;
    (void)g.enter(graph->device);
    tsamd::eval_graph_destroy(graph->g);
    delete graph;
}

struct tsamd_train_loop {
    tsamd::TrainLoopGraph *g = nullptr;
    int device = 0;
    int n_iters = 0;
    std::vector<tsamd::TrainLoopStep> steps;
};

int64_t tsamd_train_loop_workspace_bytes(int32_t n_iters) { return n_iters < 1 ? -1 : int64_t(n_iters) * 8; }

int tsamd_train_loop_create(tsamd_handle *h, float *param_dev, float *grad_dev, float *g1_dev, float *g2_dev, float *energy_ring_dev,
                            void *workspace_dev, int32_t n_iters, tsamd_train_loop **out)
{
    if (!out) return fail(TSAMD_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int rc = check_eval(h, param_dev);
    if (rc) return rc;
    if (!grad_dev || !g1_dev || !g2_dev || !energy_ring_dev || !workspace_dev) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null device pointer");
    if (n_iters < 1 || n_iters > 4096) return fail(TSAMD_ERR_INVALID_ARGUMENT, "n_iters out of range (1 .. 4096)");
    DeviceGuard g;
    TSAMD_HIP(g.enter(h->device));
    const tsamd::EvalArgs a = eval_args(h, param_dev, nullptr, 0.f, 0.f, 2, energy_ring_dev, grad_dev, nullptr);
    tsamd::TrainLoopGraph *tg = nullptr;
    TSAMD_HIP(tsamd::train_loop_create(a, param_dev, g1_dev, g2_dev, 3 * h->plan.n, workspace_dev, n_iters, &tg));
    tsamd_train_loop *w = new (std::nothrow) tsamd_train_loop();
    if (!w) {
        tsamd::train_loop_destroy(tg);
        return fail(TSAMD_ERR_INVALID_ARGUMENT, "out of host memory");
    }
    w->g = tg;
    w->device = h->device;
    w->n_iters = n_iters;
    w->steps.resize(size_t(n_iters));
    *out = w;
    return TSAMD_OK;
}

int tsamd_train_loop_launch(tsamd_train_loop *loop, const float *c1, const float *c2, const int32_t *order, float lr, float beta1, float beta2,
                            int64_t first_step, const float *grad_limit, void *stream)
{
    if (!loop || !loop->g) return fail(TSAMD_ERR_INVALID_ARGUMENT, "loop is null");
    if (!c1 || !c2 || !order || first_step < 1) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null schedule or first_step < 1");
    for (int k = 0; k < loop->n_iters; ++k) {
        tsamd::TrainLoopStep &s = loop->steps[size_t(k)];
        s.c1 = c1[k];
        s.c2 = c2[k];
        s.order = order[k];
        s.bias1 = float(1.0 - std::pow(double(beta1), double(first_step + k)));   // as tsamd_adam_uniform_step (optimizer.py:67-68)
        s.bias2 = float(1.0 - std::pow(double(beta2), double(first_step + k)));
        s.limit = grad_limit ? grad_limit[k] : -1.f;
    }
    DeviceGuard g;
    TSAMD_HIP(g.enter(loop->device));
    TSAMD_HIP(tsamd::train_loop_launch(loop->g, loop->steps.data(), lr, beta1, beta2, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

void tsamd_train_loop_destroy(tsamd_train_loop *loop)
{
    if (!loop) return;
    DeviceGuard g;
    (void)g.enter(loop->device);
    tsamd::train_loop_destroy(loop->g);
    delete loop;
}

int tsamd_read_energy_terms(tsamd_handle *h, void *stream, double *terms_host2)
{
    if (!h || !terms_host2) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    if (h->host_only) return fail(TSAMD_ERR_HOST_ONLY, "handle was created host_only; no device path");
    DeviceGuard g;
    TSAMD_HIP(g.enter(h->device));
    TSAMD_HIP(hipMemcpyAsync(terms_host2, h->d_terms, 2 * sizeof(double), hipMemcpyDeviceToHost,
                             static_cast<hipStream_t>(stream)));
    TSAMD_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_set_timing(tsamd_handle *h, int enable)
{
    if (!h) return fail(TSAMD_ERR_INVALID_ARGUMENT, "handle is null");
    if (h->host_only) return fail(TSAMD_ERR_HOST_ONLY, "handle was created host_only; no device path");
    h->timing = enable != 0;
    return TSAMD_OK;
}

int tsamd_get_timing(tsamd_handle *h, double *tile_kernel_ms, double *finish_kernel_ms, int64_t *evaluations)
{
    if (!h || !tile_kernel_ms || !finish_kernel_ms || !evaluations) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard g;
    TSAMD_HIP(g.enter(h->device));
    double a = 0.0, b = 0.0;
    const size_t n = h->events.size() / 3;
    for (size_t i = 0; i < n; ++i) {
        hipEvent_t *ev = &h->events[3 * i];
        TSAMD_HIP(hipEventSynchronize(ev[2]));
        float m0 = 0.f, m1 = 0.f;
        TSAMD_HIP(hipEventElapsedTime(&m0, ev[0], ev[1]));
        TSAMD_HIP(hipEventElapsedTime(&m1, ev[1], ev[2]));
        a += m0;
        b += m1;
    }
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    h->events.clear();
    *tile_kernel_ms = a;
    *finish_kernel_ms = b;
    *evaluations = int64_t(n);
    return TSAMD_OK;
}

int tsamd_scale(const float *in_dev, const float *scalar_dev, float *out_dev, int64_t n, void *stream)
{
    if (n < 0 || (n > 0 && (!in_dev || !scalar_dev || !out_dev))) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    TSAMD_HIP(tsamd::launch_scale(in_dev, scalar_dev, out_dev, n, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_adam_uniform_step(float *param_dev, const float *grad_dev, float *g1_dev, float *g2_dev, int64_t n, float lr,
                            float beta1, float beta2, int64_t step, float grad_limit, void *workspace_dev, void *stream)
{
    if (n < 0 || step < 1 || (n > 0 && (!param_dev || !grad_dev || !g1_dev || !g2_dev || !workspace_dev)))
        return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument, negative size or step < 1");
    const float bias1 = float(1.0 - std::pow(double(beta1), double(step)));   // optimizer.py:67-68
    const float bias2 = float(1.0 - std::pow(double(beta2), double(step)));
    TSAMD_HIP(tsamd::launch_adam_uniform(param_dev, grad_dev, g1_dev, g2_dev, n, lr, beta1, beta2, bias1, bias2, grad_limit,
                                         workspace_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int64_t tsamd_grad_limit_workspace_bytes(void) { return 16; }

int tsamd_grad_limit(float *grad_dev, int64_t n, float s_threshold, float s, void *workspace_dev, void *stream)
{
    if (n < 0 || (n > 0 && (!grad_dev || !workspace_dev))) return fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    TSAMD_HIP(tsamd::launch_grad_limit(grad_dev, n, s_threshold, s, workspace_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

}  // extern "C"
