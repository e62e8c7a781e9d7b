// C ABI of the streaming-tile path (include/tssplat_amd.h, "streaming tiles (experimental)"): its own handle type next to
// tsamd_handle -- same mathematics, another plan and another tile kernel (stream_plan.h, stream_kernels.hip).
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "capi_common.h"
#include "stream_kernels.h"
#include "stream_plan.h"

using tsamd::capi_fail;
using tsamd::DeviceGuard;

struct tsamd_stream {
    tsamd::StreamPlan plan;
    int device = -1;
    bool host_only = true;
    int64_t device_bytes = 0;
    tsamd::StreamTubeDesc *d_tubes = nullptr;
    uint8_t *d_blob = nullptr;
    int32_t *d_fin_vid = nullptr, *d_fin_off = nullptr;
    float *d_stage = nullptr, *d_energy_scratch = nullptr;
    double *d_partials = nullptr, *d_terms = nullptr;
    bool timing = false;
    std::vector<hipEvent_t> events;
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

void release(tsamd_stream *h)
{
    if (!h) return;
    if (!h->host_only) {
        DeviceGuard g;
        (void)g.enter(h->device);
        (void)hipFree(h->d_tubes);
        (void)hipFree(h->d_blob);
        (void)hipFree(h->d_fin_vid);
        (void)hipFree(h->d_fin_off);
        (void)hipFree(h->d_stage);
        (void)hipFree(h->d_energy_scratch);
        (void)hipFree(h->d_partials);
        (void)hipFree(h->d_terms);
        for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    }
    delete h;
}

int to_device(tsamd_stream *h, int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return capi_fail(TSAMD_ERR_NO_DEVICE, "no HIP device is visible (this library has no CPU fallback)");
    if (device < 0) TSAMD_HIP(hipGetDevice(&device));
    if (device >= count) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    const tsamd::StreamPlan &P = h->plan;
    if (tsamd::stream_lds_bytes(P.max_vslots, P.max_bands) > 160 * 1024)
        return capi_fail(TSAMD_ERR_TILING, "a tube needs more than 160 KiB of LDS (vertex ring + band descriptors): use tsamd_create");
    DeviceGuard g;
    TSAMD_HIP(g.enter(device));
    h->device = device;
    h->host_only = false;
    int rc;
    if ((rc = upload(h->d_tubes, P.tubes.data(), P.tubes.size(), h->device_bytes))) return rc;
    if ((rc = upload(h->d_blob, reinterpret_cast<const uint8_t *>(P.blob.data()), P.blob.size() * 4, h->device_bytes))) return rc;
    if ((rc = upload(h->d_fin_vid, P.fin_vid.data(), P.fin_vid.size(), h->device_bytes))) return rc;
    if ((rc = upload(h->d_fin_off, P.fin_off.data(), P.fin_off.size(), h->device_bytes))) return rc;
    if ((rc = upload(h->d_stage, nullptr, size_t(P.n_stage) * 3, h->device_bytes))) return rc;
    if ((rc = upload(h->d_partials, nullptr, P.tubes.size() * 2, h->device_bytes))) return rc;
    if ((rc = upload(h->d_terms, nullptr, 2, h->device_bytes))) return rc;
    if ((rc = upload(h->d_energy_scratch, nullptr, 1, h->device_bytes))) return rc;
    TSAMD_HIP(hipMemset(h->d_terms, 0, 2 * sizeof(double)));
    TSAMD_HIP(tsamd::configure_stream_kernels(tsamd::stream_lds_bytes(P.max_vslots, P.max_bands)));
    return TSAMD_OK;
}

}  // namespace

extern "C" {

int tsamd_stream_create(const float *rest_xyz, int64_t n_vertices, const int32_t *tets, int64_t n_tets, int32_t device,
                        int32_t host_only, int32_t num_threads, tsamd_stream **out)
{
    if (!out) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    tsamd_stream *h = new (std::nothrow) tsamd_stream();
    if (!h) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "out of host memory");
    std::string err;
    int rc = 0;
    try {
        rc = tsamd::build_stream_plan(rest_xyz, n_vertices, tets, n_tets, num_threads, h->plan, err);
    } catch (const std::bad_alloc &) {
        rc = TSAMD_ERR_INVALID_ARGUMENT;
        err = "out of host memory while building the streaming plan";
    }
    if (rc) {
        delete h;
        return capi_fail(rc, err);
    }
    if (!host_only) {
        rc = to_device(h, device);
        if (rc) {
            release(h);
            return rc;
        }
    }
    *out = h;
    return TSAMD_OK;
}

void tsamd_stream_destroy(tsamd_stream *h) { release(h); }

int tsamd_stream_info(const tsamd_stream *h, tsamd_stream_plan_info *out)
{
    if (!h || !out) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    const tsamd::StreamPlan &P = h->plan;
    std::memset(out, 0, sizeof(*out));
    out->n_vertices = P.n;
    out->n_tets = P.m;
    out->n_components = P.n_components;
    out->n_tubes = int64_t(P.tubes.size());
    out->total_slots = P.total_slots;
    out->total_bands = P.total_bands;
    out->total_pairs = P.total_pairs;
    out->total_chunks = P.total_chunks;
    out->shared_vertex_copies = P.n_stage;
    out->finish_vertices = int64_t(P.fin_vid.size());
    out->device_bytes = h->device_bytes;
    out->blob_bytes = int64_t(P.blob.size()) * 4;
    out->max_vertex_slots = P.max_vslots;
    out->max_bands = P.max_bands;
    out->band_slots = tsamd::kBand;
    out->lds_bytes = tsamd::stream_lds_bytes(P.max_vslots, P.max_bands);
    return TSAMD_OK;
}

int tsamd_stream_get_tube(const tsamd_stream *h, int64_t tube, tsamd_stream_tube_view *out)
{
    if (!h || !out) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    const tsamd::StreamPlan &P = h->plan;
    if (tube < 0 || tube >= int64_t(P.tubes.size())) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "tube out of range");
    const tsamd::StreamTubeDesc &d = P.tubes[size_t(tube)];
    out->n_bands = d.n_bands;
    out->n_vslots = d.n_vslots;
    out->n_owned = d.n_owned;
    out->n_slots = d.n_slots;
    out->blob = reinterpret_cast<const uint8_t *>(P.blob.data()) + d.blob_off;
    out->blob_bytes = int64_t((tube + 1 < int64_t(P.tubes.size()) ? P.tubes[size_t(tube) + 1].blob_off : uint64_t(P.blob.size()) * 4) - d.blob_off);
    out->slot_tet = P.slot_tet.data() + size_t(P.tube_band_base[size_t(tube)]) * tsamd::kBand;
    return TSAMD_OK;
}

int tsamd_stream_get_finish_lists(const tsamd_stream *h, int64_t *n_finish, int64_t *n_stage, const int32_t **vid, const int32_t **off)
{
    if (!h || !n_finish || !n_stage || !vid || !off) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
    *n_finish = int64_t(h->plan.fin_vid.size());
    *n_stage = h->plan.n_stage;
    *vid = h->plan.fin_vid.data();
    *off = h->plan.fin_off.data();
    return TSAMD_OK;
}

int tsamd_stream_forward_backward(tsamd_stream *h, const float *x_dev, const float *grad_out_dev, float c1, float c2, int order,
                                  void *stream, float *energy_dev, float *grad_dev)
{
    if (!h) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "handle is null");
    if (h->host_only) return capi_fail(TSAMD_ERR_HOST_ONLY, "handle was created host_only; no device path");
    if ((!x_dev || !grad_dev) && h->plan.n > 0) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "x_dev / grad_dev is null");
    DeviceGuard g;
    TSAMD_HIP(g.enter(h->device));
    tsamd::StreamEvalArgs a;
    a.tubes = h->d_tubes;
    a.blob = h->d_blob;
    a.fin_vid = h->d_fin_vid;
    a.fin_off = h->d_fin_off;
    a.n_tubes = int64_t(h->plan.tubes.size());
    a.n_finish = int64_t(h->plan.fin_vid.size());
    a.lds_bytes = tsamd::stream_lds_bytes(h->plan.max_vslots, h->plan.max_bands);
    a.x = x_dev;
    a.grad_out = grad_out_dev;
    a.c1 = c1;
    a.c2 = c2;
    a.order = order;
    a.grad = grad_dev;
    a.stage = h->d_stage;
    a.partials = h->d_partials;
    a.energy = energy_dev ? energy_dev : h->d_energy_scratch;
    a.terms = h->d_terms;
    if (h->timing) {
        hipEvent_t ev[3];
        for (auto &e : ev) TSAMD_HIP(hipEventCreate(&e));
        h->events.insert(h->events.end(), ev, ev + 3);
        TSAMD_HIP(tsamd::launch_stream_eval(a, static_cast<hipStream_t>(stream), ev));
    } else {
        TSAMD_HIP(tsamd::launch_stream_eval(a, static_cast<hipStream_t>(stream)));
    }
    return TSAMD_OK;
}

int tsamd_stream_set_timing(tsamd_stream *h, int enable)
{
    if (!h || h->host_only) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null or host-only handle");
    h->timing = enable != 0;
    return TSAMD_OK;
}

int tsamd_stream_get_timing(tsamd_stream *h, double *tube_kernel_ms, double *finish_kernel_ms, int64_t *evaluations)
{
    if (!h || !tube_kernel_ms || !finish_kernel_ms || !evaluations) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null argument");
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
    *tube_kernel_ms = a;
    *finish_kernel_ms = b;
    *evaluations = int64_t(n);
    return TSAMD_OK;
}

int tsamd_stream_read_energy_terms(tsamd_stream *h, void *stream, double *terms_host2)
{
    if (!h || !terms_host2 || h->host_only) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null or host-only handle");
    DeviceGuard g;
    TSAMD_HIP(g.enter(h->device));
    TSAMD_HIP(hipMemcpyAsync(terms_host2, h->d_terms, 2 * sizeof(double), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    TSAMD_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

}  // extern "C"
