// C ABI of the surface glue (include/tssplat_amd.h, "surface" section): the handle owns the device copy
// of the surface topology (ids, triangles, vertex -> face lists); positions and gradients stay with the caller.
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "capi_common.h"
#include "surface.h"

using tsamd::capi_fail;
using tsamd::DeviceGuard;

struct tsamd_surface {
    int device = -1;
    int64_t nv = 0, nf = 0, n_tet_vertices = 0;
    bool unique_vid = true;
    int32_t *d_vid = nullptr, *d_faces = nullptr, *d_off = nullptr, *d_ent = nullptr;
};

namespace {

template <class T>
int upload(T *&dst, const std::vector<T> &src)
{
    dst = nullptr;
    if (src.empty()) return TSAMD_OK;
    TSAMD_HIP(hipMalloc(reinterpret_cast<void **>(&dst), src.size() * sizeof(T)));
    TSAMD_HIP(hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return TSAMD_OK;
}

tsamd::SurfaceArgs args_of(const tsamd_surface *s)
{
    tsamd::SurfaceArgs a;
    a.surface_vid = s->d_vid;
    a.faces = s->d_faces;
    a.vf_off = s->d_off;
    a.vf_ent = s->d_ent;
    a.nv = s->nv;
    a.nf = s->nf;
    a.n_tet_vertices = s->n_tet_vertices;
    a.unique_vid = s->unique_vid;
    return a;
}

int check(const tsamd_surface *s, const void *a, const void *b, DeviceGuard &g)
{
    if (!s) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null surface handle");
    if (s->nv > 0 && (!a || !b)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null device pointer");
    TSAMD_HIP(g.enter(s->device));
    return TSAMD_OK;
}

}  // namespace

extern "C" {

int tsamd_extract_surface(const int32_t *tets, int64_t n_tets, int64_t n_vertices, int32_t *surface_vid,
                          int64_t *n_surface_vertices, int32_t *faces, int64_t *n_faces)
{
    if (n_tets < 0 || n_vertices < 0 || (n_tets > 0 && !tets) || !n_surface_vertices || !n_faces)
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "tsamd_extract_surface: null pointer or negative size");
    std::vector<int32_t> vid, f;
    std::string err;
    int rc = tsamd::extract_surface(tets, n_tets, n_vertices, vid, f, err);
    if (rc) return capi_fail(rc, err);
    const int64_t cap_v = *n_surface_vertices, cap_f = *n_faces;
    *n_surface_vertices = int64_t(vid.size());
    *n_faces = int64_t(f.size() / 3);
    if (!surface_vid && !faces) return TSAMD_OK;  // size query
    if (!surface_vid || !faces || cap_v < int64_t(vid.size()) || cap_f < int64_t(f.size() / 3))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "tsamd_extract_surface: output buffers too small (query the sizes first)");
    std::copy(vid.begin(), vid.end(), surface_vid);
    std::copy(f.begin(), f.end(), faces);
    return TSAMD_OK;
}

int tsamd_surface_create(const int32_t *surface_vid, int64_t n_surface_vertices, const int32_t *faces, int64_t n_faces,
                         int64_t n_tet_vertices, int device, tsamd_surface **out)
{
    if (!out) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null output handle");
    *out = nullptr;
    if (n_surface_vertices < 0 || n_faces < 0 || n_tet_vertices < 0 || (n_surface_vertices > 0 && !surface_vid) ||
        (n_faces > 0 && !faces))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "tsamd_surface_create: null pointer or negative size");
    if (n_tet_vertices >= (int64_t(1) << 31) / 3 || n_faces >= (int64_t(1) << 29))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "tsamd_surface_create: mesh too large for 32-bit offsets");
    for (int64_t i = 0; i < n_surface_vertices; ++i)
        if (surface_vid[i] < 0 || surface_vid[i] >= n_tet_vertices)
            return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "tsamd_surface_create: surface vertex id out of range");
    for (int64_t i = 0; i < 3 * n_faces; ++i)
        if (faces[i] < 0 || faces[i] >= n_surface_vertices)
            return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "tsamd_surface_create: triangle vertex index out of range");
    tsamd_surface *s = new (std::nothrow) tsamd_surface;
    if (!s) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "out of host memory");
    s->nv = n_surface_vertices;
    s->nf = n_faces;
    s->n_tet_vertices = n_tet_vertices;
    {
        std::vector<int32_t> sorted(surface_vid, surface_vid + n_surface_vertices);
        std::sort(sorted.begin(), sorted.end());
        s->unique_vid = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        delete s;
        return capi_fail(TSAMD_ERR_NO_DEVICE, "no HIP device is visible");
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    s->device = device;
    std::vector<int32_t> off, ent;
    tsamd::build_vertex_faces(faces, n_faces, n_surface_vertices, off, ent);
    int rc;
    {
        DeviceGuard g;
        hipError_t e = g.enter(device);
        if (e != hipSuccess) {
            delete s;
            return capi_fail(TSAMD_ERR_HIP, std::string("hipSetDevice failed: ") + hipGetErrorString(e));
        }
        rc = upload(s->d_vid, std::vector<int32_t>(surface_vid, surface_vid + n_surface_vertices));
        if (!rc) rc = upload(s->d_faces, std::vector<int32_t>(faces, faces + 3 * n_faces));
        if (!rc) rc = upload(s->d_off, off);
        if (!rc) rc = upload(s->d_ent, ent);
    }
    if (rc) {
        tsamd_surface_destroy(s);
        return rc;
    }
    *out = s;
    return TSAMD_OK;
}

void tsamd_surface_destroy(tsamd_surface *s)
{
    if (!s) return;
    DeviceGuard g;
    if (s->device >= 0 && g.enter(s->device) == hipSuccess) {
        (void)hipFree(s->d_vid);
        (void)hipFree(s->d_faces);
        (void)hipFree(s->d_off);
        (void)hipFree(s->d_ent);
    }
    delete s;
}

int tsamd_surface_positions(const tsamd_surface *s, const float *tet_v_dev, void *stream, float *v_pos_dev)
{
    DeviceGuard g;
    if (int rc = check(s, tet_v_dev, v_pos_dev, g)) return rc;
    TSAMD_HIP(tsamd::launch_surface_positions(args_of(s), tet_v_dev, v_pos_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_surface_positions_backward(const tsamd_surface *s, const float *grad_v_pos_dev, void *stream, float *grad_tet_v_dev)
{
    DeviceGuard g;
    if (int rc = check(s, grad_v_pos_dev, grad_tet_v_dev, g)) return rc;
    if (s->n_tet_vertices > 0 && !grad_tet_v_dev) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null device pointer");
    TSAMD_HIP(tsamd::launch_surface_positions_backward(args_of(s), grad_v_pos_dev, grad_tet_v_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_vertex_normals(const tsamd_surface *s, const float *v_pos_dev, void *stream, float *v_nrm_dev, float *raw_dev)
{
    DeviceGuard g;
    if (int rc = check(s, v_pos_dev, v_nrm_dev, g)) return rc;
    TSAMD_HIP(tsamd::launch_vertex_normals(args_of(s), v_pos_dev, v_nrm_dev, raw_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_vertex_normals_backward(const tsamd_surface *s, const float *v_pos_dev, const float *raw_dev, const float *grad_nrm_dev,
                                  void *workspace_dev, void *stream, float *grad_v_pos_dev)
{
    DeviceGuard g;
    if (int rc = check(s, v_pos_dev, grad_v_pos_dev, g)) return rc;
    if (s->nv > 0 && (!raw_dev || !grad_nrm_dev || !workspace_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null device pointer");
    TSAMD_HIP(tsamd::launch_vertex_normals_backward(args_of(s), v_pos_dev, raw_dev, grad_nrm_dev, static_cast<float *>(workspace_dev),
                                                    grad_v_pos_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

}  // extern "C"
