// C ABI of the renderer slice (include/tssplat_amd.h, "renderer" section): stateless entry points, the caller owns
// every buffer (positions, triangles, the depth-key workspace, outputs) and names the device by making it current.
#include <climits>
#include <string>

#include "capi_common.h"
#include "raster.h"

using tsamd::capi_fail;

namespace {

int check_image(int64_t batch, int32_t height, int32_t width)
{
    // 8192: snapped window coordinates are kept within +-2^22 sub-pixel units (1/256 pixel) = +-16384 pixels and a triangle with a
    // vertex beyond that is dropped (there is no clipping), so the cap leaves a guard band of at least one screen on every side
    if (batch < 0 || height < 0 || width < 0 || height > 8192 || width > 8192)
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "batch / height / width out of range (0 .. 8192 pixels per side)");
    return TSAMD_OK;
}

}  // namespace

extern "C" {

int64_t tsamd_rasterize_workspace_bytes(int64_t batch, int64_t n_vertices, int32_t height, int32_t width)
{
    if (batch < 0 || n_vertices < 0 || height < 0 || width < 0) return -1;
    // depth keys (padded to 16 B) + snapped vertices + one flag per view (has a vertex at w <= 0: near-plane clipping needed)
    return ((batch * int64_t(height) * int64_t(width) + 1) & ~int64_t(1)) * 8 + batch * n_vertices * 16 + ((batch * 4 + 15) & ~int64_t(15));
}

int64_t tsamd_pair_masks_bytes(int64_t batch, int32_t height, int32_t width)
{
    if (check_image(batch, height, width) != TSAMD_OK) return -1;
    return tsamd::pair_masks_bytes(batch, height, width);
}

int tsamd_rasterize(const float *pos_clip_dev, int64_t batch, int64_t n_vertices, const int32_t *tri_dev, int64_t n_triangles, int32_t height,
                    int32_t width, void *workspace_dev, float *rast_out_dev, void *pair_masks_out_dev, void *stream)
{
    int rc = check_image(batch, height, width);
    if (rc) return rc;
    if (n_vertices < 0 || n_triangles < 0 || n_triangles > (int64_t(1) << 24) - 1)
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "negative size or more than 2^24 - 1 triangles (the id + 1 is returned as a float32, exact up to 2^24)");
    if ((batch + 7) / 8 * 8 * ((n_triangles + 255) / 256) > int64_t(INT32_MAX))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "batch x triangles / 256 exceeds the grid limit (2^31 - 1 workgroups)");
    const int64_t pixels = batch * int64_t(height) * width;
    if (pixels > 0 && (!workspace_dev || !rast_out_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "workspace_dev / rast_out_dev is null");
    if (batch * n_triangles > 0 && (!pos_clip_dev || !tri_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "pos_clip_dev / tri_dev is null");
    TSAMD_HIP(tsamd::launch_rasterize(pos_clip_dev, batch, n_vertices, tri_dev, n_triangles, height, width, workspace_dev, rast_out_dev,
                                      pair_masks_out_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_interpolate(const float *attr_dev, int64_t attr_batch, int64_t n_vertices, int32_t n_channels, const float *rast_dev,
                      const int32_t *tri_dev, int64_t n_triangles, int64_t batch, int32_t height, int32_t width, float *out_dev, void *stream)
{
    int rc = check_image(batch, height, width);
    if (rc) return rc;
    if (n_vertices < 0 || n_triangles < 0 || n_channels < 1 || (attr_batch != 1 && attr_batch != batch))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "attr_batch must be 1 or batch, n_channels >= 1, sizes >= 0");
    const int64_t pixels = batch * int64_t(height) * width;
    if (pixels > 0 && (!rast_dev || !out_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "rast_dev / out_dev is null");
    if (pixels > 0 && n_triangles > 0 && (!attr_dev || !tri_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "attr_dev / tri_dev is null");
    TSAMD_HIP(tsamd::launch_interpolate(attr_dev, attr_batch, n_vertices, n_channels, rast_dev, tri_dev, n_triangles, batch, height, width, out_dev,
                                        static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_interpolate_backward(const float *attr_dev, int64_t attr_batch, int64_t n_vertices, int32_t n_channels, const float *rast_dev,
                               const int32_t *tri_dev, int64_t n_triangles, int64_t batch, int32_t height, int32_t width, const float *grad_out_dev,
                               float *grad_attr_dev, float *grad_rast_dev, void *stream)
{
    int rc = check_image(batch, height, width);
    if (rc) return rc;
    if (n_vertices < 0 || n_triangles < 0 || n_channels < 1 || (attr_batch != 1 && attr_batch != batch))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "attr_batch must be 1 or batch, n_channels >= 1, sizes >= 0");
    const int64_t pixels = batch * int64_t(height) * width;
    if (attr_batch * n_vertices > 0 && !grad_attr_dev) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "grad_attr_dev is null");
    if (pixels > 0 && (!rast_dev || !grad_out_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "rast_dev / grad_out_dev is null");
    if (pixels > 0 && n_triangles > 0 && (!attr_dev || !tri_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "attr_dev / tri_dev is null");
    TSAMD_HIP(tsamd::launch_interpolate_backward(attr_dev, attr_batch, n_vertices, n_channels, rast_dev, tri_dev, n_triangles, batch, height, width,
                                                 grad_out_dev, grad_attr_dev, grad_rast_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_rasterize_backward(const float *pos_clip_dev, int64_t batch, int64_t n_vertices, const int32_t *tri_dev, int64_t n_triangles,
                             int32_t height, int32_t width, const float *rast_dev, const float *grad_rast_dev, float *grad_pos_dev, void *stream)
{
    int rc = check_image(batch, height, width);
    if (rc) return rc;
    if (n_vertices < 0 || n_triangles < 0) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "negative size");
    const int64_t pixels = batch * int64_t(height) * width;
    if (batch * n_vertices > 0 && !grad_pos_dev) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "grad_pos_dev is null");
    if (pixels > 0 && n_triangles > 0 && (!pos_clip_dev || !tri_dev || !rast_dev || !grad_rast_dev))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null device pointer");
    TSAMD_HIP(tsamd::launch_rasterize_backward(pos_clip_dev, batch, n_vertices, tri_dev, n_triangles, height, width, rast_dev, grad_rast_dev,
                                               grad_pos_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int64_t tsamd_antialias_topology_workspace_bytes(int64_t n_triangles)
{
    if (n_triangles < 0 || n_triangles >= (int64_t(1) << 30)) return -1;   // (triangle, edge) ids are 32-bit
    return tsamd::antialias_topology_workspace_bytes(n_triangles);
}

int tsamd_antialias_topology(const int32_t *tri_dev, int64_t n_triangles, void *workspace_dev, int32_t *edge_partner_dev, void *stream)
{
    if (n_triangles < 0 || n_triangles >= (int64_t(1) << 30)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "n_triangles out of range (0 .. 2^30 - 1)");
    if (n_triangles > 0 && (!tri_dev || !workspace_dev || !edge_partner_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "null device pointer");
    TSAMD_HIP(tsamd::launch_antialias_topology(tri_dev, n_triangles, workspace_dev, edge_partner_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

namespace {
int check_antialias(int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width, int32_t n_channels)
{
    int rc = check_image(batch, height, width);
    if (rc) return rc;
    if (n_vertices < 0 || n_triangles < 0 || n_channels < 1) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "negative size or n_channels < 1");
    return TSAMD_OK;
}
}  // namespace

int64_t tsamd_antialias_prepared_bytes(int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width)
{
    if (check_image(batch, height, width) != TSAMD_OK || n_vertices < 0 || n_triangles < 0) return -1;
    return tsamd::antialias_prepared_bytes(batch, n_vertices, n_triangles, height, width);
}

int tsamd_antialias_prepare(const float *rast_dev, const float *pos_clip_dev, const int32_t *tri_dev, const int32_t *edge_partner_dev,
                            const void *pair_masks_dev, int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width,
                            void *prepared_dev, void *stream)
{
    int rc = check_antialias(batch, n_vertices, n_triangles, height, width, 1);
    if (rc) return rc;
    const int64_t pixels = batch * int64_t(height) * width;
    if ((pixels > 0 || batch * n_vertices > 0) && !prepared_dev) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "prepared_dev is null");
    if ((pixels > 0 && !rast_dev && !pair_masks_dev) || (batch * n_vertices > 0 && !pos_clip_dev))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "rast_dev (without pair_masks_dev) / pos_clip_dev is null");
    if (batch * n_triangles > 0 && (!tri_dev || !edge_partner_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "tri_dev / edge_partner_dev is null");
    TSAMD_HIP(tsamd::launch_antialias_prepare(rast_dev, pos_clip_dev, tri_dev, edge_partner_dev, pair_masks_dev, batch, n_vertices, n_triangles, height, width,
                                              prepared_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_antialias(const float *color_dev, const float *rast_dev, const float *pos_clip_dev, const void *prepared_dev, const int32_t *tri_dev,
                    const int32_t *edge_partner_dev, int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width, int32_t n_channels, float *out_dev,
                    void *stream)
{
    int rc = check_antialias(batch, n_vertices, n_triangles, height, width, n_channels);
    if (rc) return rc;
    const int64_t pixels = batch * int64_t(height) * width;
    if (pixels > 0 && (!color_dev || !rast_dev || !out_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "color_dev / rast_dev / out_dev is null");
    if (pixels > 0 && n_triangles > 0 && (!pos_clip_dev || !tri_dev || !edge_partner_dev))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "pos_clip_dev / tri_dev / edge_partner_dev is null");
    TSAMD_HIP(tsamd::launch_antialias(color_dev, rast_dev, pos_clip_dev, prepared_dev, tri_dev, edge_partner_dev, batch, n_vertices, n_triangles, height, width,
                                      n_channels, out_dev, static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

int tsamd_antialias_backward(const float *color_dev, const float *rast_dev, const float *pos_clip_dev, const void *prepared_dev,
                             const int32_t *tri_dev, const int32_t *edge_partner_dev, int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width,
                             int32_t n_channels, const float *grad_out_dev, float pos_gradient_boost, float *grad_color_dev, float *grad_pos_dev,
                             void *stream)
{
    int rc = check_antialias(batch, n_vertices, n_triangles, height, width, n_channels);
    if (rc) return rc;
    const int64_t pixels = batch * int64_t(height) * width;
    if (!grad_color_dev && !grad_pos_dev) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "grad_color_dev and grad_pos_dev are both null");
    if (pixels > 0 && (!color_dev || !rast_dev || !grad_out_dev)) return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "color_dev / rast_dev / grad_out_dev is null");
    if (pixels > 0 && n_triangles > 0 && (!pos_clip_dev || !tri_dev || !edge_partner_dev))
        return capi_fail(TSAMD_ERR_INVALID_ARGUMENT, "pos_clip_dev / tri_dev / edge_partner_dev is null");
    TSAMD_HIP(tsamd::launch_antialias_backward(color_dev, rast_dev, pos_clip_dev, prepared_dev, tri_dev, edge_partner_dev, batch, n_vertices, n_triangles, height,
                                               width, n_channels, grad_out_dev, pos_gradient_boost, grad_color_dev, grad_pos_dev,
                                               static_cast<hipStream_t>(stream)));
    return TSAMD_OK;
}

}  // extern "C"
