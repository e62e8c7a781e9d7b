// Launch interface between the C ABI (raster_capi.cpp) and the renderer-slice kernels (raster_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace tsamd {

// workspace: batch * height * width 64-bit depth keys, then batch * n_vertices 16-byte snapped vertices
// pair_masks (optional output, 16 bytes per 64 pixels): see antialias_prepare
hipError_t launch_rasterize(const float *pos_clip, int64_t batch, int64_t n_vertices, const int32_t *tri, int64_t n_tri, int height, int width,
                            void *workspace, float *rast, void *pair_masks, hipStream_t stream);
// grad_pos ([batch, n_vertices, 4]) is zero-filled by the launch
hipError_t launch_rasterize_backward(const float *pos_clip, int64_t batch, int64_t n_vertices, const int32_t *tri, int64_t n_tri, int height, int width,
                                     const float *rast, const float *grad_rast, float *grad_pos, hipStream_t stream);
// antialias (aa_kernels.hip): the edge partner table opp[3 * n_tri] is built once per triangle list
int64_t antialias_topology_workspace_bytes(int64_t n_tri);
hipError_t launch_antialias_topology(const int32_t *tri, int64_t n_tri, void *workspace, int32_t *opp, hipStream_t stream);
// `prepared` (antialias_prepared_bytes): the window coordinates of every (view, vertex) and the mask of pixel pairs with two
// different triangle ids and the per-(view, triangle) edge flags; optional input of the two launches below (null: both computed per use, same operations)
int64_t antialias_prepared_bytes(int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width);
int64_t pair_masks_bytes(int64_t batch, int height, int width);
// pair_masks: the by-product of launch_rasterize for the same image, or null (then `rast` is scanned for them)
hipError_t launch_antialias_prepare(const float *rast, const float *pos_clip, const int32_t *tri, const int32_t *opp, const void *pair_masks, int64_t batch,
                                    int64_t n_vertices, int64_t n_tri, int height, int width, void *prepared, hipStream_t stream);
hipError_t launch_antialias(const float *color, const float *rast, const float *pos_clip, const void *prepared, const int32_t *tri, const int32_t *opp,
                            int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width, int channels, float *out, hipStream_t stream);
// grad_color (a copy of grad_out plus the blends' terms) and grad_pos (zero-filled first) may each be null
hipError_t launch_antialias_backward(const float *color, const float *rast, const float *pos_clip, const void *prepared, const int32_t *tri,
                                     const int32_t *opp, int64_t batch, int64_t n_vertices, int64_t n_tri, int height, int width, int channels, const float *grad_out, float boost,
                                     float *grad_color, float *grad_pos, hipStream_t stream);
hipError_t launch_interpolate(const float *attr, int64_t attr_batch, int64_t n_vertices, int channels, const float *rast, const int32_t *tri,
                              int64_t n_tri, int64_t batch, int height, int width, float *out, hipStream_t stream);
// grad_attr is zero-filled by the launch; grad_rast may be null
hipError_t launch_interpolate_backward(const float *attr, int64_t attr_batch, int64_t n_vertices, int channels, const float *rast, const int32_t *tri,
                                       int64_t n_tri, int64_t batch, int height, int width, const float *grad_out, float *grad_attr, float *grad_rast,
                                       hipStream_t stream);

}  // namespace tsamd
