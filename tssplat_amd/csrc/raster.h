// Launch interface between the C ABI (raster_capi.cpp) and the renderer-slice kernels (raster_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace tsamd {

// workspace: batch * height * width 64-bit depth keys, then batch * n_vertices 16-byte snapped vertices
hipError_t launch_rasterize(const float *pos_clip, int64_t batch, int64_t n_vertices, const int32_t *tri, int64_t n_tri, int height, int width,
                            void *workspace, float *rast, hipStream_t stream);
hipError_t launch_interpolate(const float *attr, int64_t attr_batch, int64_t n_vertices, int channels, const float *rast, const int32_t *tri,
                              int64_t batch, int height, int width, float *out, hipStream_t stream);
// grad_attr is zero-filled by the launch; grad_rast may be null
hipError_t launch_interpolate_backward(const float *attr, int64_t attr_batch, int64_t n_vertices, int channels, const float *rast, const int32_t *tri,
                                       int64_t batch, int height, int width, const float *grad_out, float *grad_attr, float *grad_rast,
                                       hipStream_t stream);

}  // namespace tsamd
