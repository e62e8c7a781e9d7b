// Launch interface between the C ABI (stream_capi.cpp) and the streaming-tile kernels (stream_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "stream_plan.h"

namespace tsamd {

struct StreamEvalArgs {
    const StreamTubeDesc *tubes;   // device
    const uint8_t *blob;           // device
    const int32_t *fin_vid, *fin_off;
    int64_t n_tubes, n_finish;
    int32_t lds_bytes;
    const float *x;
    const float *grad_out;         // device scalar or nullptr
    float c1, c2;
    int order;
    float *grad;
    float *stage;                  // [n_stage, 3]
    double *partials;              // [n_tubes, 2]
    float *energy;
    double *terms;                 // [2]
};

int32_t stream_lds_bytes(int32_t max_vslots, int32_t max_bands);
hipError_t configure_stream_kernels(int lds_bytes);
// ev: optional 3 events (before the tube kernel, between the kernels, after the finish kernel)
hipError_t launch_stream_eval(const StreamEvalArgs &a, hipStream_t stream, hipEvent_t *ev = nullptr);

}  // namespace tsamd
