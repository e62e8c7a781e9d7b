// Launch interface between the C ABI (capi.cpp) and the gfx950 kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "plan.h"

namespace tsamd {

struct EvalArgs {
    // plan, resident in HBM
    const TileDesc *tiles;
    const uint8_t *blob;
    const int32_t *gvid, *vdst;
    const int32_t *fin_vid, *fin_off;
    int64_t n_tiles, n_finish;
    int32_t block_threads, lds_bytes;
    int32_t vert_stride = 0;      // gvid entries per tile (Plan::vert_stride)
    bool rebuild = false;         // the plan keeps rest positions instead of the Dm^-1 planes (kPlanesRebuild)
    bool weighted = false;        // the plan carries an explicit element operator (22 planes per slot; 18 when it is symmetric)
    int32_t n_planes = 13;        // dword planes per slot of this plan (where a tile's incidence list starts)
    int32_t spt = kSlotsPerLane;  // slots per lane the plan is laid out for (selects the kernel instantiation)
    // per evaluation
    const float *x;
    const float *grad_out;  // device scalar or nullptr
    float c1, c2;
    const float *coef = nullptr;  // device (c1, c2) overriding the two values above when non-null
    int order;
    float *grad;            // nullptr = energy only
    float *stage;           // [n_stage, 3]
    double *partials;       // [n_tiles, 2]
    float *energy;          // nullptr = skip
    double *terms;          // [2] (E_s, E_b), always written when energy != nullptr
};

// ev: optional 3 events recorded before the tile kernel, between the two kernels, after the finish kernel
hipError_t launch_eval(const EvalArgs &a, hipStream_t stream, hipEvent_t *ev = nullptr);
// The same evaluation as an instantiated HIP graph whose coefficients are updated per launch as kernel-node arguments.
// The EvalArgs pointers (x, grad_out, energy, grad, plan data) are baked in; e.coef must be null.
struct EvalGraph;
hipError_t eval_graph_create(const EvalArgs &a, EvalGraph **out);
hipError_t eval_graph_launch(EvalGraph *g, float c1, float c2, hipStream_t stream, float *energy_copy = nullptr, float *grad = nullptr);
void eval_graph_destroy(EvalGraph *g);
// n_iters optimisation steps -- energy + gradient, then the AdamUniform update of x from that gradient -- as ONE graph: per
// step a tile node, a finish node and the two optimiser nodes, chained; x is updated in place and read by the next step.
// e.x = the parameter, e.grad = its gradient buffer, e.energy = a ring of n_iters floats (step k writes slot k); ws = 2 * n_iters
// dwords.  A launch takes the per-step coefficients, orders and optimiser scalars as kernel-node arguments.
struct TrainLoopGraph;
struct TrainLoopStep {
    float c1, c2;
    int order;
    float bias1, bias2, limit;   // AdamUniform: 1 - beta^step, and the step limit (< 0: none)
};
hipError_t train_loop_create(const EvalArgs &e, float *param, float *g1, float *g2, int64_t n_param, void *ws, int n_iters, TrainLoopGraph **out);
hipError_t train_loop_launch(TrainLoopGraph *g, const TrainLoopStep *steps, float lr, float b1, float b2, hipStream_t stream);
void train_loop_destroy(TrainLoopGraph *g);
hipError_t launch_scale(const float *in, const float *scalar, float *out, int64_t n, hipStream_t stream);
hipError_t launch_grad_limit(float *grad, int64_t n, float thr, float s, void *workspace, hipStream_t stream);
hipError_t launch_adam_uniform(float *p, const float *grad, float *g1, float *g2, int64_t n, float lr, float b1, float b2,
                               float bias1, float bias2, float limit, void *workspace, hipStream_t stream);
hipError_t configure_kernels(int lds_bytes);
// is there a tile kernel for `spt` slots per lane and workgroups of up to `max_threads` threads?
bool lane_layout_supported(int spt, int max_threads);

}  // namespace tsamd
