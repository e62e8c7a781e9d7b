/*
 * tssplat_amd -- EXPERIMENTAL and DIAGNOSTIC entry points of libtssplat_amd.so.
 *
 * Nothing in this header is part of the drop-in boundary (include/tssplat_amd.h): the product path never calls
 * these, their signatures may change between rounds, and the diagnostic switches make results WRONG on purpose.
 * They are exported so that the measurement tools (tools/ablate.py, tools/bench_stream.py) and the tests that keep
 * the parked streaming-tile experiment honest (tests/test_stream_*.py) can reach them.
 */
#ifndef TSSPLAT_AMD_EXPERIMENTAL_H
#define TSSPLAT_AMD_EXPERIMENTAL_H

#include "tssplat_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic only (tools/ablate.py): switch parts of the tile kernel off to price them.
 * Any nonzero value makes results WRONG; production code never calls this. */
int tsamd_debug_set_ablation(tsamd_handle *h, int flags);
/* Diagnostic only, meaningful in -DTSAMD_ABLATION builds: the first call arms 16 shader-clock
 * stamps per wave (phase boundaries seen by lane 0 of each of up to 16 waves of a workgroup), later calls
 * copy the stamps of the most recent evaluation to host_out (capacity >= 256 * n_tiles,
 * index (16 * tile + wave) * 16 + stamp). */
int tsamd_debug_read_clocks(tsamd_handle *h, long long *host_out, int64_t capacity);


/* ------------------------------------------------------------------------------------------------------------------
 * Streaming tiles (EXPERIMENTAL, round 3): the same energy and gradient as tsamd_forward_backward -- same reference
 * entry points replaced, tet_spheres_cuda.cu:118-263 -- from another plan and another tile kernel.  A tet-sphere is cut
 * into a few tubes, each swept level by level (breadth-first over face adjacency: neighbours are within +-1 level) with
 * a rolling window of band records in LDS and four wave groups running four stages on four bands per barrier interval
 * (tssplat_amd/csrc/stream_plan.h).  1.08 instead of 1.28 tile slots per tet on the headline scene.  Its own handle
 * type; tsamd_create / TetSpheres remain the product path: measured on the headline scene this path takes 0.73 ms
 * against 0.43 ms (profiles/r03_experiments.md).  Built-in uniform
 * operator only.  TSAMD_ERR_TILING = a component cannot be cut into tubes whose widest level fits a band: use tsamd_create.
 */
typedef struct tsamd_stream tsamd_stream;
typedef struct tsamd_stream_plan_info {
    int64_t n_vertices, n_tets, n_components, n_tubes;
    int64_t total_slots;           /* owned + side-halo tets over all tubes                         */
    int64_t total_bands, total_pairs, total_chunks;
    int64_t shared_vertex_copies, finish_vertices;
    int64_t device_bytes, blob_bytes;
    int32_t max_vertex_slots, max_bands, band_slots, lds_bytes;
} tsamd_stream_plan_info;
/* One tube as host pointers into the handle (tests replay exactly the data the kernel consumes; layout: stream_plan.h). */
typedef struct tsamd_stream_tube_view {
    int32_t n_bands, n_vslots, n_owned, n_slots;
    const uint8_t *blob;         /* n_bands band descriptors (24 B each), then the bands' planes and lists */
    int64_t blob_bytes;
    const int32_t *slot_tet;     /* n_bands x band_slots global tet ids (-1 = padding)              */
} tsamd_stream_tube_view;
int tsamd_stream_create(const float *rest_xyz, int64_t n_vertices, const int32_t *tets, int64_t n_tets, int32_t device,
                        int32_t host_only, int32_t num_threads, tsamd_stream **out);
void tsamd_stream_destroy(tsamd_stream *h);
int tsamd_stream_info(const tsamd_stream *h, tsamd_stream_plan_info *out);
int tsamd_stream_get_tube(const tsamd_stream *h, int64_t tube, tsamd_stream_tube_view *out);
int tsamd_stream_get_finish_lists(const tsamd_stream *h, int64_t *n_finish, int64_t *n_stage, const int32_t **vid, const int32_t **off);
int tsamd_stream_forward_backward(tsamd_stream *h, const float *x_dev, const float *grad_out_dev, float c1, float c2, int order,
                                  void *stream, float *energy_dev, float *grad_dev);
int tsamd_stream_set_timing(tsamd_stream *h, int enable);
int tsamd_stream_get_timing(tsamd_stream *h, double *tube_kernel_ms, double *finish_kernel_ms, int64_t *evaluations);
int tsamd_stream_read_energy_terms(tsamd_stream *h, void *stream, double *terms_host2);

#ifdef __cplusplus
}
#endif
#endif /* TSSPLAT_AMD_EXPERIMENTAL_H */
