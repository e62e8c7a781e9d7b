/*
 * tssplat_amd -- C ABI of the MI355X (gfx950) tet-sphere geometry-energy path.
 *
 * This is the drop-in boundary for the one hot path of gmh14/tssplat: the
 * per-iteration energy  E = c1 * 1/2 |L G x|^2 + c2 * sum_e max(-det F_e, 0)^p
 * and its analytic gradient.  Every entry point below replaces one piece of the
 * reference's native extension `tet_spheres.tet_spheres_ext`
 * (citations into /root/reference/tssplat_ext/tet_spheres/):
 *
 *   tsamd_create            <- TetSpheres::TetSpheres(int nv, double*, int ntet, int*)   tet_spheres.cpp:119-126
 *                              + TetSpheres::init                                        tet_spheres.cpp:140-203
 *                              + the pybind array constructor                            tet_spheres.cpp:234-258
 *   tsamd_create_with_operator <- the same, with the element Laplacian that the reference takes from libpgo
 *                              (pgo_create_tet_biharmonic_gradient_matrix, tet_spheres.cpp:148) passed as data
 *   tsamd_create_from_veg   <- TetSpheres::TetSpheres(const std::string& filename)       tet_spheres.cpp:108-117
 *   tsamd_destroy           <- TetSpheres::~TetSpheres                                   tet_spheres.cpp:128-138
 *   tsamd_forward           <- tet_spheres_smooth_barrier          (forward)             tet_spheres_cuda.cu:118-195
 *   tsamd_backward          <- tet_spheres_smooth_barrier_backward (backward)            tet_spheres_cuda.cu:197-263
 *   tsamd_forward_backward  <- both of the above fused into one pass (no reference twin:
 *                              the reference recomputes G x in backward, .cu:221)
 *   tsamd_scale             <- the cublasSscal by gradH.item() at                        tet_spheres_cuda.cu:257-258
 *   tsamd_grad_limit        <- tet_spheres_grad_limit (intended semantics, see below)    tet_spheres_cuda.cu:265-303
 *   tsamd_num_vertices/_tets<- TetSpheres::n / ::nele                                    tet_spheres.h:40
 *   tsamd_last_error        <- the stderr line + std::runtime_error of cudaUtils.h:10-44
 *
 * Plain pointers and sizes only; no torch, no C++ types.  All `*_dev` pointers
 * are HIP device pointers on the device the handle was created on; `stream`
 * is a hipStream_t passed as void* (NULL = the null stream).  No entry point
 * synchronises the host with the device except tsamd_create / tsamd_destroy /
 * tsamd_read_energy_terms.  Every function returns 0 on success and a nonzero
 * tsamd_status otherwise; the message is available from tsamd_last_error()
 * (thread-local).  A handle is not re-entrant and is single-stream: its staging rows, per-tile partials and
 * energy terms are per-handle scratch, so evaluations of ONE handle must be ordered (one stream, or events between
 * streams); different handles are independent (same rule as the reference's per-object scratch,
 * tet_spheres.h:36-40).
 */
#ifndef TSSPLAT_AMD_H
#define TSSPLAT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsamd_handle tsamd_handle;

typedef enum tsamd_status {
    TSAMD_OK = 0,
    TSAMD_ERR_INVALID_ARGUMENT = 1,  /* null pointer, negative size, index out of range           */
    TSAMD_ERR_BAD_MESH = 2,          /* non-manifold face (>2 tets), singular rest tet             */
    TSAMD_ERR_NO_DEVICE = 3,         /* no HIP device / not a gfx950-class device                  */
    TSAMD_ERR_HIP = 4,               /* a HIP runtime call failed (text in tsamd_last_error)       */
    TSAMD_ERR_IO = 5,                /* tsamd_create_from_veg: unreadable or malformed file        */
    TSAMD_ERR_TILING = 6,            /* mesh cannot be tiled into the LDS budget                   */
    TSAMD_ERR_HOST_ONLY = 7          /* device entry point called on a host_only handle            */
} tsamd_status;

/* Zero-initialise, set struct_size = sizeof(tsamd_options) and abi_version = TSAMD_ABI_VERSION, override what you need. */
typedef struct tsamd_options {
    int32_t struct_size;
    int32_t device;            /* HIP device ordinal; -1 = current device                          */
    int32_t lds_budget_bytes;  /* LDS per workgroup a tile may use; 0 = 81920 (two workgroups per gfx950 CU) */
    int32_t max_threads;       /* workgroup size cap, multiple of 64, <= 768 (see slots_per_thread for the other lane layouts); 0 = 768 */
    int32_t target_owned;      /* owned tets per tile the partitioner aims for; 0 = auto: the fullest tiles that fit, except that a
                                * plan of <= 1024 tiles built with lds_budget_bytes = max_threads = target_owned = 0 is re-tiled at
                                * 768 (a small batch pays for a tile's latency, not for its halo) */
    int32_t lane_search_sweeps;/* sweeps of the search that seats a tile's tets on the lanes against LDS bank conflicts of the neighbour
                                * gathers (plan time, ~1.5 ms per tile, sweep and host thread; tile kernel -2 %): 0 = 2, -1 = none
                                * (tets stay in Morton order).  Results do not depend on it beyond the order of fp32 sums. */
    int32_t host_only;         /* 1 = build the tiling plan only, never touch HIP (CPU tests)       */
    int32_t num_threads;       /* host threads used to build the plan; 0 = hardware concurrency     */
    int32_t debug_flags;       /* bit1 = no LDS-conflict-aware ordering at all (measurements); other bits ignored */
    int32_t slots_per_thread;  /* tets per lane: 0 = 2.  Lane layouts with a kernel: 2 (max_threads <= 768, two 80 KiB workgroups per
                                * CU: the default, every operator variant); 3 (max_threads <= 512, or <= 1024 for one workgroup per
                                * CU) and 4 (max_threads <= 768): fewer, fatter waves, built-in operator only -- with lds_budget_bytes
                                * up to 163840 these hold a whole ~3 k-tet sphere as ONE tile (no halo, no shared vertices).
                                * Anything else is TSAMD_ERR_INVALID_ARGUMENT. */
    int32_t rebuild_dminv;     /* 0 = auto: stream Dm^-1, except for plans of 513 ... 2 048 tiles built with default tiling options
                                * whose tiles the rebuilt form does not shrink (a few rounds of workgroups of small spheres: 256 x 3 k-tet
                                * spheres 24.8 -> 22.8 us per step), which rebuild it;
                                * 2 = always stream; 1 = always rebuild:
                                * keep rest positions (16 B per tile vertex) instead of the Dm^-1 planes (36 of the
                                * 52 B per tile slot) and invert Dm per slot in fp32 registers: 45 % fewer bytes per
                                * evaluation, entries of Dm^-1 within 2.8e-7 relative of the exact inverse instead of
                                * 5.9e-8 (double -> fp32 rounding, tet_spheres.cpp:43-45).  Not with an explicit operator. */
    int32_t abi_version;       /* must be TSAMD_ABI_VERSION of the header the caller was compiled against: ABI 1 and 2 shared one
                                * struct size while two fields changed their meaning in place, so a stale caller was re-interpreted
                                * instead of rejected.  From ABI 3 on the size changes with the version AND the version is checked. */
} tsamd_options;

/* Introspection of the tiling plan (host side; valid for host_only handles too). */
typedef struct tsamd_plan_info {
    int64_t n_vertices, n_tets;
    int64_t n_tiles;
    int64_t n_components;        /* face-connected components (= tet-spheres)                       */
    int64_t total_slots;         /* sum over tiles of owned + halo tets                             */
    int64_t total_tile_vertices; /* sum over tiles of local vertices                                */
    int64_t shared_vertex_copies;/* tile-vertex copies that go through the staging buffer           */
    int64_t finish_vertices;     /* global vertices written by the finish kernel                    */
    int64_t device_bytes;        /* bytes of plan data resident in HBM                              */
    int32_t max_slots, max_tile_vertices, block_threads, lds_bytes;
    int32_t slots_per_thread;
    int32_t n_planes;            /* dword planes per tile slot: 13; 22 with an explicit element operator (18 when it is symmetric); 4 with rebuild_dminv */
} tsamd_plan_info;

/* One tile of the plan, as host pointers into the handle (valid until tsamd_destroy).
 * Exists so tests can replay the exact data the kernels consume. */
typedef struct tsamd_tile_view {
    int32_t n_slots, n_owned, s_pad, n_verts, n_excl;
    int64_t stage_off;
    int32_t n_rows;            /* rows of the tile's per-vertex force array = most slots any tile vertex meets (<= 64) */
    int32_t rec_base;          /* LDS byte address of record 0: the neighbour tokens are (rec_base + 48 idx) / 4 + rot    */
    const uint32_t *planes;    /* n_planes planes of s_pad dwords: lv01, lv23, nb01, nb23, dminv[0..8]
                                * (+ L[e,e], L[e,n_0..3], L[n_0..3,e] as fp32 with an explicit operator);
                                * a 16-bit vertex field = local vertex | rank << 10                                      */
    const uint16_t *row_start; /* 72 entries: row r of the force array starts at entry row_start[r]; entry (v, r) of
                                * local vertex v = row_start[r] + v (vertices are numbered by falling slot count)        */
    const int32_t *gvid;       /* n_verts global vertex ids (a vertex met by more than 64 slots appears several times)   */
    const int32_t *vdst;       /* n_verts destinations: >= 0 row of grad (the vertex is this tile's alone), < 0: staging
                                * row ~vdst, summed by the finish kernel                                                  */
    const int32_t *slot_tet;   /* s_pad global tet ids (-1 = padding)                                                     */
    const float *rest;         /* rebuild_dminv plans: n_verts x float4 rest positions (tile vertex order), else NULL    */
} tsamd_tile_view;

const char *tsamd_last_error(void);
const char *tsamd_version(void);
/* Bumped whenever a struct layout or a signature of this header changes incompatibly (0.1 had no such call: a caller that
 * cannot find the symbol is talking to an older library).  Compare with TSAMD_ABI_VERSION of the header you compiled against. */
#define TSAMD_ABI_VERSION 3
int32_t tsamd_abi_version(void);

/*
 * rest_xyz: n_vertices*3 float32 (host), tets: n_tets*4 int32 (host, 0-based).
 * Exactly the arrays energies/smooth_barrier.py:38-40 hands to TetSpheres(v_flat, f_flat).
 * The rest shape is frozen here (positions promoted to double, Dm^-1 built in
 * double and rounded to fp32 -- tet_spheres.cpp:252-255, :43-45).
 */
int tsamd_create(const float *rest_xyz, int64_t n_vertices, const int32_t *tets, int64_t n_tets,
                 const tsamd_options *options, tsamd_handle **out);
int tsamd_create_from_veg(const char *path, const tsamd_options *options, tsamd_handle **out);
/*
 * Same as tsamd_create, with the element operator L of the smoothness term H = (L (x) I9) G x given as data
 * (SURVEY 8(b) "optional create inputs: explicit CSR adjacency + weights").  The reference takes L from
 * libpgo -- pgo_create_tet_biharmonic_gradient_matrix(geo, faceNeighbor=1, scale=0), tet_spheres.cpp:148 --
 * whose source is not part of the reference; tsamd_create ASSUMES the uniform face-adjacency umbrella
 * (L[e,e] = number of face neighbours, L[e,e'] = -1).  This entry point substitutes any other operator with
 * the same sparsity (e.g. the matrix libpgo really builds, dumped by tools/pin_L_with_pypgo.py, or the
 * row-scaled scale=1 variant): m x m CSR over tets, double values (rounded to fp32 like the reference's
 * matrix values, tet_spheres.cpp:43-45), 0-based.  Entries must lie on the diagonal or on a face adjacency of
 * the mesh (duplicates are summed); anything else is TSAMD_ERR_INVALID_ARGUMENT.  L need not be symmetric:
 * the gradient applies L^T explicitly.  Cost: 9 more fp32 planes per tile slot (88 instead of 52 bytes); 5 (72 bytes) when
 * the operator is symmetric after the rounding to fp32 -- its weights are then stored once and serve both passes.
 */
int tsamd_create_with_operator(const float *rest_xyz, int64_t n_vertices, const int32_t *tets, int64_t n_tets,
                               const int64_t *op_rowptr, const int32_t *op_col, const double *op_val,
                               const tsamd_options *options, tsamd_handle **out);
void tsamd_destroy(tsamd_handle *h);

int64_t tsamd_num_vertices(const tsamd_handle *h);
int64_t tsamd_num_tets(const tsamd_handle *h);
int tsamd_get_plan_info(const tsamd_handle *h, tsamd_plan_info *out);
int tsamd_get_tile(const tsamd_handle *h, int64_t tile, tsamd_tile_view *out);
/* finish lists: vertex k (global id vid[k]) = sum of staging rows [off[k], off[k+1]);
 * idx[tile.stage_off + j] = the staging row the tile's j-th shared tile vertex (in tile vertex order) writes -- the same
 * rows tsamd_tile_view.vdst names (off[n_finish] entries) */
int tsamd_get_finish_lists(const tsamd_handle *h, int64_t *n_finish, const int32_t **vid,
                           const int32_t **off, const int32_t **idx);
/* face adjacency computed at create time: 4 ints per tet, -1 = boundary face */
int tsamd_get_adjacency(const tsamd_handle *h, const int32_t **nbr);

/*
 * energy_dev: 1 float, receives E = c1*E_s + c2*E_b.  x_dev: n_vertices*3 floats.
 * order: 2 or 4 are meaningful; any other value makes the penalty term 0
 * (tet_spheres_cuda.cu:57-63).
 */
int tsamd_forward(tsamd_handle *h, const float *x_dev, float c1, float c2, int order,
                  void *stream, float *energy_dev);
/*
 * grad_dev: n_vertices*3 floats, overwritten with grad_out * dE/dx.
 * grad_out_dev: 1 float on the device (the autograd grad_output), or NULL for 1.0.
 * Read on the device: no .item() style host sync (contrast tet_spheres_cuda.cu:257).
 */
int tsamd_backward(tsamd_handle *h, const float *x_dev, const float *grad_out_dev, float c1, float c2,
                   int order, void *stream, float *grad_dev);
/* One pass producing both; energy_dev may be NULL. */
int tsamd_forward_backward(tsamd_handle *h, const float *x_dev, const float *grad_out_dev, float c1,
                           float c2, int order, void *stream, float *energy_dev, float *grad_dev);
/*
 * The same evaluation with the two coefficients read ON THE DEVICE: coef_dev = {c1, c2} (2 floats).  The
 * reference's schedule changes c1 and c2 every iteration (energies/smooth_barrier.py:47-58); with the
 * coefficients in device memory a HIP graph captured once replays across iterations (the caller updates
 * coef_dev, e.g. with a captured copy from pinned host memory).  energy_dev or grad_dev may be NULL (not both).
 * No reference twin: the reference launches ~10 library calls per evaluation from the host every time.
 */
int tsamd_evaluate_dev_coef(tsamd_handle *h, const float *x_dev, const float *grad_out_dev, const float *coef_dev,
                            int order, void *stream, float *energy_dev, float *grad_dev);
/*
 * The fused evaluation as an instantiated HIP graph (tile kernel -> finish kernel), for batches whose 10-30 us of kernels
 * drown in per-step host work (the reference pays ~80 us of launches, descriptor set-up and three blocking device reads
 * per iteration at trainer.py:94,130).  tsamd_graph_create bakes in the pointers (x_dev, grad_out_dev -- may be NULL --,
 * energy_dev, grad_dev; at least one of the two outputs) and `order`; tsamd_graph_launch replays it on `stream` with the
 * coefficients of THIS iteration -- they are kernel arguments of the graph's nodes and are updated on the host
 * (hipGraphExecKernelNodeSetParams), so following the reference's schedule (energies/smooth_barrier.py:47-58) costs
 * nothing on the device.  Same kernels, same results as tsamd_forward_backward.  The graph borrows the handle's scratch:
 * the single-stream rule of the handle applies, and the handle must outlive the graph.
 */
typedef struct tsamd_graph tsamd_graph;
int tsamd_graph_create(tsamd_handle *h, const float *x_dev, const float *grad_out_dev, int order, float *energy_dev,
                       float *grad_dev, tsamd_graph **out);
int tsamd_graph_launch(tsamd_graph *graph, float c1, float c2, void *stream);
/* The same replay with its outputs redirected for THIS launch (per-launch arguments of the graph's nodes, like the coefficients):
 *   energy_copy_dev  the energy is ALSO written there (one float; NULL = not): how an energy exchange across GPUs gets a launch's
 *                    value into its own ring slot without a copy on the stream (tssplat_amd/sharding.py: OverlappedEnergyAllReduce);
 *                    requires a graph created with energy_dev;
 *   grad_dev         the gradient goes THERE instead of the buffer the graph was created with ([n_vertices, 3] floats; NULL = the
 *                    graph's own): a caller that hands every evaluation's gradient on (x.grad of a training loop) gets it in a
 *                    fresh buffer without a device copy; requires a graph created with grad_dev. */
int tsamd_graph_launch_to(tsamd_graph *graph, float c1, float c2, void *stream, float *energy_copy_dev, float *grad_dev);
void tsamd_graph_destroy(tsamd_graph *graph);

/* n_iters optimisation steps as ONE HIP graph: per step the evaluation above (energy into energy_ring_dev[k], gradient into
 * grad_dev, upstream gradient 1) followed by tsamd_adam_uniform_step on param_dev with that gradient -- the sequence
 * loss.backward(); optimizer.step() of /root/reference/trainer.py:130-133 for a loss that is the energy alone, with nothing on
 * the host between the steps.  The pointers are baked in (param_dev [n_vertices, 3] is read and updated in place; g1_dev /
 * g2_dev are the optimiser's moments; workspace_dev: tsamd_train_loop_workspace_bytes(n_iters)).  A launch takes the schedule of
 * its n_iters steps -- c1[k], c2[k], order[k] (energies/smooth_barrier.py:47-63), the optimiser's step number of the first one
 * (first_step >= 1: bias corrections 1 - beta^step as in tsamd_adam_uniform_step) and optionally grad_limit[k] (NULL or < 0:
 * none) -- as kernel-node arguments.  Same single-stream rule as the handle; the handle must outlive the loop. */
typedef struct tsamd_train_loop tsamd_train_loop;
int64_t tsamd_train_loop_workspace_bytes(int32_t n_iters);
int tsamd_train_loop_create(tsamd_handle *h, float *param_dev, float *grad_dev, float *g1_dev, float *g2_dev, float *energy_ring_dev,
                            void *workspace_dev, int32_t n_iters, tsamd_train_loop **out);
int tsamd_train_loop_launch(tsamd_train_loop *loop, const float *c1, const float *c2, const int32_t *order, float lr, float beta1, float beta2,
                            int64_t first_step, const float *grad_limit, void *stream);
void tsamd_train_loop_destroy(tsamd_train_loop *loop);

/* Blocking read of the last evaluation's (E_s, E_b) in double, for diagnostics. */
int tsamd_read_energy_terms(tsamd_handle *h, void *stream, double *terms_host2);

/*
 * Kernel timing for bench.py's roofline leg.  While enabled, every evaluation records HIP
 * events on its stream around the tile kernel and the finish kernel (no host sync).
 * tsamd_get_timing synchronises those events, returns the accumulated milliseconds and the
 * number of evaluations, and clears the record.
 */
int tsamd_set_timing(tsamd_handle *h, int enable);
int tsamd_get_timing(tsamd_handle *h, double *tile_kernel_ms, double *finish_kernel_ms, int64_t *evaluations);

/* out[i] = in[i] * (*scalar_dev); in == out allowed. */
int tsamd_scale(const float *in_dev, const float *scalar_dev, float *out_dev, int64_t n, void *stream);
/*
 * In-place step clamp with the semantics the reference intended
 * (utils/optimizer.py:84-86: if max|g| > threshold: g *= s / max|g|), not the
 * shipped bug that reads g[0] (tet_spheres_cuda.cu:278).  No host sync.
 * workspace_dev: >= tsamd_grad_limit_workspace_bytes() bytes of scratch.
 */
int64_t tsamd_grad_limit_workspace_bytes(void);
int tsamd_grad_limit(float *grad_dev, int64_t n, float s_threshold, float s, void *workspace_dev,
                     void *stream);

/*
 * SURVEY 8(f) row 3 -- the optimiser step that follows every backward: one fused AdamUniform step
 * (reference utils/optimizer.py:38-89) over a contiguous float32 parameter, in place, no host sync
 * (the reference takes two .max() reductions and, with grad_limit, a Python `if s > m` per step).
 *   g1 = b1 g1 + (1-b1) grad;  g2 = b2 g2 + (1-b2) grad^2;            optimizer.py:61-62
 *   gr = (g1 / (1-b1^step)) / (1e-8 + max sqrt(g2 / (1-b2^step)));    optimizer.py:67-74
 *   if grad_limit > 0 and max|gr| > grad_limit: gr *= grad_limit / max|gr|;   optimizer.py:84-86
 *   param -= lr * gr                                                   optimizer.py:88
 * `step` is the 1-based count AFTER the increment of optimizer.py:55.  workspace_dev: >= 16 bytes.
 */
int tsamd_adam_uniform_step(float *param_dev, const float *grad_dev, float *g1_dev, float *g2_dev, int64_t n,
                            float lr, float beta1, float beta2, int64_t step, float grad_limit,
                            void *workspace_dev, void *stream);

/*
 * SURVEY 8(f) row 2 -- the surface glue around the energy in every iteration
 * (reference geometry/tetmesh_geometry.py:27-66) and the one-off boundary extraction
 * (geometry/mesh_utils.py:5-35).  All kernels are per-vertex gathers over fixed lists: no atomics,
 * bitwise repeatable.  float32 values, int32 indices, [k,3] arrays row-major and contiguous.
 */
typedef struct tsamd_surface tsamd_surface;

/* get_surface_vf(elem) (mesh_utils.py:5-35), host only: boundary triangles (faces of exactly one tet) ordered
 * by their sorted vertex triple, oriented like the tet-local pattern they come from, vertex ids compacted to
 * their rank in `surface_vid` (ascending tet-vertex ids).  Call once with surface_vid = faces = NULL to get the
 * sizes, then with buffers of *n_surface_vertices and 3 * *n_faces entries (capacities passed in the counters). */
int tsamd_extract_surface(const int32_t *tets, int64_t n_tets, int64_t n_vertices, int32_t *surface_vid,
                          int64_t *n_surface_vertices, int32_t *faces, int64_t *n_faces);

/* Device copy of the surface topology: surface_vid[ns] (TetMeshGeometry.surface_vid, tetmesh_geometry.py:146),
 * faces[nf,3] indexing surface vertices (surface_fid, :148).  Host pointers; device < 0 = current device. */
int tsamd_surface_create(const int32_t *surface_vid, int64_t n_surface_vertices, const int32_t *faces, int64_t n_faces,
                         int64_t n_tet_vertices, int device, tsamd_surface **out);
void tsamd_surface_destroy(tsamd_surface *s);

/* v_pos = tet_v[surface_vid]  (tetmesh_geometry.py:33) and its adjoint: grad_tet_v[n_tet_vertices,3] is
 * overwritten (zero, then the rows of grad_v_pos scattered; accumulated if surface_vid has duplicates). */
int tsamd_surface_positions(const tsamd_surface *s, const float *tet_v_dev, void *stream, float *v_pos_dev);
int tsamd_surface_positions_backward(const tsamd_surface *s, const float *grad_v_pos_dev, void *stream,
                                     float *grad_tet_v_dev);

/* _compute_vertex_normal() (tetmesh_geometry.py:39-66): face normals (v1-v0)x(v2-v0) summed per vertex,
 * |n|^2 <= 1e-20 -> (0,0,1), n / max(|n|, 1e-12).  raw_dev (optional, [ns,3]) receives the unnormalised sums,
 * which the backward needs.  Backward: grad_v_pos = (d v_nrm / d v_pos)^T grad_nrm; workspace: 12 * ns bytes. */
int tsamd_vertex_normals(const tsamd_surface *s, const float *v_pos_dev, void *stream, float *v_nrm_dev, float *raw_dev);
int tsamd_vertex_normals_backward(const tsamd_surface *s, const float *v_pos_dev, const float *raw_dev,
                                  const float *grad_nrm_dev, void *workspace_dev, void *stream, float *grad_v_pos_dev);

/* ------------------------------------------------------------------------------------------------------------------
 * Renderer slice: the nvdiffrast operators the reference's renderer calls,
 *   tsamd_rasterize            <- dr.rasterize(ctx, pos_clip, tri, resolution=[H, W], grad_db=False)[0]   renderers/mesh_rasterizer.py:103
 *   tsamd_interpolate          <- dr.interpolate(attr, rast, tri)[0]                                     renderers/mesh_rasterizer.py:117,145,153
 *   tsamd_interpolate_backward <- the backward of the latter w.r.t. attr and rast's (u, v)
 *   tsamd_rasterize_backward, tsamd_antialias* (below) <- the backward of dr.rasterize, dr.antialias and its backward   :103,107,128
 * nvdiffrast is a separate library (not vendored by the reference, version unpinned): the semantics implemented are a
 * restatement of its published algorithm, fixed in every detail by oracle/raster_oracle.py -- clip-space input, OpenGL
 * conventions (row 0 = bottom), no culling, nearest depth, output (u, v, z/w, triangle_id + 1), 0 = background.  PARITY
 * UNPINNED (no nvdiffrast here).  Clipping: against the near plane (z + w >= 0) for triangles with a vertex at w <= 0 (the
 * clipped polygon is rasterised under the triangle's own id; pixels beyond the far plane or the image fail their own tests).
 * Not offered: depth peeling, `ranges`, image-space derivatives (grad_db).  Limits: height, width <= 8192 (window coordinates
 * are snapped to 1/256 pixel and kept within +-16384 pixels; a triangle with a vertex beyond that guard band, or a non-finite
 * one, is dropped whole), n_triangles <= 2^24 - 1 (the id + 1 travels as a float32).  Stateless: the caller owns all buffers and
 * the current HIP device is used.  pos_clip_dev: [batch, n_vertices, 4] f32; tri_dev: [n_triangles, 3] i32;
 * rast: [batch, height, width, 4] f32; attr_dev: [attr_batch (1 or batch), n_vertices, n_channels] f32.
 */
/* workspace: one 64-bit depth key per pixel, one 16-byte snapped vertex per (view, vertex), one flag per view */
int64_t tsamd_rasterize_workspace_bytes(int64_t batch, int64_t n_vertices, int32_t height, int32_t width);
/* pair_masks_out_dev (OPTIONAL, tsamd_pair_masks_bytes(batch, height, width) bytes): a by-product for tsamd_antialias_prepare --
 * per 64 consecutive pixels two 64-bit words, bit l = pixel 64 k + l and its right (word 0) / upper (word 1) neighbour show
 * two different triangles.  The resolve pass has the ids in hand; without it the antialias side reads `rast` back to find them. */
int64_t tsamd_pair_masks_bytes(int64_t batch, int32_t height, int32_t width);
int tsamd_rasterize(const float *pos_clip_dev, int64_t batch, int64_t n_vertices, const int32_t *tri_dev, int64_t n_triangles,
                    int32_t height, int32_t width, void *workspace_dev, float *rast_out_dev, void *pair_masks_out_dev, void *stream);
/* A pixel whose id names no triangle of tri_dev (id > n_triangles, e.g. a rast image made with another list) or a triangle
 * with a vertex index outside [0, n_vertices) is treated as background: zero output, zero gradient, nothing read or written
 * out of bounds. */
int tsamd_interpolate(const float *attr_dev, int64_t attr_batch, int64_t n_vertices, int32_t n_channels, const float *rast_dev,
                      const int32_t *tri_dev, int64_t n_triangles, int64_t batch, int32_t height, int32_t width, float *out_dev, void *stream);
/* grad_attr_dev ([attr_batch, n_vertices, n_channels]) is zero-filled and accumulated by the call; grad_rast_dev
 * ([batch, height, width, 4], channels 2-3 = 0) may be NULL. */
int tsamd_interpolate_backward(const float *attr_dev, int64_t attr_batch, int64_t n_vertices, int32_t n_channels, const float *rast_dev,
                               const int32_t *tri_dev, int64_t n_triangles, int64_t batch, int32_t height, int32_t width, const float *grad_out_dev,
                               float *grad_attr_dev, float *grad_rast_dev, void *stream);

/* Gradient of tsamd_rasterize w.r.t. pos_clip from the gradient of its (u, v) outputs (what flows back from
 * tsamd_interpolate_backward's grad_rast; z/w and the id carry none).  grad_pos_dev [batch, n_vertices, 4] is zero-filled and
 * accumulated by the call.  <- the backward of dr.rasterize(..., grad_db=False), renderers/mesh_rasterizer.py:103 */
int tsamd_rasterize_backward(const float *pos_clip_dev, int64_t batch, int64_t n_vertices, const int32_t *tri_dev, int64_t n_triangles,
                             int32_t height, int32_t width, const float *rast_dev, const float *grad_rast_dev, float *grad_pos_dev, void *stream);

/* tsamd_antialias <- dr.antialias(color, rast, pos_clip, tri, topology_hash=None, pos_gradient_boost=1.0)   renderers/mesh_rasterizer.py:107,128
 * -- the only differentiable path from the alpha image to the geometry.  Every pair of adjacent pixels with different
 * triangle ids is analysed: the closer triangle's silhouette edge that passes between the two pixel centres blends the two
 * colours by the position of the crossing (specification: oracle/raster_oracle.py; PARITY UNPINNED like the rest of the slice).
 * edge_partner_dev [3 * n_triangles] i32 is the topology table (nvdiffrast's topology hash): built once per triangle list by
 * tsamd_antialias_topology with a workspace of tsamd_antialias_topology_workspace_bytes(n_triangles).
 * color_dev / out_dev / grad_*: [batch, height, width, n_channels] f32.  The backward recomputes the analysis (no state is
 * kept between the calls); grad_color_dev and grad_pos_dev ([batch, n_vertices, 4], zero-filled first) may each be NULL.
 * prepared_dev is OPTIONAL (NULL: everything is computed per use, identical results): tsamd_antialias_prepare fills it
 * (tsamd_antialias_prepared_bytes bytes) from the same rast_dev / pos_clip_dev / tri_dev / edge_partner_dev / sizes with (1) the
 * window coordinates of every (view, vertex) in float64, (2) a 2-bit-per-pixel mask of the pixel pairs with two different
 * triangle ids, (3) per (view, triangle) which of its edges can blend at all (no partner, or a fold).  One prepared buffer
 * serves the forward and the backward call: the image is scanned once instead of twice, the analysis of a pair on an
 * interior triangle boundary ends at one byte, and no float64 division is left in it. */
int64_t tsamd_antialias_topology_workspace_bytes(int64_t n_triangles);
int tsamd_antialias_topology(const int32_t *tri_dev, int64_t n_triangles, void *workspace_dev, int32_t *edge_partner_dev, void *stream);
int64_t tsamd_antialias_prepared_bytes(int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width);
/* pair_masks_dev: tsamd_rasterize's by-product for the SAME rast image, or NULL (rast_dev is scanned) */
int tsamd_antialias_prepare(const float *rast_dev, const float *pos_clip_dev, const int32_t *tri_dev, const int32_t *edge_partner_dev,
                            const void *pair_masks_dev, int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width,
                            void *prepared_dev, void *stream);
int tsamd_antialias(const float *color_dev, const float *rast_dev, const float *pos_clip_dev, const void *prepared_dev, const int32_t *tri_dev,
                    const int32_t *edge_partner_dev, int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width, int32_t n_channels, float *out_dev,
                    void *stream);
int tsamd_antialias_backward(const float *color_dev, const float *rast_dev, const float *pos_clip_dev, const void *prepared_dev,
                             const int32_t *tri_dev, const int32_t *edge_partner_dev, int64_t batch, int64_t n_vertices, int64_t n_triangles, int32_t height, int32_t width,
                             int32_t n_channels, const float *grad_out_dev, float pos_gradient_boost, float *grad_color_dev, float *grad_pos_dev,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TSSPLAT_AMD_H */
