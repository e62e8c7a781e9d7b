// What does an LDS bank conflict cost on gfx950, and do the tile kernel's gather phases overlap LDS and VALU?
//
// Part 1 (cost model of a conflicting ds_read_b128 / ds_read_b32).  A wave64 ds_read_b128 is served in four fixed
// groups of 16 lanes (MI355X_MICROARCH.md, LDS table); lanes of one group that hit the same 16-byte column at
// different rows serialise.  Which of these is the price of an instruction:
//     sum over the groups of (max multiplicity in the group)      -- groups are served one after the other, or
//     4 x (max over the groups)                                   -- the slowest group paces all four?
// and is a column hit twice in a group as bad when it is ONE column (1 extra cycle) as when it is all of them?
// Patterns below put k lanes of chosen groups on one column (different rows), everything else conflict-free.
//
// Part 2 (the tile kernel's phase shape).  Per "slot": R dependent-address gathers (8 ds_read_b128 + 4 ds_read_b32
// with the measured ~1.6x conflict factor), s_waitcnt lgkmcnt(0), then V VALU instructions in a dependent 3x3-like
// chain; two slots per wave and phase, an s_barrier after every phase -- at the tile kernel's occupancy (2 workgroups x
// 12 waves per CU).  Time per phase with LDS only, VALU only, both: max(...) = overlap, sum = none.
//
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench_conflicts.hip -o tools/_bin/ubench_conflicts && tools/_bin/ubench_conflicts
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                           \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) {                                            \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                       \
        }                                                                  \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// ds_read_b128 lane groups of gfx950
__host__ __device__ inline void group_of(int lane, int &g, int &i)
{
    const int h = lane >> 5, l = lane & 31;
    static const signed char tab[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
    g = 2 * h + tab[l];
    int k = 0;
    for (int j = 0; j < l; ++j) k += tab[j] == tab[l];
    i = k;
}

// offs[lane] = byte offset (within the wave's 4 KiB window: 16 rows x 256 B) of the lane's 16-byte read
template <int WIDTH>
__global__ __launch_bounds__(768) void conflict_probe(const unsigned *offs, float *out, int iters)
{
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned base = wave * 4096u + offs[lane];
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        v4f r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (WIDTH == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(r[k]) : "v"(base));
            if (WIDTH == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(r[k].x) : "v"(base));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int k = 0; k < 8; ++k) s += r[k].x;
    }
    if (s == 123.456f) out[0] = s + smem[0];
}

static float *g_out = nullptr;
static unsigned *g_offs = nullptr;

template <int WIDTH>
static double run_conflict(const char *what, const unsigned *offs)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 2, iters = 4000;
    CHECK(hipMemcpy(g_offs, offs, 64 * 4, hipMemcpyHostToDevice));
    auto k = conflict_probe<WIDTH>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(768), 80 * 1024, 0, g_offs, g_out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(768), 80 * 1024, 0, g_offs, g_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_inst = double(ms) * 1e6 / iters / (24.0 * 8.0);   // per wave-instruction per CU
    printf("%-66s b%-3d %7.3f ns per wave-instruction per CU = %5.2f cycles at 2.4 GHz\n", what, WIDTH * 8, ns_inst, ns_inst * 2.4);
    return ns_inst;
}

// pattern: lanes of group g with in-group index < k[g] all read column 0 at rows 0..k-1; the others read their own column
static void make_pattern(unsigned *offs, const int k[4])
{
    for (int lane = 0; lane < 64; ++lane) {
        int g, i;
        group_of(lane, g, i);
        if (i < k[g])
            offs[lane] = unsigned(i) * 256u + 0u;   // column 0, row i
        else
            offs[lane] = 15u * 256u + unsigned(i) * 16u;   // row 15, own column (column 0 of row 15 is free: i >= 1 here or k = 0)
        if (k[g] == 0) offs[lane] = 15u * 256u + unsigned(i) * 16u;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Part 2: the phase shape
template <int V, int R, bool BARRIER>
__global__ __launch_bounds__(768) void phase_probe(const unsigned *tok, float *out, int iters)
{
    extern __shared__ unsigned char smem[];
    // every lane holds 2 slots x 4 neighbour "tokens" (byte addresses of 48-byte records, random within the tile)
    unsigned nb[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) nb[p][k] = tok[(blockIdx.x & 1) * 768 * 8 + (threadIdx.x * 2 + p) * 4 + k];
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = float(j) + threadIdx.x;
    const float b = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (R) {
                v4f q[8];
                float t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(q[2 * k]) : "v"(nb[p][k] & ~15u));
                    asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(q[2 * k + 1]) : "v"(nb[p][k] & ~15u));
                    asm volatile("ds_read_b32 %0, %1" : "=v"(t[k]) : "v"(nb[p][k]));
                }
                asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[0] -= q[2 * k].x; acc[1] -= q[2 * k].y; acc[2] -= q[2 * k].z; acc[3] -= q[2 * k].w;
                    acc[4] -= q[2 * k + 1].x; acc[5] -= q[2 * k + 1].y; acc[6] -= q[2 * k + 1].z; acc[7] -= q[2 * k + 1].w;
                    acc[8] -= t[k];
                }
            }
            // V more VALU instructions: 3x3-like products, each result feeding the next row (dependent chains of 3)
#pragma unroll
            for (int v = 0; v < V; ++v)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[v % 9]) : "v"(acc[(v + 3) % 9]), "v"(b));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) s += acc[j];
    if (s == 123.456f) out[0] = s + smem[0];
}

static unsigned *g_tok = nullptr;

template <int V, int R, bool BARRIER>
static double run_phase(const char *what)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 2, iters = 2000;
    auto k = phase_probe<V, R, BARRIER>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(768), 80 * 1024, 0, g_tok, g_out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(768), 80 * 1024, 0, g_tok, g_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = double(ms) * 1e6 / iters;   // per phase (2 slots per lane) per CU, 2 workgroups resident
    printf("%-58s V=%3d R=%d barrier=%d: %8.1f ns per phase per CU = %7.0f cycles at 2.4 GHz  (%5.2f cycles per slot)\n", what, V, R, int(BARRIER), ns,
           ns * 2.4, ns * 2.4 / (2.0 * 1536.0));
    return ns;
}

int main()
{
    CHECK(hipMalloc(&g_out, 4));
    CHECK(hipMalloc(&g_offs, 64 * 4));
    unsigned offs[64];
    struct {
        const char *what;
        int k[4];
    } pats[] = {
        {"conflict-free", {0, 0, 0, 0}},
        {"group 0: 2 lanes on one column", {2, 0, 0, 0}},
        {"groups 0,1: 2 lanes on one column", {2, 2, 0, 0}},
        {"all groups: 2 lanes on one column", {2, 2, 2, 2}},
        {"group 0: 3 lanes on one column", {3, 0, 0, 0}},
        {"group 0: 4 lanes on one column", {4, 0, 0, 0}},
        {"group 0: 4, group 2: 4", {4, 0, 4, 0}},
        {"all groups: 4 lanes on one column", {4, 4, 4, 4}},
        {"group 0: 8 lanes on one column", {8, 0, 0, 0}},
        {"all groups: 8 lanes on one column", {8, 8, 8, 8}},
        {"group 0: 16 lanes on one column", {16, 0, 0, 0}},
    };
    for (auto &p : pats) {
        make_pattern(offs, p.k);
        run_conflict<16>(p.what, offs);
    }
    // every column twice in every group (8 columns x 2 rows)
    for (int lane = 0; lane < 64; ++lane) {
        int g, i;
        group_of(lane, g, i);
        offs[lane] = unsigned(i & 1) * 256u + unsigned(i >> 1) * 16u;
    }
    run_conflict<16>("all groups: every column twice (8 columns x 2 rows)", offs);
    // random columns and rows (what an unplanned gather looks like)
    srand(12345);
    double acc_ns = 0;
    for (int rep = 0; rep < 4; ++rep) {
        for (int lane = 0; lane < 64; ++lane) offs[lane] = unsigned(rand() % 16) * 256u + unsigned(rand() % 16) * 16u;
        acc_ns += run_conflict<16>("random columns", offs);
    }
    printf("random columns, mean of 4 patterns: %.3f ns\n", acc_ns / 4);
    // ds_read_b32: groups of 32 lanes, 32 banks
    for (int lane = 0; lane < 64; ++lane) offs[lane] = 15u * 256u + unsigned(lane & 31) * 4u + unsigned(lane >> 5) * 128u;
    run_conflict<4>("b32 conflict-free", offs);
    for (int lane = 0; lane < 64; ++lane) offs[lane] = (lane & 31) < 2 ? unsigned(lane & 31) * 256u + unsigned(lane >> 5) * 128u : 15u * 256u + unsigned(lane & 31) * 4u + unsigned(lane >> 5) * 128u;
    run_conflict<4>("b32: 2 lanes of each half on one bank", offs);
    for (int lane = 0; lane < 64; ++lane) offs[lane] = (lane & 31) < 4 ? unsigned(lane & 31) * 256u + unsigned(lane >> 5) * 128u : 15u * 256u + unsigned(lane & 31) * 4u + unsigned(lane >> 5) * 128u;
    run_conflict<4>("b32: 4 lanes of each half on one bank", offs);
    for (int lane = 0; lane < 64; ++lane) offs[lane] = (lane < 4) ? unsigned(lane) * 256u : 15u * 256u + unsigned(lane & 31) * 4u + unsigned(lane >> 5) * 128u;
    run_conflict<4>("b32: 4 lanes of the first half on one bank", offs);
    for (int lane = 0; lane < 64; ++lane) offs[lane] = unsigned(rand() % 16) * 256u + unsigned(rand() % 64) * 4u;
    run_conflict<4>("b32 random banks", offs);

    // ---- part 2 ----
    // tokens: byte address of a random record's ninth entry (48-byte records, 1536 per workgroup), as the planes hold them
    unsigned *tok = (unsigned *)malloc(2 * 768 * 8 * 4);
    for (int i = 0; i < 2 * 768 * 8; ++i) {
        const unsigned idx = unsigned(rand()) % 1536u;
        tok[i] = 48u * idx + ((idx >> 1) & 12u);
    }
    CHECK(hipMalloc(&g_tok, 2 * 768 * 8 * 4));
    CHECK(hipMemcpy(g_tok, tok, 2 * 768 * 8 * 4, hipMemcpyHostToDevice));
    run_phase<0, 1, false>("gathers only (random records)");
    run_phase<0, 1, true>("gathers only + barrier");
    run_phase<50, 0, false>("VALU only (pass-2 sized)");
    run_phase<50, 1, false>("gathers + 50 VALU");
    run_phase<50, 1, true>("gathers + 50 VALU + barrier");
    run_phase<155, 0, false>("VALU only (pass-3 sized)");
    run_phase<155, 0, true>("VALU only (pass-3 sized) + barrier");
    run_phase<155, 1, false>("gathers + 155 VALU");
    run_phase<155, 1, true>("gathers + 155 VALU + barrier");
    // conflict-free gathers: every lane reads its own record (consecutive 48-byte records)
    for (int b = 0; b < 2; ++b)
        for (int t = 0; t < 768; ++t)
            for (int p = 0; p < 2; ++p)
                for (int k = 0; k < 4; ++k) {
                    const unsigned idx = unsigned(p * 768 + t);
                    tok[b * 768 * 8 + (t * 2 + p) * 4 + k] = 48u * idx + ((idx >> 1) & 12u);
                }
    CHECK(hipMemcpy(g_tok, tok, 2 * 768 * 8 * 4, hipMemcpyHostToDevice));
    run_phase<0, 1, false>("gathers only (own record: conflict-free)");
    run_phase<50, 1, true>("conflict-free gathers + 50 VALU + barrier");
    run_phase<155, 1, true>("conflict-free gathers + 155 VALU + barrier");
    return 0;
}
