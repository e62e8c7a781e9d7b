// Do VALU issue and LDS data movement overlap on a gfx950 CU, or do they add up?
//
// The tile kernel runs both pipes at ~50 % and everything that removed work from only one of them bought a
// fraction of the predicted time (profiles/r02_experiments.md).  This probe runs, at the tile kernel's occupancy
// (2 workgroups x 768 threads per CU = 6 waves per SIMD), loops of V independent v_fma_f32, R conflict-free
// ds_read_b128 and W ds_write_b128 per iteration -- in the same wave, and split over different waves of the same
// SIMD -- and prints wall-clock ns per iteration per CU.  max(t_V, t_LDS) = the pipes overlap; t_V + t_LDS = they
// share something (VGPR ports / issue).
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap.hip -o tools/_bin/ubench_overlap && tools/_bin/ubench_overlap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// role: 0 = every wave does V + R + W; 1 = waves with (wave / 4) even do V only, odd do R + W only (waves w and
// w + 4 share a SIMD, so every SIMD hosts both roles); 2 = as 1 but the LDS waves idle (VALU half alone);
// 3 = as 1 but the VALU waves idle (LDS half alone).
template <int V, int R, int W, int ROLE, int RDW>
__global__ __launch_bounds__(768) void probe(float *out, int iters)
{
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // each wave owns 8 KiB: lane-major 16-B (or RDW-byte) columns, 8 rows of 1 KiB
    const unsigned base = wave * 8192u + lane * 16u;
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = float(j) + threadIdx.x;
    const float b = 1.0001f, c = 1e-9f;
    v4f w4 = {1.f, 2.f, 3.f, 4.f};
    const bool do_v = ROLE == 0 || ((ROLE == 1 || ROLE == 2) && ((wave >> 2) & 1) == 0);
    const bool do_l = ROLE == 0 || ((ROLE == 1 || ROLE == 3) && ((wave >> 2) & 1) == 1);
    for (int i = 0; i < iters; ++i) {
        if (ROLE == 0) {
            constexpr int STEPS = R + W > 0 ? R + W : 1;
            constexpr int VPS = (V + STEPS - 1) / STEPS;
            int v_done = 0;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (s < R) {
                    v4f r;
                    if (RDW == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(base + (s & 7) * 1024u));
                    if (RDW == 8) asm volatile("ds_read_b64 %0, %1" : "=v"(r.xy) : "v"(base + (s & 7) * 1024u));
                    if (RDW == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(r.x) : "v"(base + (s & 7) * 1024u));
                } else if (s < R + W) {
                    if (RDW == 16) asm volatile("ds_write_b128 %0, %1" : : "v"(base + (s & 7) * 1024u), "v"(w4));
                    if (RDW == 8) asm volatile("ds_write_b64 %0, %1" : : "v"(base + (s & 7) * 1024u), "v"(w4.xy));
                    if (RDW == 4) asm volatile("ds_write_b32 %0, %1" : : "v"(base + (s & 7) * 1024u), "v"(w4.x));
                }
#pragma unroll
                for (int k = 0; k < VPS; ++k)
                    if (v_done < V) {
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[v_done & 15]) : "v"(b), "v"(c));
                        ++v_done;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else {
            if (do_v) {
#pragma unroll
                for (int k = 0; k < 2 * V; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k & 15]) : "v"(b), "v"(c));
            }
            if (do_l) {
#pragma unroll
                for (int s = 0; s < 2 * R; ++s) {
                    v4f r;
                    if (RDW == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(base + (s & 7) * 1024u));
                    if (RDW == 8) asm volatile("ds_read_b64 %0, %1" : "=v"(r.xy) : "v"(base + (s & 7) * 1024u));
                    if (RDW == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(r.x) : "v"(base + (s & 7) * 1024u));
                }
#pragma unroll
                for (int s = 0; s < 2 * W; ++s) {
                    if (RDW == 16) asm volatile("ds_write_b128 %0, %1" : : "v"(base + (s & 7) * 1024u), "v"(w4));
                    if (RDW == 8) asm volatile("ds_write_b64 %0, %1" : : "v"(base + (s & 7) * 1024u), "v"(w4.xy));
                    if (RDW == 4) asm volatile("ds_write_b32 %0, %1" : : "v"(base + (s & 7) * 1024u), "v"(w4.x));
                }
                asm volatile("s_waitcnt lgkmcnt(0)");
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a[j];
    if (s == 123.456f) out[0] = s + smem[0];
}

template <int V, int R, int W, int ROLE, int RDW = 16>
double run(const char *what)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 4000;
    const int blocks = cus * 2;
    static float *out = nullptr;
    if (!out) CHECK(hipMalloc(&out, 4));
    auto k = probe<V, R, W, ROLE, RDW>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(768), 80 * 1024, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(768), 80 * 1024, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per CU and iteration: 24 waves x (V fma + R reads + W writes)
    const double ns_iter = double(ms) * 1e6 / iters;
    printf("%-58s V=%3d R=%2d W=%2d x%2dB role %d: %8.2f ns per iteration per CU (24 waves)  = %6.1f cycles at 2.4 GHz\n", what, V, R, W, RDW,
           ROLE, ns_iter, ns_iter * 2.4);
    return ns_iter;
}

int main()
{
    // same wave: VALU only, LDS only, both
    run<48, 0, 0, 0>("VALU only");
    run<0, 8, 0, 0>("ds_read_b128 only");
    run<48, 8, 0, 0>("VALU + ds_read_b128, same waves, interleaved");
    run<0, 0, 4, 0>("ds_write_b128 only");
    run<48, 0, 4, 0>("VALU + ds_write_b128, same waves");
    run<0, 8, 4, 0>("ds_read_b128 + ds_write_b128");
    run<48, 8, 4, 0>("VALU + reads + writes, same waves");
    run<96, 8, 4, 0>("2x VALU + reads + writes, same waves");
    run<0, 16, 0, 0, 4>("ds_read_b32 only");
    run<48, 16, 0, 0, 4>("VALU + ds_read_b32");
    run<0, 16, 0, 0, 8>("ds_read_b64 only");
    run<0, 0, 8, 0, 4>("ds_write_b32 only");
    run<48, 0, 8, 0, 4>("VALU + ds_write_b32");
    // split roles: half the waves of each SIMD do 2V, the other half 2R / 2W (same totals per SIMD)
    run<48, 8, 0, 2>("split: VALU half alone");
    run<48, 8, 0, 3>("split: read half alone");
    run<48, 8, 0, 1>("split: VALU waves + read waves");
    run<48, 0, 4, 3>("split: write half alone");
    run<48, 0, 4, 1>("split: VALU waves + write waves");
    run<48, 8, 4, 1>("split: VALU waves + read/write waves");
    return 0;
}
