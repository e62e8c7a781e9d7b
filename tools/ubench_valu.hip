// Instruction-throughput probe for gfx950: how many cycles a wave64 VALU instruction of each flavour
// occupies its SIMD.  Decides whether the 3x3 algebra of the tile kernel should be written packed
// (v_pk_*_f32) and what an in-register fp64 Dm^-1 costs.  Build + run: tools/ubench.sh
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum Op { FMA32, PKFMA32, PKMUL32, PKADD32, FMA64, MUL64, ADD64, MOV32, AND32, MAD_U32_U24, LSHL_ADD, RCP32, RCP64, CVT_F64_F32 };

template <int OP>
__global__ __launch_bounds__(256) void probe(float *out, int iters, long long *clk)
{
    constexpr int N = 16;
    float a[N];
    v2f p[N];
    double d[N];
    unsigned u[N];
    const float b = 1.0001f + threadIdx.x * 1e-7f, c = 1e-9f;
    const v2f pb = {b, b}, pc = {c, c};
    const double db = b, dc = c;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        a[j] = float(j) + threadIdx.x;
        p[j] = v2f{a[j], a[j] + 1.f};
        d[j] = a[j];
        u[j] = unsigned(j) + threadIdx.x;
    }
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
            if (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pb), "v"(pc));
            if (OP == PKMUL32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pb));
            if (OP == PKADD32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pc));
            if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(db), "v"(dc));
            if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(db));
            if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(dc));
            if (OP == MOV32) asm volatile("v_mov_b32 %0, %1" : "+v"(a[j]) : "v"(b));
            if (OP == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[j]) : "v"(0xfffffu));
            if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[j]) : "v"(3u), "v"(1u));
            if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[j]) : "v"(1u));
            if (OP == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));
            if (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[j]));
            if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[j]) : "v"(a[j]));
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) s += a[j] + p[j].x + p[j].y + float(d[j]) + float(u[j]);
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x % 64 == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 2000, N = 16;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    float *out;
    long long *clk;
    CHECK(hipMalloc(&out, 4));
    CHECK(hipMalloc(&clk, sizeof(long long) * blocks * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    double mean = 0;
    for (long long v : h) mean += double(v);
    mean /= double(h.size());
    const double instr_per_wave = double(iters) * N;
    // every SIMD runs waves_per_simd waves concurrently for `mean` ticks
    const double ticks_per_instr = mean / (instr_per_wave * waves_per_simd);
    const double ns_per_instr = double(ms) * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-14s waves/SIMD %d: %6.2f clock64 ticks per wave-instr per SIMD, %6.3f ns (wall) -> %5.2f cycles at 2.4 GHz\n", name,
           waves_per_simd, ticks_per_instr, ns_per_instr, ns_per_instr * 2.4);
    CHECK(hipFree(out));
    CHECK(hipFree(clk));
}

int main()
{
    for (int w : {1, 4, 8}) {
        run<FMA32>("v_fma_f32", w);
        run<PKFMA32>("v_pk_fma_f32", w);
        run<PKMUL32>("v_pk_mul_f32", w);
        run<PKADD32>("v_pk_add_f32", w);
        run<FMA64>("v_fma_f64", w);
        run<MUL64>("v_mul_f64", w);
        run<ADD64>("v_add_f64", w);
        run<MOV32>("v_mov_b32", w);
        run<AND32>("v_and_b32", w);
        run<MAD_U32_U24>("v_mad_u32_u24", w);
        run<LSHL_ADD>("v_lshl_add_u32", w);
        run<RCP32>("v_rcp_f32", w);
        run<RCP64>("v_rcp_f64", w);
        run<CVT_F64_F32>("v_cvt_f64_f32", w);
    }
    return 0;
}
