// How fast can ONE workgroup of a gfx950 CU pull a cold, contiguous block out of HBM?
//
// The tile kernel's memory phase moves ~98 KB per tile (13 planes of 8 bytes per lane + lists + positions) and takes
// ~9-11 k cycles = 9-11 bytes per cycle per CU, while the chip as a whole sits at 62 % of the bandwidth a pure stream
// reaches (profiles/r04_pipeline_model.md).  Is that per-CU rate a property of the CU's memory pipeline (requests in
// flight / latency), of the request width, or of the HBM?  This probe launches W workgroups of 768 threads (1 or 2 per
// CU, or on a fraction of the CUs), each reading its own contiguous block once -- N loads of WIDTH bytes per lane, all
// issued back to back, then one wait -- and reports bytes per cycle per workgroup from shader-clock stamps around the
// loads, for 8- and 16-byte loads, plane-like (N strided streams) and flat layouts.
//
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench_ingest.hip -o tools/_bin/ubench_ingest && tools/_bin/ubench_ingest
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                           \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) {                                            \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                       \
        }                                                                  \
    } while (0)

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// Every workgroup reads block_bytes = 768 * N * WIDTH bytes starting at base + block * stride_bytes.
// PLANES: load k of a lane sits at k * (768 * WIDTH) + lane * WIDTH (N streams, like the tile blob); otherwise at
// lane * (N * WIDTH) + k * WIDTH (each lane owns N * WIDTH consecutive bytes).
template <int WIDTH, int N, bool PLANES>
__global__ __launch_bounds__(768) void ingest(const unsigned char *base, size_t stride_bytes, long long *clk, unsigned *sink, int dependent_first)
{
    const int tid = threadIdx.x;
    const unsigned char *blk = base + size_t(blockIdx.x) * stride_bytes;
    long long t0 = clock64();
    unsigned acc = 0;
    if (dependent_first) {   // a dependent load ahead of the block, like the tile descriptor
        const unsigned off = *reinterpret_cast<const unsigned *>(blk + 64 * (tid >> 6));
        acc = off & 1u;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc));
    }
    if (WIDTH == 8) {
        v2u r[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
            r[k] = *reinterpret_cast<const v2u *>(blk + (PLANES ? size_t(k) * 768 * 8 + size_t(tid) * 8 : size_t(tid) * N * 8 + size_t(k) * 8));
#pragma unroll
        for (int k = 0; k < N; ++k) acc += r[k].x ^ r[k].y;
    } else {
        v4u r[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
            r[k] = *reinterpret_cast<const v4u *>(blk + (PLANES ? size_t(k) * 768 * 16 + size_t(tid) * 16 : size_t(tid) * N * 16 + size_t(k) * 16));
#pragma unroll
        for (int k = 0; k < N; ++k) acc += r[k].x ^ r[k].y ^ r[k].z ^ r[k].w;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc));
    long long t1 = clock64();
    if ((tid & 63) == 0) {
        clk[(size_t(blockIdx.x) * 12 + (tid >> 6)) * 2] = t0;
        clk[(size_t(blockIdx.x) * 12 + (tid >> 6)) * 2 + 1] = t1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static unsigned char *g_buf = nullptr;
static size_t g_bytes = 0;
static long long *g_clk = nullptr;
static unsigned *g_sink = nullptr;
static size_t g_cursor = 0;

template <int WIDTH, int N, bool PLANES>
static void run(const char *what, int blocks, int dependent_first = 0)
{
    const size_t block_bytes = size_t(768) * N * WIDTH;
    const size_t stride = (block_bytes + 4095) & ~size_t(4095);
    // a fresh region every launch: nothing of it is in L2 / MALL (the buffer is 6 GB, MALL 256 MB)
    if (g_cursor + size_t(blocks) * stride > g_bytes) g_cursor = 0;
    const unsigned char *base = g_buf + g_cursor;
    g_cursor += size_t(blocks) * stride;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((ingest<WIDTH, N, PLANES>), dim3(blocks), dim3(768), 0, 0, base, stride, g_clk, g_sink, dependent_first);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> clk(size_t(blocks) * 24);
    CHECK(hipMemcpy(clk.data(), g_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    // per workgroup: first wave's start -> last wave's data landed
    std::vector<double> dur;
    for (int b = 0; b < blocks; ++b) {
        long long s = clk[size_t(b) * 24], e = clk[size_t(b) * 24 + 1];
        for (int w = 1; w < 12; ++w) {
            s = std::min(s, clk[(size_t(b) * 12 + w) * 2]);
            e = std::max(e, clk[(size_t(b) * 12 + w) * 2 + 1]);
        }
        dur.push_back(double(e - s));
    }
    std::sort(dur.begin(), dur.end());
    const double med = dur[dur.size() / 2], p10 = dur[dur.size() / 10], p90 = dur[dur.size() * 9 / 10];
    printf("%-52s %4d WGs x %6.1f KB: %8.0f cycles/WG (p10 %6.0f, p90 %6.0f) = %5.1f B/cycle/WG;  kernel %.4f ms = %5.2f TB/s\n", what, blocks,
           block_bytes / 1024.0, med, p10, p90, block_bytes / med, ms, double(blocks) * block_bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    g_bytes = size_t(6) << 30;
    CHECK(hipMalloc(&g_buf, g_bytes));
    CHECK(hipMemset(g_buf, 1, g_bytes));
    CHECK(hipMalloc(&g_clk, size_t(4096) * 24 * 8));
    CHECK(hipMalloc(&g_sink, 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        printf("---- pass %d ----\n", rep);
        // the tile kernel's shape: 13 planes x 8 bytes per lane = 78 KB per workgroup
        run<8, 13, true>("13 planes x 8 B, one WG on 1/4 of the CUs", cus / 4);
        run<8, 13, true>("13 planes x 8 B, one WG per CU", cus);
        run<8, 13, true>("13 planes x 8 B, two WGs per CU", 2 * cus);
        run<8, 13, true>("13 planes x 8 B, eight WGs per CU (stream)", 8 * cus);
        run<8, 13, true>("13 planes x 8 B, one WG per CU, dependent first load", cus, 1);
        run<8, 13, false>("13 x 8 B lane-contiguous, one WG per CU", cus);
        run<16, 7, true>("7 planes x 16 B (86 KB), one WG on 1/4 of the CUs", cus / 4);
        run<16, 7, true>("7 planes x 16 B (86 KB), one WG per CU", cus);
        run<16, 7, true>("7 planes x 16 B (86 KB), two WGs per CU", 2 * cus);
        run<16, 7, false>("7 x 16 B lane-contiguous (112 B per lane), one WG per CU", cus);
        run<16, 3, true>("3 planes x 16 B (37 KB), one WG per CU", cus);
        run<8, 6, true>("6 planes x 8 B (37 KB), one WG per CU", cus);
        run<16, 14, true>("14 planes x 16 B (172 KB), one WG per CU", cus);
    }
    return 0;
}
