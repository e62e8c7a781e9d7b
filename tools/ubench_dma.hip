// LDS-DMA (global_load_lds_dwordx4) as the tile kernel's stream: what does it buy on gfx950?  
//
// The tile kernel's stage A pulls ~80 KB per tile through 8-byte loads into VGPRs; a CU ingests ~11 B/cycle that way
// (tools/ubench_ingest.hip).  A wave-specialised resident workgroup would instead have loader waves DMA the NEXT tile's planes
// into an LDS ring while consumer waves run passes 2-3 (LDS gathers + 3x3 algebra) on the current tile.  Three questions,
// priced here before tile_body is touched:
//   1. ingest rate of a workgroup whose whole stream is LDS-DMA (16 B per lane, 1 KiB per wave-instruction), one / two / many
//      workgroups per CU, against the VGPR loads of ubench_ingest.hip;
//   2. a loader/consumer workgroup (1 024 threads = the largest a CU takes: 4 loader + 12 consumer waves, 72 KB of records +
//      80 KB ring): the loaders' rate with the consumers idle and busy, the consumers' phase time with the loaders idle and busy
//      -- do the DMA writes into LDS and the gathers out of it overlap, or do they queue on the one LDS pipe?
//   3. the same consumer phase at the production occupancy (2 x 12 waves per CU, no loaders) for the per-slot reference.
//
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench_dma.hip -o tools/_bin/ubench_dma && tools/_bin/ubench_dma
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

typedef float v4f __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// one wave-instruction: 64 lanes x 16 B from gsrc (per lane) to LDS bytes [lds_base, lds_base + 1024)
__device__ __forceinline__ void dma16(const unsigned char *gsrc, unsigned lds_base)
{
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)gsrc, (LDS_AS void *)(uintptr_t)lds_base, 16, 0, 0);
}

// ---- 1. a workgroup's whole stream by LDS-DMA: 768 threads, KB_PER_WG KiB, all issued back to back, one wait ----
template <int PIECES_PER_WAVE>   // 1 KiB pieces per wave: 12 waves x 7 = 84 KiB
__global__ __launch_bounds__(768) void ingest_dma(const unsigned char *base, size_t stride_bytes, long long *clk, unsigned *sink)
{
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char *blk = base + size_t(blockIdx.x) * stride_bytes;
    const long long t0 = clock64();
#pragma unroll
    for (int k = 0; k < PIECES_PER_WAVE; ++k) {
        const unsigned piece = unsigned(k * 12 + wave);   // plane-like: piece p of the block, waves interleaved
        dma16(blk + size_t(piece) * 1024 + size_t(lane) * 16, piece * 1024u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    if (lane == 0) {
        clk[(size_t(blockIdx.x) * 12 + wave) * 2] = t0;
        clk[(size_t(blockIdx.x) * 12 + wave) * 2 + 1] = t1;
    }
    __syncthreads();
    if (smem[threadIdx.x * 16] == 0x5a && smem[4097] == 0x11) sink[0] = 1;
}

// the same bytes through VGPRs (16-byte loads), for a same-binary reference
template <int PIECES_PER_WAVE>
__global__ __launch_bounds__(768) void ingest_vgpr(const unsigned char *base, size_t stride_bytes, long long *clk, unsigned *sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char *blk = base + size_t(blockIdx.x) * stride_bytes;
    const long long t0 = clock64();
    v4f r[PIECES_PER_WAVE];
#pragma unroll
    for (int k = 0; k < PIECES_PER_WAVE; ++k) r[k] = *reinterpret_cast<const v4f *>(blk + size_t(k * 12 + wave) * 1024 + size_t(lane) * 16);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < PIECES_PER_WAVE; ++k) acc += r[k].x + r[k].w;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc));
    const long long t1 = clock64();
    if (lane == 0) {
        clk[(size_t(blockIdx.x) * 12 + wave) * 2] = t0;
        clk[(size_t(blockIdx.x) * 12 + wave) * 2 + 1] = t1;
    }
    if (acc == 123.456f) sink[0] = 1;
}

// ---- 0. where in a 160 KiB LDS can an LDS-DMA land?  (one wave copies 1 KiB of a known pattern to `base`, reads it back with ds_read) ----
__global__ __launch_bounds__(64) void dma_where(const unsigned *src, unsigned base, unsigned *mismatches)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x;
    for (unsigned i = lane; i < 163840u / 4u; i += 64) ((LDS_AS unsigned *)(uintptr_t)0)[i] = 0xdeadbeefu;
    __syncthreads();
    dma16(reinterpret_cast<const unsigned char *>(src) + size_t(lane) * 16, base);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned bad = 0;
    for (int j = 0; j < 4; ++j) bad += ((LDS_AS unsigned *)(uintptr_t)(base + 16u * unsigned(lane)))[j] != src[4 * lane + j];
    // where did it go instead?  (first dword of the pattern)
    unsigned found = 0xffffffffu;
    for (unsigned i = lane; i < 163840u / 4u; i += 64)
        if (((LDS_AS unsigned *)(uintptr_t)0)[i] == src[0] && 4u * i < found) found = 4u * i;
    atomicAdd(mismatches, bad);
    if (found != 0xffffffffu) atomicMin(mismatches + 1, found);
}

// ---- 2. loader / consumer workgroup ----
// LDS map: records [0, 72 KB) (1 536 records of 48 B), ring [72 KB, 152 KB).
constexpr unsigned kRecBytes = 1536u * 48u, kRingBytes = 80u * 1024u;
constexpr int kLoaderWaves = 4;

// one pass-3-shaped phase per slot: 4 random record gathers (2 x b128 + 1 x b32 each), wait, V dependent VALU, one 48-byte store
template <int V>
__device__ __forceinline__ void consumer_phase(const unsigned (&nb)[2][4], unsigned own[2], float (&acc)[9])
{
    const float b = 1.0001f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        v4f q[8];
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(q[2 * k]) : "v"(nb[p][k] & ~15u));
            asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(q[2 * k + 1]) : "v"(nb[p][k] & ~15u));
            asm volatile("ds_read_b32 %0, %1" : "=v"(t[k]) : "v"(nb[p][k]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[0] -= q[2 * k].x; acc[1] -= q[2 * k].y; acc[2] -= q[2 * k].z; acc[3] -= q[2 * k].w;
            acc[4] -= q[2 * k + 1].x; acc[5] -= q[2 * k + 1].y; acc[6] -= q[2 * k + 1].z; acc[7] -= q[2 * k + 1].w;
            acc[8] -= t[k];
        }
#pragma unroll
        for (int j = 0; j < V - 36; ++j) acc[j % 9] = __builtin_fmaf(acc[j % 9], b, acc[(j + 4) % 9]);
        const v4f s0 = {acc[0], acc[1], acc[2], acc[3]}, s1 = {acc[4], acc[5], acc[6], acc[7]};
        asm volatile("ds_write_b128 %0, %1 offset:16" : : "v"(own[p]), "v"(s0) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:32" : : "v"(own[p]), "v"(s1) : "memory");
    }
}

// mode bit 0: loaders run, bit 1: consumers run.  Loaders: `tiles` x 80 KiB from fresh memory, each loader wave keeps <= 20 pieces
// in flight (one tile's share), waits for them, goes on.  Consumers: `phases` phases each (no barrier: pricing, not a pipeline).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void ring_probe(const unsigned char *base, size_t stride_bytes, const unsigned *tok, int tiles, int phases,
                                                      int mode, long long *clk, float *out)
{
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool loader = THREADS == 1024 && wave < kLoaderWaves;
    const long long t0 = clock64();
    if (loader) {
        if (mode & 1) {
            const unsigned char *blk = base + size_t(blockIdx.x) * stride_bytes;
            for (int t = 0; t < tiles; ++t) {
#pragma unroll
                for (int k = 0; k < 20; ++k) {
                    const unsigned piece = unsigned(k * kLoaderWaves + wave);
                    dma16(blk + (size_t(t) * 80 + piece) * 1024 + size_t(lane) * 16, kRecBytes + piece * 1024u);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    } else if (mode & 2) {
        const int ct = threadIdx.x - (THREADS == 1024 ? kLoaderWaves * 64 : 0);
        unsigned nb[2][4], own[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            own[p] = unsigned(p * 768 + ct) * 48u;
#pragma unroll
            for (int k = 0; k < 4; ++k) nb[p][k] = tok[(ct * 2 + p) * 4 + k];
        }
        float acc[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] = float(j) + ct;
        for (int it = 0; it < phases; ++it) consumer_phase<155>(nb, own, acc);
        if (acc[0] == 123.456f) out[0] = acc[0] + smem[0];
    }
    const long long t1 = clock64();
    if (lane == 0) {
        clk[(size_t(blockIdx.x) * 16 + wave) * 2] = t0;
        clk[(size_t(blockIdx.x) * 16 + wave) * 2 + 1] = t1;
    }
}

static unsigned char *g_buf = nullptr;
static size_t g_bytes = 0, g_cursor = 0;
static long long *g_clk = nullptr;
static unsigned *g_sink = nullptr, *g_tok = nullptr;
static float *g_out = nullptr;
static int g_cus = 256;

static const unsigned char *fresh(size_t bytes)
{
    if (g_cursor + bytes > g_bytes) g_cursor = 0;
    const unsigned char *p = g_buf + g_cursor;
    g_cursor += (bytes + 4095) & ~size_t(4095);
    return p;
}

template <class K>
static void run_ingest(const char *what, K kern, int blocks, size_t block_bytes, size_t lds)
{
    const size_t stride = (block_bytes + 4095) & ~size_t(4095);
    const unsigned char *base = fresh(size_t(blocks) * stride);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(768), lds, 0, base, stride, g_clk, g_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> clk(size_t(blocks) * 24);
    CHECK(hipMemcpy(clk.data(), g_clk, clk.size() * 8, hipMemcpyDeviceToHost));
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
    const double med = dur[dur.size() / 2];
    printf("%-58s %4d WGs x %5.1f KB: %7.0f clk/WG (p10 %6.0f p90 %6.0f) = %5.1f B/clk/WG;  kernel %.4f ms = %5.2f TB/s\n", what, blocks,
           block_bytes / 1024.0, med, dur[dur.size() / 10], dur[dur.size() * 9 / 10], block_bytes / med, ms,
           double(blocks) * block_bytes / (ms * 1e-3) / 1e12);
}

template <int THREADS>
static void run_ring(const char *what, int wgs_per_cu, int tiles, int phases, int mode)
{
    const int blocks = g_cus * wgs_per_cu;
    const size_t per_wg = size_t(tiles) * 80 * 1024, stride = per_wg + 4096;
    const unsigned char *base = fresh(size_t(blocks) * stride);
    auto k = ring_probe<THREADS>;
    const size_t lds = THREADS == 1024 ? kRecBytes + kRingBytes : kRecBytes + 6144;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), lds, 0, base, stride, g_tok, tiles, phases, mode, g_clk, g_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> clk(size_t(blocks) * 32);
    CHECK(hipMemcpy(clk.data(), g_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    // median over workgroups of the loaders' and the consumers' spans
    std::vector<double> dl, dc;
    const int nw = THREADS / 64, nl = THREADS == 1024 ? kLoaderWaves : 0;
    for (int b = 0; b < blocks; ++b) {
        double l = 0, c = 0;
        for (int w = 0; w < nw; ++w) {
            const double d = double(clk[(size_t(b) * 16 + w) * 2 + 1] - clk[(size_t(b) * 16 + w) * 2]);
            if (w < nl) l = std::max(l, d); else c = std::max(c, d);
        }
        dl.push_back(l);
        dc.push_back(c);
    }
    std::sort(dl.begin(), dl.end());
    std::sort(dc.begin(), dc.end());
    const double ml = dl[dl.size() / 2], mc = dc[dc.size() / 2];
    printf("%-64s kernel %.4f ms", what, ms);
    if (mode & 1) printf(" | loaders %8.0f clk = %5.1f B/clk/WG, chip %5.2f TB/s", ml, per_wg / ml, double(blocks) * per_wg / (ms * 1e-3) / 1e12);
    if (mode & 2) printf(" | consumers %8.0f clk = %6.1f clk per phase of 1 536 slots (%4.2f clk/slot/WG)", mc, mc / phases, mc / phases / 1536.0);
    printf("\n");
}

int main()
{
    g_bytes = size_t(8) << 30;
    CHECK(hipMalloc(&g_buf, g_bytes));
    CHECK(hipMemset(g_buf, 1, g_bytes));
    CHECK(hipMalloc(&g_clk, size_t(8192) * 32 * 8));
    CHECK(hipMalloc(&g_sink, 4));
    CHECK(hipMalloc(&g_out, 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    // random record tokens: byte address of a record's tail dword (record base = token & ~15) inside the 1 536 records
    std::vector<unsigned> tok(768 * 8);
    unsigned s = 12345u;
    for (auto &t : tok) {
        s = s * 1664525u + 1013904223u;
        const unsigned rec = (s >> 8) % 1536u;
        t = rec * 48u + ((rec >> 1) & 12u);
    }
    CHECK(hipMalloc(&g_tok, tok.size() * 4));
    CHECK(hipMemcpy(g_tok, tok.data(), tok.size() * 4, hipMemcpyHostToDevice));
    {   // where can an LDS-DMA land?
        std::vector<unsigned> pat(256);
        for (int i = 0; i < 256; ++i) pat[i] = 0x51000000u + unsigned(i);
        unsigned *d_pat, *d_mm;
        CHECK(hipMalloc(&d_pat, 1024));
        CHECK(hipMalloc(&d_mm, 8));
        CHECK(hipMemcpy(d_pat, pat.data(), 1024, hipMemcpyHostToDevice));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(dma_where), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        for (unsigned base : {0u, 32768u, 65536u - 1024u, 65536u, 98304u, 131072u - 1024u, 131072u, 140000u & ~15u, 162816u}) {
            unsigned init[2] = {0u, 0xffffffffu}, mm[2];
            CHECK(hipMemcpy(d_mm, init, 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(dma_where, dim3(1), dim3(64), 163840, 0, d_pat, base, d_mm);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(mm, d_mm, 8, hipMemcpyDeviceToHost));
            printf("LDS-DMA to LDS byte %6u: %3u of 256 dwords wrong; the pattern's first dword sits at LDS byte %u\n", base, mm[0], mm[1]);
        }
    }
    printf("clock64 ticks at the shader clock's constant reference (100 MHz class counters report fewer 'cycles' -- compare lines, and kernel ms)\n");
    auto kd = ingest_dma<7>;
    auto kv = ingest_vgpr<7>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        printf("---- pass %d: whole-stream ingest, 84 KiB per workgroup of 768 threads ----\n", rep);
        run_ingest("VGPR 16 B loads, one WG on 1/4 of the CUs", kv, g_cus / 4, 84 * 1024, 0);
        run_ingest("LDS-DMA 16 B,    one WG on 1/4 of the CUs", kd, g_cus / 4, 84 * 1024, 84 * 1024);
        run_ingest("VGPR 16 B loads, one WG per CU", kv, g_cus, 84 * 1024, 0);
        run_ingest("LDS-DMA 16 B,    one WG per CU", kd, g_cus, 84 * 1024, 84 * 1024);
        run_ingest("VGPR 16 B loads, two WGs per CU (LDS-limited like the tile kernel)", kv, 2 * g_cus, 84 * 1024, 80 * 1024);
        run_ingest("LDS-DMA 16 B,    two WGs per CU (LDS-limited: 1 resident per 84 KiB)", kd, 2 * g_cus, 84 * 1024, 84 * 1024);
        run_ingest("VGPR 16 B loads, 16 WGs per CU (stream)", kv, 16 * g_cus, 84 * 1024, 80 * 1024);
        run_ingest("LDS-DMA 16 B,    16 WGs per CU (stream, 1 resident per CU)", kd, 16 * g_cus, 84 * 1024, 84 * 1024);
    }
    for (int rep = 0; rep < 2; ++rep) {
        printf("---- pass %d: loader / consumer workgroup (4 + 12 waves, 152 KiB LDS, one per CU); 40 tiles of 80 KiB, 40 phases ----\n", rep);
        run_ring<1024>("loaders alone", 1, 40, 40, 1);
        run_ring<1024>("consumers alone (12 waves per CU)", 1, 40, 40, 2);
        run_ring<1024>("loaders + consumers", 1, 40, 40, 3);
        run_ring<1024>("loaders + consumers, 2x the consumer work", 1, 40, 80, 3);
        run_ring<768>("reference: 2 x 12 consumer waves per CU, no loaders (production occupancy)", 2, 40, 40, 2);
        run_ring<768>("reference: 1 x 12 consumer waves per CU", 1, 40, 40, 2);
    }
    return 0;
}
