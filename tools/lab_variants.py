#!/usr/bin/env python3
"""Laboratory builds of the tile kernel WITHOUT touching the product source: every variant is a list of textual patches applied
to a COPY of tssplat_amd/csrc/kernels.hip (and, optionally, plan.cpp), compiled into tssplat_amd/libtssplat_amd_<name>.so next to
the product library and selected per process with TSSPLAT_AMD_LIB (tools/ab_variants.py does that).  The shipped kernels carry no
experiment switches; pricing builds ("what does this phase cost?": results WRONG on purpose) and candidate
changes live here, reproducible from this file.

    python tools/lab_variants.py --list
    python tools/lab_variants.py nogather nostore ...      # build
    python tools/ab_variants.py base nogather nostore --spheres 512
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tssplat_amd import _build  # noqa: E402

K = "kernels.hip"
P = "plan.cpp"

# name -> (description, [(file, old, new), ...])
VARIANTS = {
    # ---- pricing builds: one stage removed, results wrong, cost right ----
    "exit_p1": ("leave after pass 1 (stream + F): what the memory phase costs", [
        (K, "    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----",
            "    if (e_b == 12345.678f) g_partials[0] = e_b;\n    if (WITH_GRAD) return;\n    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----")]),
    "noscatter": ("no scatter of the corner forces (table reads and address arithmetic stay)", [
        (K, "                if (p * nq + tid < td.n_slots) {   // (padding slots", "                if (p * nq + tid < td.n_slots && D[p][0] == 12345.678f) {   // (padding slots")]),
    "nogather": ("per-vertex sums skipped (stores of zeros stay)", [
        (K, "            for (int r = 0; r < rows; r += 4) {\n                const LDS_AS float *f[4];", "            for (int r = 0; r < rows && gscale == 12345.678f; r += 4) {\n                const LDS_AS float *f[4];")]),
    "nostore": ("result stores skipped", [
        (K, "            if (v < td.n_verts) {\n                const bool excl = row >= 0;", "            if (v < td.n_verts && gx == 12345.678f) {\n                const bool excl = row >= 0;")]),
    "notail": ("no scatter, no sums, no stores: everything behind pass 3", [
        (K, "                if (p * nq + tid < td.n_slots) {   // (padding slots", "                if (p * nq + tid < td.n_slots && D[p][0] == 12345.678f) {   // (padding slots"),
        (K, "            for (int r = 0; r < rows; r += 4) {\n                const LDS_AS float *f[4];", "            for (int r = 0; r < rows && gscale == 12345.678f; r += 4) {\n                const LDS_AS float *f[4];"),
        (K, "            if (v < td.n_verts) {\n                const bool excl = row >= 0;", "            if (v < td.n_verts && gx == 12345.678f) {\n                const bool excl = row >= 0;")]),
    "nopen3": ("pricing: the inverted-tet branch of pass 3 (F rebuilt, cofactor) skipped -- what moving it out of stage B could buy at most", [
        (K, "                if (scal[p] != 0.f) {  // inverted owned tet: rebuild F", "                if (scal[p] == 12345.678f) {  // inverted owned tet: rebuild F")]),
    # ---- round 6: what could ANY design that hides the stream reach?  (profiles/r06_experiments.md) ----
    "nostream": ("pricing: every tile reads tile 0's planes (L2 hits: the stream costs almost nothing, results wrong) at the production occupancy -- "
                 "the floor of a PERFECT next-tile prefetch with two workgroups per CU", [
        (K, "reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + td.blob_off);", "reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + (td.blob_off & 0));")]),
    "nostream32": ("pricing: tile t reads the planes of tile t mod 32 (2.6 MB: L2 / MALL hits spread over all channels -- `nostream` "
                   "hammers ONE tile's 80 KB from every CU and is slower than the real stream) at the production occupancy", [
        (K, "reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + td.blob_off);", "reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + a.tiles[tile & 31].blob_off);")]),
    "onewg_nostream32": ("pricing: one workgroup per CU AND the planes from L2 (tile t mod 32) -- the floor of the LDS-DMA loader / consumer design", [
        (K, "reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + td.blob_off);", "reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + a.tiles[tile & 31].blob_off);"),
        (K, "        r.tile_lds = size_t(e.lds_bytes);", "        r.tile_lds = size_t(100 * 1024);"),
        (K, "    if (lds_bytes <= configured[dev]) return hipSuccess;", "    lds_bytes = 100 * 1024;\n    if (lds_bytes <= configured[dev]) return hipSuccess;")]),
    "onewg_exit_p1": ("pricing: one workgroup per CU, leave after pass 1 -- stage A of a LONE workgroup (stream + F); `onewg` minus this = the chain of passes a "
                      "lone 12-wave workgroup needs behind it, i.e. what a loader / consumer workgroup's consumers would still have to do per tile", [
        (K, "    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----",
            "    if (e_b == 12345.678f) g_partials[0] = e_b;\n    if (WITH_GRAD) return;\n    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----"),
        (K, "        r.tile_lds = size_t(e.lds_bytes);", "        r.tile_lds = size_t(100 * 1024);"),
        (K, "    if (lds_bytes <= configured[dev]) return hipSuccess;", "    lds_bytes = 100 * 1024;\n    if (lds_bytes <= configured[dev]) return hipSuccess;")]),
    # ---- round 6: a tile FED BY LDS-DMA, built and measured.  One 1 024-thread workgroup per CU (160 KiB) runs TWO
    #      consecutive tiles: waves 0-11 evaluate tile A as the product does (planes from HBM into registers) while waves 12-15 -- the loaders --
    #      fetch tile B's vertex ids and positions and pull its eleven non-neighbour planes into an LDS ring behind tile A's map with
    #      global_load_lds (16 B per lane, no VGPRs); tile B then runs with its planes and positions read from LDS.  Results are the
    #      product's (same arithmetic).  t(pair) - t(tile A alone, `onewg`) = what a DMA-fed tile costs a lone 12-wave workgroup. ----
    "dma2": ("built: two tiles per 1 024-thread workgroup, the second one fed by LDS-DMA loader waves (see the comment above)", [
        (K, """template <bool WITH_GRAD, bool WEIGHTED, bool REBUILD, int SPT>
__device__ __forceinline__ void tile_body(const KernelArgs &a, const int tile, int32_t gv0)
{""", """constexpr uint32_t kRing = 81920u;                                // LDS ring of the DMA-fed tile: planes 0-1 and 4-12 of the device image ...
constexpr uint32_t kRingPos = kRing + 11u * 1536u * 4u + 1024u;    // ... and its staged positions (float4 per vertex)
template <bool WITH_GRAD, bool WEIGHTED, bool REBUILD, int SPT, int FEED = 0>
__device__ __forceinline__ void tile_body(const KernelArgs &a, const int tile, int32_t gv0)
{"""),
        (K, "    const int tid = threadIdx.x, nthr = blockDim.x;", "    const int tid = threadIdx.x, nthr = blockDim.x > 768 ? 768 : blockDim.x;   // (the loader waves are not the tile's)"),
        (K, """    auto pair_u = [&](int q, VU &lo, VU &hi) {
        if (kPaired) {""", """    auto pair_u = [&](int q, VU &lo, VU &hi) {
        if (FEED == 1 && q == 0) {   // the vertex planes out of the ring
            const v4u t = *lds_at<const v4u>(kRing + 16u * uint32_t(lt));
            lo[0] = t.x, lo[1] = t.y, hi[0] = t.z, hi[1] = t.w;
        } else if (kPaired) {"""),
        (K, """    auto pair_f = [&](int q, VF &lo, VF &hi) {
        if (kPaired) {""", """    auto pair_f = [&](int q, VF &lo, VF &hi) {
        if (FEED == 1 && q < 12) {   // Dm^-1 out of the ring (planes 4 ... sit behind planes 0-1: two planes further down)
            const v4f t = *lds_at<const v4f>(kRing + 4u * uint32_t(q - 2) * uint32_t(td.s_pad) + 16u * uint32_t(lt));
            lo[0] = t.x, lo[1] = t.y, hi[0] = t.z, hi[1] = t.w;
        } else if (kPaired) {"""),
        (K, """        const size_t gv = size_t(gv0) * 3;
        px = g_x[gv], py = g_x[gv + 1], pz = g_x[gv + 2];""", """        if (FEED == 1) {   // staged by the loader waves behind the ring
            const v4f p4 = *lds_at<const v4f>(kRingPos + 16u * uint32_t(tid < td.n_verts ? tid : 0));
            px = p4.x, py = p4.y, pz = p4.z;
        } else {
            const size_t gv = size_t(gv0) * 3;
            px = g_x[gv], py = g_x[gv + 1], pz = g_x[gv + 2];
        }"""),
        (K, "        dm[8] = plane_f(12);", """        if (FEED == 1) {
            const v2f t = *lds_at<const v2f>(kRing + 40u * uint32_t(td.s_pad) + 8u * uint32_t(lt));
            dm[8][0] = t.x, dm[8][1] = t.y;
        } else {
            dm[8] = plane_f(12);
        }"""),
        (K, """struct FinishArgs {""", """// one wave-instruction of LDS-DMA: 64 lanes x 16 bytes from gsrc (per lane) to LDS bytes [lds_base, lds_base + 1024)
__device__ __forceinline__ void dma16(const GLOBAL_AS unsigned char *gsrc, uint32_t lds_base)
{
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)gsrc, (LDS_AS void *)(uintptr_t)lds_base, 16, 0, 0);
}

__global__ __launch_bounds__(1024, 4) void tile_energy_dma2_kernel(const KernelArgs a)
{
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile_end = (xcd + 1) * a.tiles_per_xcd < a.n_tiles ? (xcd + 1) * a.tiles_per_xcd : a.n_tiles;
    const int tile_a = xcd * a.tiles_per_xcd + 2 * jb, tile_b = tile_a + 1;
    if (2 * jb >= a.tiles_per_xcd || tile_a >= tile_end) return;
    const bool have_b = 2 * jb + 1 < a.tiles_per_xcd && tile_b < tile_end;
    const int tid = threadIdx.x;
    if (tid < 768) {   // ---- the twelve waves that evaluate: tile A from HBM, tile B from the ring ----
        const int32_t gv0 = __builtin_nontemporal_load(&as_global(a.gvid)[size_t(tile_a) * size_t(a.vert_stride) + size_t(tid < a.vert_stride ? tid : 0)]);
        __builtin_amdgcn_sched_barrier(0);
        tile_body<true, false, false, 2, 0>(a, tile_a, gv0);
        __syncthreads();   // tile A is done with its LDS; the loaders arrive here with tile B's planes landed
        __syncthreads();   // (one barrier MORE between the loaders' wait and the first read of the ring: the guide's rule for two wave groups)
        if (have_b) tile_body<true, false, false, 2, 1>(a, tile_b, 0);
    } else {           // ---- the four loader waves ----
        const int ll = tid - 768, lw = ll >> 6, lane = ll & 63;
        if (have_b) {
            const TileDesc tb = a.tiles[tile_b];
            const auto g_gvid = as_global(a.gvid);
            const auto g_x = as_global(a.x);
            // ids -> positions of tile B's vertices (two per lane), staged behind the ring
            const size_t vb = size_t(tile_b) * size_t(a.vert_stride);
            const int v0 = ll, v1 = ll + 256;
            const int32_t g0 = g_gvid[vb + size_t(v0 < a.vert_stride ? v0 : 0)], g1 = g_gvid[vb + size_t(v1 < a.vert_stride ? v1 : 0)];
            const float p0x = g_x[size_t(g0) * 3], p0y = g_x[size_t(g0) * 3 + 1], p0z = g_x[size_t(g0) * 3 + 2];
            const float p1x = g_x[size_t(g1) * 3], p1y = g_x[size_t(g1) * 3 + 1], p1z = g_x[size_t(g1) * 3 + 2];
            if (v0 < tb.n_verts) *lds_at<v4f>(kRingPos + 16u * uint32_t(v0)) = v4f{p0x, p0y, p0z, 0.f};
            if (v1 < tb.n_verts) *lds_at<v4f>(kRingPos + 16u * uint32_t(v1)) = v4f{p1x, p1y, p1z, 0.f};
            // ... behind tile A's second barrier: its consumers have their own planes by then, the CU's ingest is free, and the DMA runs
            // beside tile A's passes 2-3 / scatter / sums (issued at the head of tile A instead, it shares the ingest with tile A's
            // own stream: 0.5975 ms against 0.5662 for one workgroup per CU without any DMA)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            // the planes: device-image bytes of planes 0-1 and of planes 4-12, 1 KiB per wave-instruction, no VGPRs
            const GLOBAL_AS unsigned char *src = as_global(a.blob) + tb.blob_off;
            const uint32_t b01 = 8u * uint32_t(tb.s_pad), b412 = 36u * uint32_t(tb.s_pad);
            // (s_pad is a multiple of 4, not of 256: the last piece of a range is cut to the range by the lane mask -- a lane that
            // is off writes nothing)
            for (uint32_t k = uint32_t(lw); k * 1024u < b01; k += 4u) {
                const uint32_t piece = uint32_t(__builtin_amdgcn_readfirstlane(int(k)));
                if (piece * 1024u + uint32_t(lane) * 16u < b01) dma16(src + size_t(piece) * 1024 + size_t(lane) * 16, kRing + piece * 1024u);
            }
            for (uint32_t k = uint32_t(lw); k * 1024u < b412; k += 4u) {
                const uint32_t piece = uint32_t(__builtin_amdgcn_readfirstlane(int(k)));
                if (piece * 1024u + uint32_t(lane) * 16u < b412)
                    dma16(src + size_t(16u * uint32_t(tb.s_pad)) + size_t(piece) * 1024 + size_t(lane) * 16, kRing + b01 + piece * 1024u);
            }
        }
        // tile A's six barriers (raw: a pending LDS-DMA must not be drained by them), the DMA landed before the seventh
        for (int b = have_b ? 2 : 0; b < 6; ++b) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        if (have_b)
            for (int b = 0; b < 6; ++b) __builtin_amdgcn_s_barrier();
    }
}

struct FinishArgs {"""),
        (K, "        return grad ? TSAMD_FN(true, 768, 6, false, false, 2) : TSAMD_FN(false, 768, 6, false, false, 2);\n    }\n    if (weighted || rebuild) return nullptr;",
            "        return grad ? reinterpret_cast<const void *>(&tile_energy_dma2_kernel) : TSAMD_FN(false, 768, 6, false, false, 2);\n    }\n    if (weighted || rebuild) return nullptr;"),
        (K, """        r.tile_block = dim3(unsigned(e.block_threads));
        r.tile_grid = dim3(unsigned(8 * k.tiles_per_xcd));
        r.tile_lds = size_t(e.lds_bytes);""", """        r.tile_block = dim3(unsigned(e.block_threads));
        r.tile_grid = dim3(unsigned(8 * k.tiles_per_xcd));
        r.tile_lds = size_t(e.lds_bytes);
        if (r.tile_fn == reinterpret_cast<const void *>(&tile_energy_dma2_kernel)) {
            r.tile_block = dim3(1024u);
            r.tile_grid = dim3(unsigned(8 * ((k.tiles_per_xcd + 1) / 2)));
            r.tile_lds = size_t(163840);
        }"""),
        (K, "    if (lds_bytes <= configured[dev]) return hipSuccess;", "    lds_bytes = 163840;\n    if (lds_bytes <= configured[dev]) return hipSuccess;")]),
    "fold_price": ("pricing: NO finish launch; every workgroup ends with a device-scope release fence + one atomic on a counter (in the first dword of the "
                   "energy partials' last pair: results wrong) -- the floor of a finish folded into the tile kernel for small plans (the last tile of a "
                   "sphere to arrive sums its shared vertices, the last of all reduces the energy)", [
        (K, "        if (64 * nw < td.n_verts) {   // more vertices than lanes (rare;", 
            "        __threadfence();\n        __syncthreads();\n        if (tid == 0) atomicAdd(reinterpret_cast<unsigned int *>(a.partials + 2 * size_t(a.n_tiles)), 1u);\n        if (64 * nw < td.n_verts) {   // more vertices than lanes (rare;"),
        (K, "    if (f.n_finish > 0) {\n        // one vertex per thread", "    if (f.n_finish < 0) {\n        // one vertex per thread"),
        (K, "    } else if (f.energy) {\n        r.finish_fn = reinterpret_cast<const void *>(&energy_reduce_kernel);", "    } else if (f.energy && !e.grad) {\n        r.finish_fn = reinterpret_cast<const void *>(&energy_reduce_kernel);"),
        ("capi.cpp", "upload(h->d_partials, nullptr, P.tiles.size() * 2, h->device_bytes)", "upload(h->d_partials, nullptr, P.tiles.size() * 2 + 2, h->device_bytes)")]),
    "ntstore_excl": ("candidate: non-temporal stores for the EXCLUSIVE result rows (grad: not read again by this evaluation); staging rows stay temporal", [
        (K, "                dst[0] = gx * sc;\n                dst[1] = gy * sc;\n                dst[2] = gz * sc;",
            "                if (excl) {\n                    __builtin_nontemporal_store(gx * sc, dst);\n                    __builtin_nontemporal_store(gy * sc, dst + 1);\n                    __builtin_nontemporal_store(gz * sc, dst + 2);\n                } else {\n                    dst[0] = gx * sc;\n                    dst[1] = gy * sc;\n                    dst[2] = gz * sc;\n                }")]),
    "fin_nodep": ("pricing: the finish kernel's staging rows at an address computed from the vertex number (2 or 3 rows from 2.5 k: results wrong) instead of "
                  "from fin_off[k] -- what a class-major staging layout (rows at computable addresses, one dependent level less) could buy at most", [
        (K, "        const int32_t e0 = FIN_LOAD(&a.fin_off[k]), e1 = FIN_LOAD(&a.fin_off[k + 1]);   // consecutive rows, tile order",
            "        const int32_t e0 = int32_t((k * 5) >> 1), e1 = e0 + 2 + int32_t(k & 1);   // (pricing)")]),
    "ntfinish": ("the finish kernel's loads non-temporal (measured: finish kernel +50 %)", [(K, "#define FIN_LOAD(p) (*(p))", "#define FIN_LOAD(p) __builtin_nontemporal_load(p)")]),
    "tids": ("vertex ids, destinations and row table through ordinary (temporal) loads, as until round 6", [
        (K, "    const int32_t gv0 = __builtin_nontemporal_load(&as_global(a.gvid)[size_t(tile) * size_t(a.vert_stride) + size_t(int(threadIdx.x) < a.vert_stride ? threadIdx.x : 0)]);",
            "    const int32_t gv0 = as_global(a.gvid)[size_t(tile) * size_t(a.vert_stride) + size_t(int(threadIdx.x) < a.vert_stride ? threadIdx.x : 0)];"),
        (K, "dst_row = __builtin_nontemporal_load(&g_vdst[td.vert_off + 64 * vb0 + lane]);", "dst_row = g_vdst[td.vert_off + 64 * vb0 + lane];"),
        (K, "row0 = __builtin_nontemporal_load(&g_rowtab[tid < 65 ? tid : 64]);", "row0 = g_rowtab[tid < 65 ? tid : 64];")]),
    "onewg": ("pricing: ONE workgroup per CU (100 KiB of dynamic LDS requested per workgroup, same tiles, same kernel)", [
        (K, "        r.tile_lds = size_t(e.lds_bytes);", "        r.tile_lds = size_t(100 * 1024);"),
        (K, "    if (lds_bytes <= configured[dev]) return hipSuccess;", "    lds_bytes = 100 * 1024;\n    if (lds_bytes <= configured[dev]) return hipSuccess;")]),
    "onewg_nostream": ("pricing: one workgroup per CU AND no stream -- the floor of the LDS-DMA loader / consumer design (a CU's LDS holds ONE tile's records "
                       "next to the next tile's ring; 1 024 threads per workgroup cap the consumers at 12 waves + 4 loaders)", [
        (K, "reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + td.blob_off);", "reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + (td.blob_off & 0));"),
        (K, "        r.tile_lds = size_t(e.lds_bytes);", "        r.tile_lds = size_t(100 * 1024);"),
        (K, "    if (lds_bytes <= configured[dev]) return hipSuccess;", "    lds_bytes = 100 * 1024;\n    if (lds_bytes <= configured[dev]) return hipSuccess;")]),
    "rows8": ("candidate: per-vertex sums walk EIGHT rows per trip instead of four (a.veg's fullest vertices carry up to 56 rows: 7 dependent LDS round trips instead of 14)", [
        (K, "            for (int r = 0; r < rows; r += 4) {\n                const LDS_AS float *f[4];\n#pragma unroll\n                for (int u = 0; u < 4; ++u) {",
            "            for (int r = 0; r < rows; r += 8) {\n                const LDS_AS float *f[8];\n#pragma unroll\n                for (int u = 0; u < 8; ++u) {"),
        (K, "#pragma unroll\n                for (int u = 0; u < 4; ++u) {\n                    gx += f[u][0];",
            "#pragma unroll\n                for (int u = 0; u < 8; ++u) {\n                    gx += f[u][0];")]),
    "tplanes": ("the planes through ordinary (temporal) loads, as until round 6 (the product loads them non-temporally)", [
        (K, "return __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS PlaneU *>(pl + q * td.s_pad + SPT * lt)); };", "return *reinterpret_cast<const GLOBAL_AS PlaneU *>(pl + q * td.s_pad + SPT * lt); };"),
        (K, "return __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS PlaneF *>(pl + q * td.s_pad + SPT * lt)); };", "return *reinterpret_cast<const GLOBAL_AS PlaneF *>(pl + q * td.s_pad + SPT * lt); };"),
        (K, "            const VU2 t = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS VU2 *>(pl + q * td.s_pad + 2 * SPT * lt));", "            const VU2 t = *reinterpret_cast<const GLOBAL_AS VU2 *>(pl + q * td.s_pad + 2 * SPT * lt);"),
        (K, "            const VF2 t = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS VF2 *>(pl + q * td.s_pad + 2 * SPT * lt));", "            const VF2 t = *reinterpret_cast<const GLOBAL_AS VF2 *>(pl + q * td.s_pad + 2 * SPT * lt);")]),
    # ---- candidates ----
    "stagger": ("the second workgroup of every CU starts half a tile late (first 512 workgroups: 256-511 sleep ~2.7 us)", [
        (K, "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n",
            "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n    if (blockIdx.x >= 256 && blockIdx.x < 512)\n        __builtin_amdgcn_s_sleep(100);   // (64 clocks per unit: 6 400 clocks)\n")]),
    "staggerL": ("the workgroup whose LDS allocation does not start at 0 (= the second one of its CU) starts 6 400 clocks late, first 512 workgroups only", [
        (K, "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n",
            "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n    if (blockIdx.x < 512 && (__builtin_amdgcn_s_getreg((11 << 11) | 6) & 0xfff) != 0) __builtin_amdgcn_s_sleep(100);\n")]),
    "staggerL2": ("same, 3 840 clocks", [
        (K, "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n",
            "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n    if (blockIdx.x < 512 && (__builtin_amdgcn_s_getreg((11 << 11) | 6) & 0xfff) != 0) __builtin_amdgcn_s_sleep(60);\n")]),
    "staggerJ": ("odd workgroups of an XCD among its first 64 start 6 400 clocks late", [
        (K, "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n",
            "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n    if (jb < 64 && (jb & 1)) __builtin_amdgcn_s_sleep(100);\n")]),
    "stagger2": ("same, ~1.3 us", [
        (K, "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n",
            "    if (jb >= a.tiles_per_xcd || tile >= tile_end) return;\n    if (blockIdx.x >= 256 && blockIdx.x < 512)\n        __builtin_amdgcn_s_sleep(50);\n")]),
    "nofence": ("no scheduling fence between a lane's slots", [
        (K, "#define SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)", "#define SLOT_FENCE() ((void)0)")]),
    "nokeeph": ("own H re-read from LDS in pass 3 (18 VGPRs less across the barrier)", [
        (K, "constexpr int kKeepF = SPT, kKeepH = WEIGHTED ? 0 : SPT;", "constexpr int kKeepF = SPT, kKeepH = 0;")]),
    "keepf3": ("own F kept in registers through pass 3 (own H re-read from LDS): an inverted tet's F is not rebuilt", [
        (K, "constexpr int kKeepF = SPT, kKeepH = WEIGHTED ? 0 : SPT;", "constexpr int kKeepF = SPT, kKeepH = 0;"),
        (K, "                    const uint32_t w0 = q_lv01[p], w1 = q_lv23[p];\n                    float F[9], C[9];\n                    slot_F(kXS, pos_lo(w0), pos_hi(w0), pos_lo(w1), pos_hi(w1), dm, p, F);\n                    cof3(F, C);",
            "                    float C[9];\n                    cof3(Fk[p], C);")]),
    "b96": ("12-byte LDS accesses of the force array as ONE instruction each (ds_write_b96 / ds_read_b96)", [
        (K, "                    f0[0] = (-d[0] - d[3]) - d[6], f0[1] = (-d[1] - d[4]) - d[7], f0[2] = (-d[2] - d[5]) - d[8];   // (= -(f1 + f2 + f3), bit for bit)\n                    f1[0] = d[0], f1[1] = d[1], f1[2] = d[2];\n                    f2[0] = d[3], f2[1] = d[4], f2[2] = d[5];\n                    f3[0] = d[6], f3[1] = d[7], f3[2] = d[8];",
            "                    typedef float v3f __attribute__((ext_vector_type(3)));\n                    const v3f q0 = {-(d[0] + d[3] + d[6]), -(d[1] + d[4] + d[7]), -(d[2] + d[5] + d[8])}, q1 = {d[0], d[1], d[2]}, q2 = {d[3], d[4], d[5]}, q3 = {d[6], d[7], d[8]};\n                    asm volatile(\"ds_write_b96 %0, %1\" : : \"v\"(fdst[p][0]), \"v\"(q0) : \"memory\");\n                    asm volatile(\"ds_write_b96 %0, %1\" : : \"v\"(fdst[p][1]), \"v\"(q1) : \"memory\");\n                    asm volatile(\"ds_write_b96 %0, %1\" : : \"v\"(fdst[p][2]), \"v\"(q2) : \"memory\");\n                    asm volatile(\"ds_write_b96 %0, %1\" : : \"v\"(fdst[p][3]), \"v\"(q3) : \"memory\");\n                    (void)f0; (void)f1; (void)f2; (void)f3;")]),
    "fma4": ("4 * own - first neighbour as one fused multiply-add per entry (5 instructions less per gathered slot and pass)", [
        (K, """    Mat9 g0 = load_slot(nb[0]);
    acc.p01 *= 4.f; acc.p23 *= 4.f; acc.p45 *= 4.f; acc.p67 *= 4.f;
    acc.p8 *= 4.f;
    Mat9 g1 = load_slot(nb[1]);
    sub9(acc, g0);""", """    Mat9 g0 = load_slot(nb[0]);
    Mat9 g1 = load_slot(nb[1]);
    const v2f four = {4.f, 4.f};
    acc.p01 = __builtin_elementwise_fma(acc.p01, four, -g0.p01); acc.p23 = __builtin_elementwise_fma(acc.p23, four, -g0.p23);
    acc.p45 = __builtin_elementwise_fma(acc.p45, four, -g0.p45); acc.p67 = __builtin_elementwise_fma(acc.p67, four, -g0.p67);
    acc.p8 = __builtin_fmaf(acc.p8, 4.f, -g0.p8);""")]),
    "sdwa": ("row-table address of a corner in one SDWA instruction ((w >> 8) & 0xfc = byte 1 of the field & 0xfc)", [
        (K, """            const uint32_t r0 = *lds_at<const uint32_t>((w0 >> 8) & 0xfcu), r1 = *lds_at<const uint32_t>((w0 >> 24) & 0xfcu),
                           r2 = *lds_at<const uint32_t>((w1 >> 8) & 0xfcu), r3 = *lds_at<const uint32_t>((w1 >> 24) & 0xfcu);""",
            """            uint32_t ta0, ta1, ta2, ta3;
            const uint32_t kfc = 0xfcu;
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ta0) : "s"(kfc), "v"(w0));
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ta1) : "s"(kfc), "v"(w0));
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ta2) : "s"(kfc), "v"(w1));
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ta3) : "s"(kfc), "v"(w1));
            const uint32_t r0 = *lds_at<const uint32_t>(ta0), r1 = *lds_at<const uint32_t>(ta1), r2 = *lds_at<const uint32_t>(ta2), r3 = *lds_at<const uint32_t>(ta3);""")]),
    "fs": ("fma4 + sdwa", []),
    "prioA": ("wave priority by stage: pass 1 = 0, pass 2 = 1, pass 3 = 2, tail = 3", [
        (K, '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n', '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n' + "    " + '__builtin_amdgcn_s_setprio(0);\n'),
        (K, '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n', '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n' + "    " + '__builtin_amdgcn_s_setprio(1);\n'),
        (K, '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n', '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n' + "    " + '__builtin_amdgcn_s_setprio(2);\n'),
        (K, '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n', '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
    ]),
    "prioB": ("head (load issue) = 3, pass 1 = 0, pass 2 = 1, pass 3 = 2, tail = 3", [
        (K, '    const int32_t gv0 = as_global(a.gvid)[size_t(tile) * size_t(a.vert_stride) + size_t(int(threadIdx.x) < a.vert_stride ? threadIdx.x : 0)];\n', '    const int32_t gv0 = as_global(a.gvid)[size_t(tile) * size_t(a.vert_stride) + size_t(int(threadIdx.x) < a.vert_stride ? threadIdx.x : 0)];\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
        (K, '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n', '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n' + "    " + '__builtin_amdgcn_s_setprio(0);\n'),
        (K, '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n', '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n' + "    " + '__builtin_amdgcn_s_setprio(1);\n'),
        (K, '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n', '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n' + "    " + '__builtin_amdgcn_s_setprio(2);\n'),
        (K, '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n', '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
    ]),
    "prioC": ("pass 1 = 0, everything behind it = 3", [
        (K, '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n', '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n' + "    " + '__builtin_amdgcn_s_setprio(0);\n'),
        (K, '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n', '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
        (K, '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n', '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
        (K, '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n', '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
    ]),
    "prioD": ("reversed: pass 1 = 3, pass 2 = 2, pass 3 = 1, tail = 0", [
        (K, '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n', '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
        (K, '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n', '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n' + "    " + '__builtin_amdgcn_s_setprio(2);\n'),
        (K, '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n', '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n' + "    " + '__builtin_amdgcn_s_setprio(1);\n'),
        (K, '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n', '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n' + "    " + '__builtin_amdgcn_s_setprio(0);\n'),
    ]),
    "prioE": ("head = 3, pass 1 = 1, pass 2 = 2, pass 3 = 3, tail = 3", [
        (K, '    const int32_t gv0 = as_global(a.gvid)[size_t(tile) * size_t(a.vert_stride) + size_t(int(threadIdx.x) < a.vert_stride ? threadIdx.x : 0)];\n', '    const int32_t gv0 = as_global(a.gvid)[size_t(tile) * size_t(a.vert_stride) + size_t(int(threadIdx.x) < a.vert_stride ? threadIdx.x : 0)];\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
        (K, '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n', '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n' + "    " + '__builtin_amdgcn_s_setprio(1);\n'),
        (K, '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n', '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n' + "    " + '__builtin_amdgcn_s_setprio(2);\n'),
        (K, '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n', '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
        (K, '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n', '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
    ]),
    "prioF": ("pass 1 = 0, pass 2 = 0, pass 3 = 2, tail = 3", [
        (K, '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n', '    // ---- pass 1: F = Ds Dm^-1, inversion penalty ----\n' + "    " + '__builtin_amdgcn_s_setprio(0);\n'),
        (K, '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n', '    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----\n' + "    " + '__builtin_amdgcn_s_setprio(0);\n'),
        (K, '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n', '        // ---- pass 3: P = L^T H + (c2 / c1) dpen cof(F);  d = P Dm^-T (per-tet vertex forces, c1 applied at the end) ----\n' + "    " + '__builtin_amdgcn_s_setprio(2);\n'),
        (K, '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n', '        // ---- scatter the vertex forces: (f0, f1, f2, f3), f0 = -(f1 + f2 + f3), 12 bytes each ----\n' + "    " + '__builtin_amdgcn_s_setprio(3);\n'),
    ]),
    "asmpart": ("the tile's energy terms stored by an untracked (inline asm) store: no vmcnt(0) at the join of the lane-0 branch, the last wave does not start pass 3 behind a store acknowledgement", [
        (K, """            if (lane == 0) {
                g_partials[2 * size_t(tile)] = s;
                g_partials[2 * size_t(tile) + 1] = b;
            }""", """            {
                const v2u sw = __builtin_bit_cast(v2u, s), bw = __builtin_bit_cast(v2u, b);
                const v4u val = {uint32_t(__builtin_amdgcn_readlane(int(sw.x), 0)), uint32_t(__builtin_amdgcn_readlane(int(sw.y), 0)),
                                 uint32_t(__builtin_amdgcn_readlane(int(bw.x), 0)), uint32_t(__builtin_amdgcn_readlane(int(bw.y), 0))};
                GLOBAL_AS double *dst = g_partials + 2 * size_t(tile);
                asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(val) : "memory");
            }""")]),
    "identblocks": ("vertex blocks of the per-vertex sums in wave order (no reversal of waves 4-7)", [
        (K, "        if (wave & 4) {   // (a permutation", "        if (false) {   // (a permutation")]),
}


# debugging aids of `dma2`: one ingredient of the DMA-fed tile taken from HBM again
_D = VARIANTS["dma2"][1]
VARIANTS["dma2_chk"] = ("dma2 debug: tile B takes Dm^-1 from HBM, compares the ring with it and adds 1000 per mismatching lane-load to the barrier energy", _D + [
    (K, """            const v4f t = *lds_at<const v4f>(kRing + 4u * uint32_t(q - 2) * uint32_t(td.s_pad) + 16u * uint32_t(lt));
            lo[0] = t.x, lo[1] = t.y, hi[0] = t.z, hi[1] = t.w;""",
        """            const v4f t = *lds_at<const v4f>(kRing + 4u * uint32_t(q - 2) * uint32_t(td.s_pad) + 16u * uint32_t(lt));
            const VF2 h = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS VF2 *>(pl + q * td.s_pad + 2 * SPT * lt));
            if (active && (t.x != h[0] || t.y != h[1] || t.z != h[2] || t.w != h[3])) dbg_mismatch += 1000.f;
            lo[0] = h[0], lo[1] = h[1], hi[0] = h[2], hi[1] = h[3];"""),
    (K, "    auto pair_u = [&](int q, VU &lo, VU &hi) {", "    float dbg_mismatch = 0.f;\n    auto pair_u = [&](int q, VU &lo, VU &hi) {"),
    (K, "    float e_b = 0.f, e_s = 0.f;", "    float e_b = dbg_mismatch, e_s = 0.f;\n    if (FEED == 1 && tid == 0 && *lds_at<const uint32_t>(kRingPos - 16u) != uint32_t(tile)) e_b += 1.0e6f;"),
    (K, """        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();""", """        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (ll == 0) *lds_at<uint32_t>(kRingPos - 16u) = uint32_t(tile_b);   // (debug: "everything of tile B has landed")
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();""")])

# `dma2` leaves the head of the DMA-fed tile on HBM: the LDS is full (two tile maps' worth: records + ring), so the neighbour planes, the row
# table and the descriptor of tile B are fetched when tile B starts -- with its planes already in LDS there is nothing left to hide them
# behind.  `dma2p` has the CONSUMERS fetch them into registers (five VGPRs; the 1 024-thread workgroup has 128) a phase before tile A ends.
VARIANTS["dma2p"] = ("dma2 + tile B's neighbour planes / row table / descriptor prefetched into the consumers' registers at tile A's scatter", _D + [
    (K, "constexpr uint32_t kRing = 81920u; ", "struct NextHead {\n    TileDesc td;\n    uint32_t nb[4];\n    uint32_t row0;\n};\nconstexpr uint32_t kRing = 81920u; "),
    (K, "__device__ __forceinline__ void tile_body(const KernelArgs &a, const int tile, int32_t gv0)\n{",
        "__device__ __forceinline__ void tile_body(const KernelArgs &a, const int tile, int32_t gv0, NextHead *nh = nullptr)\n{"),
    (K, "    const TileDesc td = a.tiles[tile];", "    const TileDesc td = FEED == 1 ? nh->td : a.tiles[tile];"),
    (K, "    pair_u(2, q_nb01, q_nb23);", """    if (FEED == 1) {
        q_nb01[0] = nh->nb[0], q_nb01[1] = nh->nb[1], q_nb23[0] = nh->nb[2], q_nb23[1] = nh->nb[3];
    } else {
        pair_u(2, q_nb01, q_nb23);
    }"""),
    (K, "    if (WITH_GRAD) row0 = __builtin_nontemporal_load(&g_rowtab[tid < 65 ? tid : 64]);",
        "    if (WITH_GRAD) row0 = FEED == 1 ? nh->row0 : __builtin_nontemporal_load(&g_rowtab[tid < 65 ? tid : 64]);"),
    (K, "        __syncthreads();   // all waves done with H and with the staged positions", """        if (FEED == 0 && nh) {   // the head of the NEXT tile, a phase ahead (behind dst_row: the in-order wait for that one leaves these in flight)
            const GLOBAL_AS uint32_t *npl = reinterpret_cast<const GLOBAL_AS uint32_t *>(g_blob + nh->td.blob_off);
            const int nlt = tid < nh->td.s_pad / SPT ? tid : 0;
            const v4u t = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS v4u *>(npl + 2 * nh->td.s_pad + 4 * nlt));
            nh->nb[0] = t.x, nh->nb[1] = t.y, nh->nb[2] = t.z, nh->nb[3] = t.w;
            const GLOBAL_AS uint16_t *nrt = reinterpret_cast<const GLOBAL_AS uint16_t *>(npl + size_t(kPlanes) * nh->td.s_pad);
            nh->row0 = __builtin_nontemporal_load(&nrt[tid < 65 ? tid : 64]);
        }
        __syncthreads();   // all waves done with H and with the staged positions"""),
    (K, """        tile_body<true, false, false, 2, 0>(a, tile_a, gv0);""", """        NextHead nh;
        nh.td = a.tiles[have_b ? tile_b : tile_a];   // (scalar loads: back with tile A's own descriptor)
        tile_body<true, false, false, 2, 0>(a, tile_a, gv0, &nh);"""),
    (K, "        if (have_b) tile_body<true, false, false, 2, 1>(a, tile_b, 0);", "        if (have_b) tile_body<true, false, false, 2, 1>(a, tile_b, 0, &nh);")])

# where the time of a pair goes: 100 MHz timestamps (s_memrealtime) of wave 0 and of loader wave 12 of two mid-launch workgroups, printed
_T = "__builtin_amdgcn_s_memrealtime()"
VARIANTS["dma2p_t"] = ("dma2p + phase timestamps of two mid-launch workgroups (printf; run a handful of evaluations only)", VARIANTS["dma2p"][1] + [(_f, _o, _n.replace("@T@", _T)) for _f, _o, _n in [
    (K, "    uint32_t row0;\n};\nconstexpr uint32_t kRing", "    uint32_t row0;\n    unsigned long long ts[3];\n};\nconstexpr uint32_t kRing"),
    (K, "    __syncthreads();\n    // behind the barrier: what pass 2 and the end of the tile need",
        "    __syncthreads();\n    if (nh) nh->ts[0] = @T@;\n    // behind the barrier: what pass 2 and the end of the tile need"),
    (K, "    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----",
        "    if (nh) nh->ts[1] = @T@;\n    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----"),
    (K, "        if (FEED == 0 && nh) {   // the head of the NEXT tile", "        if (nh) nh->ts[2] = @T@;\n        if (FEED == 0 && nh) {   // the head of the NEXT tile"),
    (K, "        NextHead nh;\n", "        NextHead nh;\n        const unsigned long long T0 = @T@;\n"),
    (K, "        if (have_b) tile_body<true, false, false, 2, 1>(a, tile_b, 0, &nh);", """        const unsigned long long T2 = @T@;
        NextHead nhb = nh;
        if (have_b) tile_body<true, false, false, 2, 1>(a, tile_b, 0, &nhb);
        const unsigned long long T3 = @T@;
        if (tid == 0 && have_b && (blockIdx.x == 4000 || blockIdx.x == 4003 || blockIdx.x == 7001))
            printf("cons blk %d (x10 ns)  A: positions %llu pass1 %llu scatter %llu end %llu | join %llu | B: positions %llu pass1 %llu scatter %llu end %llu\\n", int(blockIdx.x),
                   nh.ts[0] - T0, nh.ts[1] - T0, nh.ts[2] - T0, T1 - T0, T2 - T1, nhb.ts[0] - T2, nhb.ts[1] - T2, nhb.ts[2] - T2, T3 - T2);"""),
    (K, "        __syncthreads();   // tile A is done with its LDS; the loaders arrive here", "        const unsigned long long T1 = @T@;\n        __syncthreads();   // tile A is done with its LDS; the loaders arrive here"),
    (K, "        const int ll = tid - 768, lw = ll >> 6, lane = ll & 63;", "        const int ll = tid - 768, lw = ll >> 6, lane = ll & 63;\n        const unsigned long long L0 = @T@;\n        unsigned long long L1 = L0, L2 = L0;"),
    (K, "            // the planes: device-image bytes of planes 0-1 and of planes 4-12, 1 KiB per wave-instruction, no VGPRs", "            L1 = @T@;\n            // the planes: device-image bytes of planes 0-1 and of planes 4-12, 1 KiB per wave-instruction, no VGPRs"),
    (K, "        // tile A's six barriers (raw: a pending LDS-DMA must not be drained by them), the DMA landed before the seventh", "        L2 = @T@;\n        // tile A's six barriers (raw: a pending LDS-DMA must not be drained by them), the DMA landed before the seventh"),
    (K, """        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        if (have_b)""", """        const unsigned long long L3 = @T@;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long L4 = @T@;
        if (ll == 0 && have_b && (blockIdx.x == 4000 || blockIdx.x == 4003 || blockIdx.x == 7001))
            printf("load blk %d (x10 ns)  dma issue starts %llu, issued %llu, tile A's barriers passed %llu, landed %llu\\n", int(blockIdx.x), L1 - L0, L2 - L0, L3 - L0, L4 - L0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        if (have_b)""")]])

# the product kernel's phases as seen by wave 0 of a few mid-launch workgroups (two workgroups per CU: each one's phases stretch by what the other takes)
_BLK = "(blockIdx.x == 4000 || blockIdx.x == 4003 || blockIdx.x == 7001 || blockIdx.x == 9002 || blockIdx.x == 12005 || blockIdx.x == 15000)"
VARIANTS["stamps"] = ("product kernel + 100 MHz phase timestamps of wave 0 in six mid-launch workgroups (printf; run a handful of evaluations only)", [(_f, _o, _n.replace("@T@", _T).replace("@BLK@", _BLK)) for _f, _o, _n in [
    (K, "    const TileDesc td = a.tiles[tile];", "    unsigned long long ts[8];\n    ts[0] = @T@;\n    const TileDesc td = a.tiles[tile];"),
    (K, "    __syncthreads();\n    // behind the barrier: what pass 2 and the end of the tile need", "    __syncthreads();\n    ts[1] = @T@;\n    // behind the barrier: what pass 2 and the end of the tile need"),
    (K, "    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----", "    ts[2] = @T@;\n    // ---- pass 2: H = L F on owned slots (0 on halo slots), E_s = 1/2 |H|^2 ----"),
    (K, "    __syncthreads();  // every read of F is done; overwrite it with H in place", "    __syncthreads();  // every read of F is done; overwrite it with H in place\n    ts[3] = @T@;"),
    (K, "        publish_wave_sums();\n        __syncthreads();\n        sum_waves();\n", "        publish_wave_sums();\n        __syncthreads();\n        ts[4] = @T@;\n        sum_waves();\n"),
    (K, "        __syncthreads();   // all waves done with H and with the staged positions", "        __syncthreads();   // all waves done with H and with the staged positions\n        ts[5] = @T@;"),
    (K, "        __syncthreads();\n\n        // ---- per-vertex sums: lane = vertex", "        __syncthreads();\n        ts[6] = @T@;\n\n        // ---- per-vertex sums: lane = vertex"),
    (K, "    }\n#undef t_own\n}", """        ts[7] = @T@;
        if ((tid == 0 || tid == 704) && @BLK@)
            printf("blk %d wave %d (x10 ns): positions %llu pass1 %llu pass2 %llu H-stored %llu pass3 %llu scatter %llu sums+stores %llu | total %llu\\n", int(blockIdx.x), tid >> 6,
                   ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6], ts[7] - ts[0]);
    }
#undef t_own
}""")]])

# ... and the inside of the per-vertex sums (wave 0 = the block of the fullest vertices): row loop / wait for the destination ids / stores issued
VARIANTS["stamps2"] = ("stamps + the per-vertex sums split into row loop, wait for vdst, stores", [(_f, _o, _n.replace("@T@", _T).replace("@BLK@", _BLK)) for _f, _o, _n in [
    (K, "    const TileDesc td = a.tiles[tile];", "    unsigned long long ts[4];\n    int sum_rows = 0;\n    const TileDesc td = a.tiles[tile];"),
    (K, "        __syncthreads();\n\n        // ---- per-vertex sums: lane = vertex", "        __syncthreads();\n        ts[0] = @T@;\n\n        // ---- per-vertex sums: lane = vertex"),
    (K, "            if (v < td.n_verts) {\n                const bool excl = row >= 0;", "            sum_rows = rows;\n            asm volatile(\"\" : \"+v\"(gx), \"+v\"(gy), \"+v\"(gz));\n            ts[1] = @T@;\n            asm volatile(\"\" : \"+v\"(row));\n            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n            ts[2] = @T@;\n            if (v < td.n_verts) {\n                const bool excl = row >= 0;"),
    (K, "    }\n#undef t_own\n}", """        ts[3] = @T@;
        if ((tid == 0 || tid == 64 || tid == 192) && @BLK@)
            printf("blk %d wave %d (x10 ns): rows %d: row loop %llu, wait for vdst %llu, stores issued %llu\\n", int(blockIdx.x), tid >> 6, sum_rows, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2]);
    }
#undef t_own
}""")]])

# candidate (round 6, from the stamps): the row loop of the per-vertex sums is one wave's dependent chain -- ~45 instructions per four rows, then a
# wait for eight LDS reads: 0.17 us per trip, 2.3 us for a.veg's 56-row hubs (wave 0 is the last wave of its workgroup by that much).  Two trips in
# flight: the reads of trip k + 1 are issued before trip k is added up (same order of additions: bit-identical results).
VARIANTS["sums_pipe"] = ("candidate: per-vertex sums software-pipelined, two trips of four rows in flight", [
    (K, """            for (int r = 0; r < rows; r += 4) {
                const LDS_AS float *f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t base = uint32_t(__builtin_amdgcn_readlane(int(tab), r + u)), w = uint32_t(__builtin_amdgcn_readlane(int(wid), r + u));
                    f[u] = lds_at<const float>(v12 < w ? base + v12 : kZeroEntry);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    gx += f[u][0];
                    gy += f[u][1];
                    gz += f[u][2];
                }
            }""", """            float A[4][3], B[4][3];
            auto fetch = [&](int r, float (&d)[4][3]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t base = uint32_t(__builtin_amdgcn_readlane(int(tab), r + u)), w = uint32_t(__builtin_amdgcn_readlane(int(wid), r + u));
                    const LDS_AS float *f = lds_at<const float>(v12 < w ? base + v12 : kZeroEntry);
                    d[u][0] = f[0], d[u][1] = f[1], d[u][2] = f[2];
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto add = [&](const float (&d)[4][3]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    gx += d[u][0];
                    gy += d[u][1];
                    gz += d[u][2];
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (rows > 0) {
                fetch(0, A);
                int r = 4;
                for (; r + 4 < rows; r += 8) {   // trips r / 4 and r / 4 + 1 both exist
                    fetch(r, B);
                    add(A);
                    fetch(r + 4, A);
                    add(B);
                }
                if (r < rows) {
                    fetch(r, B);
                    add(A);
                    add(B);
                } else {
                    add(A);
                }
            }""")])

COMBOS = {"fs": ["fma4", "sdwa"], "stamps_rows8": ["stamps", "rows8"]}    # ("undef" -- inactive lanes' arrays left undefined -- was adopted by the product kernel)


FLAG_VARIANTS = {   # compiler-flag builds of the unpatched source: name -> function(list of device flags) -> list
    "sched_default": lambda f: [x for x in f if x not in ("-mllvm", "-amdgpu-sched-strategy=max-ilp")],
    "sched_memclause": lambda f: [("-amdgpu-sched-strategy=max-memory-clause" if x == "-amdgpu-sched-strategy=max-ilp" else x) for x in f],
    "sched_iterilp": lambda f: [("-amdgpu-sched-strategy=iterative-ilp" if x == "-amdgpu-sched-strategy=max-ilp" else x) for x in f],
}
# -D builds: the define goes to EVERY translation unit (host and device), the source is unpatched
DEFINE_VARIANTS = {
    "unpaired": ("the device image of rounds 1-5: planes one after the other, thirteen 8-byte loads per lane (the product interleaves the planes that are "
                 "loaded together: six 16-byte + one 8-byte load, plan.h: planes_paired)", ["-DTSAMD_PAIRED_PLANES=0"]),
}
for _n, (_d, _f) in DEFINE_VARIANTS.items():
    VARIANTS[_n] = (_d, [])
VARIANTS["stamps_rows8"] = ("stamps on top of rows8: does halving the longest walk shorten the workgroup, and what do the other phases do?", [])
for _n in FLAG_VARIANTS:
    VARIANTS[_n] = (f"compiler flags: {_n}", [(K, "// gfx950 (MI355X / CDNA4) kernels", "// gfx950 (MI355X / CDNA4) kernels")])


def build(name: str) -> str:
    if name in DEFINE_VARIANTS:
        return _build.build_variant(name, DEFINE_VARIANTS[name][1])
    desc, patches = VARIANTS[name]
    if name in COMBOS:
        patches = [q for n in COMBOS[name] for q in VARIANTS[n][1]]
    srcs = {}
    for f, old, new in patches:
        if f not in srcs:
            srcs[f] = open(os.path.join(_build.CSRC, f)).read()
        if old not in srcs[f]:
            raise SystemExit(f"variant {name}: patch anchor not found in {f}: {old[:70]!r}")
        srcs[f] = srcs[f].replace(old, new)
    hipcc = _build._hipcc()
    _build.build()                                      # the product objects of everything that is not patched
    os.makedirs(_build._OBJ, exist_ok=True)
    objs = []
    for src in _build.SOURCES:
        obj = os.path.join(_build._OBJ, os.path.splitext(src)[0] + ".o")
        if src in srcs:
            patched = os.path.join(_build.CSRC, f"_lab_{name}_{src}")      # (next to the original: relative includes resolve)
            open(patched, "w").write(srcs[src])
            obj = os.path.join(_build._OBJ, f"{os.path.splitext(src)[0]}_lab_{name}.o")
            try:
                if src.endswith(".hip"):
                    dflags = FLAG_VARIANTS[name](list(_build.DEVICE_FLAGS)) if name in FLAG_VARIANTS else _build.DEVICE_FLAGS
                    cmd = [hipcc, "-c", patched, "-o", obj] + _build.HOST_FLAGS + dflags + _build.SOURCE_FLAGS.get(src, [])
                else:
                    cmd = [hipcc] + _build.HOST_FLAGS + ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c", patched, "-o", obj]
                subprocess.check_call(cmd)
            finally:
                os.remove(patched)
        objs.append(obj)
    out = os.path.join(os.path.dirname(_build.LIB), f"libtssplat_amd_{name}.so")
    subprocess.check_call([hipcc, "-shared", "-o", out] + objs + [f"--offload-arch={_build.ARCH}", "-pthread"])
    return out


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] == "--list":
        for k, (d, _) in VARIANTS.items():
            print(f"{k:14s} {d}")
        raise SystemExit(0)
    for n in sys.argv[1:]:
        print(build(n))
