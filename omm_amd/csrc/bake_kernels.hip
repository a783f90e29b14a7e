// bake_kernels.hip -- HIP kernels of the MI355X opacity-micromap baker (gfx950, wave64).
//
//   classify_tiles      coarse (SAT) + fine (level-line) classification of micro-triangles,
//                       LDS work queue between the two passes, 2-/1-bit packed output
//   digest_items        XXH64(seed 42) of the 3-state byte stream of each non-uniform work item
//   sat_* / tail_*      summed-area-table build, dedup/sort/pack helpers
//
// No MFMA anywhere: this is a sampling / reduction path (fp32 VALU + sqrt/div + texel gathers).
// Compile with -ffp-contract=off (see classify_device.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "bake_types.h"
#include "classify_device.h"
#include "bake_kernels.h"

namespace ommx {

// ------------------------------------------------------------------------------------------------
// Classification.
//
// The reference classifies every micro-triangle on its own (4^N SAT tests + level-line passes per work item).  The
// bird curve is hierarchical -- micro-triangle i of level N lies inside micro-triangle i >> 2(N-k) of level k -- so a
// whole sub-triangle can be settled with ONE summed-area-table query when its texel footprint is uniformly <= / >
// cutoff (region_state() in classify_device.h proves the result equals the per-micro-triangle coarse pass):
//
//   triage_items      one lane per work item: the whole triangle (level-0 sub-triangle).  Uniform items need no
//                     per-micro-triangle work and no state storage at all.
//   classify_tiles    one workgroup (256 threads = 4 waves) per tile of TILE = 1024 consecutive micro-triangles of
//                     an ACTIVE (non-uniform) item:
//        0. tile-level query (level N-5), then one query per 64-micro-triangle group (level N-3): a wave whose group is
//           settled skips phase 1 for it entirely (a wave = exactly one group)
//        1. remaining lanes: bird-curve micro-triangle -> coarse SAT test; unresolved ones are appended to an LDS
//           queue by wave-ballot compaction, so that
//        2. the expensive level-line pass runs on densely packed lanes,
//        3. states are packed LSB-first into 32-bit words (coalesced stores) and the tile's state mask is OR-reduced
//           for the uniform-OMM ("special index") detection.
// ------------------------------------------------------------------------------------------------
// curve-free-region test (region_curve.h) of the 64-groups inside the persistent kernel: inlined, or as a call with a register allocation of its own
// (measured on the metric configuration: 15.55 vs 15.60 ms -- the same; inlined it needs 40 bytes per lane less scratch)
#ifndef OMMX_RC_CALL
#define OMMX_RC_CALL 0
#endif
#if OMMX_RC_CALL
#define OMMX_RC_GROUP_TEST region_curve_state_call
#else
#define OMMX_RC_GROUP_TEST region_curve_state_impl
#endif
#ifndef OMMX_RC_LEVELS   // bit 0: items, bit 1: tiles, bit 2: 64-groups (A/B builds; the product ships all three)
#define OMMX_RC_LEVELS 7
#endif
constexpr int BLOCK = 256;
constexpr int GROUP = 64;

__device__ __forceinline__ float item_max_abs(const float* __restrict__ uv)
{
    float m = __builtin_fabsf(uv[0]);
    #pragma unroll
    for (int k = 1; k < 6; ++k) { const float a = __builtin_fabsf(uv[k]); m = a > m ? a : m; }
    return m;
}

// Block-uniform values that arrive through vector loads / LDS (the tile record, the item's UVs) are moved to scalar registers: the
// classification loops are VGPR-bound (80 VGPRs = 6 waves per SIMD), and a uniform value parked in a VGPR costs 64 lanes of it.
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uniform_f32(float v) { return __uint_as_float(uniform_u32(__float_as_uint(v))); }

__device__ __forceinline__ TexWindow no_window()
{
    TexWindow W; W.tex = (lds_float*)0; W.sat = (lds_u32*)0; W.base = nullptr; W.sx = W.sy = 0; W.w = W.h = 0;
    return W;
}

__global__ __launch_bounds__(256) void triage_items(ClassifyParams P, const float* __restrict__ uv, const uint8_t* __restrict__ level, const uint8_t* __restrict__ degenerate,
                                                    const SetupCounters* __restrict__ counters, uint32_t* __restrict__ stateMask, uint8_t* __restrict__ active,
                                                    uint32_t* __restrict__ zero, uint32_t zeroWords)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t w = i; w < zeroWords; w += gridDim.x * blockDim.x) zero[w] = 0u;   // (run_prep's tile states: the launch behind this one)
    if (i >= counters->numItems) return;
    int st = -1;
    const float* t = uv + 6ull * i;
    const MicroTri whole = micro_triangle(t, 0u, 0u);
    const float maxAbs = item_max_abs(t);
    if (P.useCoarse) st = region_state<ModeDynamic>(P, whole, maxAbs, no_window());
    // small triangles in a mixed neighbourhood that the level curve does not reach (region_curve.h): uniform as well
    if ((OMMX_RC_LEVELS & 1) && st < 0 && region_curve_applies(P) && !degenerate[i]) {
        const DevMip& m = P.mips[0];
        st = region_curve_state<ModeDynamic>(P, P.texIsFp32 != 0, rc_shape(t, m.fw, m.fh, m.w, m.h, level[i]), whole, maxAbs);
    }
    stateMask[i] = st >= 0 ? (1u << st) : 0u;
    active[i] = st >= 0 ? 0 : 1;
}

void launch_triage(const ClassifyParams& P, const float* uv, const uint8_t* level, const uint8_t* degenerate, const SetupCounters* counters, uint32_t maxItems,
                   uint32_t* stateMask, uint8_t* active, void* prepScratch, hipStream_t stream)
{
    if (maxItems == 0) return;
    hipLaunchKernelGGL(triage_items, dim3((maxItems + 255u) / 256u), dim3(256), 0, stream, P, uv, level, degenerate, counters, stateMask, active,
                       (uint32_t*)prepScratch, prep_state_words(maxItems));
}

// index narrowing (bake_cpu_impl.cpp:1872-1902) for results that stay on the device
__global__ __launch_bounds__(256) void narrow_indices(const int32_t* __restrict__ in, uint32_t n, int bytesPerIndex, void* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (bytesPerIndex == 1) ((int8_t*)out)[i] = (int8_t)in[i];
    else if (bytesPerIndex == 2) ((int16_t*)out)[i] = (int16_t)in[i];
    else ((int32_t*)out)[i] = in[i];
}
void launch_narrow_indices(const int32_t* in, uint32_t n, int bytesPerIndex, void* out, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(narrow_indices, dim3((n + 255u) / 256u), dim3(256), 0, stream, in, n, bytesPerIndex, out);
}

// ------------------------------------------------------------------------------------------------
// Tile triage.  A tile = TILE consecutive bird-curve micro-triangles of an active item = its level-(N - log4 TILE) sub-triangle.
// One lane per tile asks the summed-area table about the whole sub-triangle (region_state, global SAT reads):
//   settled  -> the wave writes the tile's constant packed states (one coalesced 16-byte store per lane) and the lane folds the
//               state into the item's mask / known count; the tile never becomes a workgroup
//   open     -> a 16-byte record {item, tile, level, texel rectangle} is appended to the device-resident tile queue that the
//               persistent classify_tiles launch drains (all levels of one tile size share one queue: no per-level launches)
// At the bench configuration 59 % of the 2.04 M tiles are settled here.
// ------------------------------------------------------------------------------------------------
// The 4096-tile queue is cut into sections that ONE persistent launch drains in order (TileLevels / TileSections / ClassifyPlan: bake_kernels.h).  Ordinary
// bakes: one section.  Streamed bakes (ommCpuBake) with K ranges: section 2k = the tiles of range k's own items, at records [cut[k], ..) of the queue's
// first copy; section 2k + 1 = the tiles of EARLY items of later ranges whose family starts in range k, in the queue's second copy ([total, 2 total)),
// ordered by range with a staging pass (early_tiles_*).
// record = 7 x uint4 (everything a tile's workgroup needs, in ONE memory round trip):
//   [0] x = item | degenerate << 30 | rectOk << 31, y = tile in item | dead << 23 | level << 24 | no-single-texel-micro-triangle << 31, z = sx | sy << 16, w = ex | ey << 16 (addressed texel rectangle)
//   [1] the item's uv[0..3]      [2] uv[4], uv[5], address of the tile's packed states (lo, hi)
//   [3..6] one verdict byte per 64-group of the tile (triage_groups below; a 1024-tile uses the first 16)
constexpr uint32_t kTileRecordWords = 7;   // uint4 per record (bake_kernels.h: kTileRecordBytes)
static_assert(kTileRecordWords * 16u == kTileRecordBytes, "tile record size");
// verdict byte of a 64-group: 0..3 = every micro-triangle of the group has that state; kGvUnknown = test them one by one; kGvAllOpen = none of them is resolved by
// the coarse pass; kGvEdgeFree | code = region_curve_state()'s edge-free verdict -(kRegionEdgeFreeBase + code), code = 16 above + mask of the wrong-side corners
constexpr uint32_t kGvUnknown = 0xFFu, kGvAllOpen = 0xFEu, kGvEdgeFree = 0x80u;
constexpr uint32_t kTileFat = 1u << 22;    // word [0].y of a record: the work item has the corner bound (RcShape::ok & fat, region_curve.h): fine_single_texel may skip corner votes (rc_corners_far)
constexpr uint32_t kTileDead = 1u << 23;   // word [0].y of a record: every group of the tile was settled by triage_groups (the tile index needs 12 of the 24 low bits)
__device__ __forceinline__ uint32_t group_verdict_byte(int gs)
{
    return gs >= 0 ? (uint32_t)gs : (gs == kRegionUnknown ? kGvUnknown : (gs == kRegionAllOpen ? kGvAllOpen : (kGvEdgeFree | ((uint32_t)(-gs - kRegionEdgeFreeBase) & 31u))));
}
__device__ __forceinline__ int group_verdict(uint32_t b)
{
    return b < 4u ? (int)b : (b == kGvUnknown ? kRegionUnknown : (b == kGvAllOpen ? kRegionAllOpen : -(kRegionEdgeFreeBase + (int)(b & 31u))));
}
template <int TILE>
__global__ __launch_bounds__(256) void triage_tiles(ClassifyParams P, ItemArrays A, const uint32_t* __restrict__ activeIds, TileLevels L,
                                                    uint4* __restrict__ queue, uint32_t* __restrict__ queueCtl, TileSections S,
                                                    const uint8_t* __restrict__ early, const uint32_t* __restrict__ earlyLead, uint4* __restrict__ earlyStage)
{
    constexpr uint32_t TILE_LOG4 = TILE == 4096 ? 6u : 5u;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t* sectionTails = queueCtl + kSecTails;
    const uint32_t stride = S.n > 1u ? 2u : 1u;   // (streamed: the ranges' own sections are the even ones)
    if (t < S.n) queueCtl[kSecBases + stride * t] = S.cut[t];   // (the persistent launch walks the sections by these; the odd ones: early_tiles_bases)
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = t < L.tileStart[L.n];
    uint32_t sec = 0; bool isEarly = false, bigMicro = false, fatItem = false;
    int st = -1; uint32_t item = 0, tileInItem = 0, level = TILE_LOG4; TexRect r; r.sx = r.sy = r.ex = r.ey = 0; r.ok = false;
    float uvv[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    const uint32_t bits = (uint32_t)P.format, tileBytes = (uint32_t)TILE * bits / 8u;
    if (live) {
        uint32_t k = 0;
        while (k + 1 < L.n && t >= L.tileStart[k + 1]) ++k;
        level = L.level[k];
        const uint32_t shift = 2u * (level - TILE_LOG4), rel = t - L.tileStart[k];   // tiles per item = 4^(level - log4 TILE)
        item = activeIds[L.first[k] + (rel >> shift)]; tileInItem = rel & ((1u << shift) - 1u);
        if (S.n > 1u) {
            isEarly = TILE == 4096 && early && early[item] == 1;
            while (sec + 1u < S.n && t >= S.cut[sec + 1u]) ++sec;           // the item's own range ...
            if (isEarly) { const uint32_t ls = section_of_position(earlyLead[item], L, S); sec = ls < sec ? ls : sec; }   // ... or the earlier one of its family's first member (early_range)
        }
        const float* uv = A.uv + 6ull * item;
        #pragma unroll
        for (int q = 0; q < 6; ++q) uvv[q] = uv[q];
        const float maxAbs = item_max_abs(uv);
        {   // every micro-triangle of an item has the item's bounding-box extents / 2^level (up to rounding): wider or taller than a texel means that no raster box
            // is one texel, i.e. the single-texel pass of classify_tiles would send all of them to the generic path anyway (asset-sized triangles: it is skipped)
            const DevMip& m0 = P.mips[0];
            const float ls = __uint_as_float((127u - level) << 23);
            const float ex = (std_max(std_max(uv[0], uv[2]), uv[4]) - std_min(std_min(uv[0], uv[2]), uv[4])) * m0.fw * ls;
            const float ey = (std_max(std_max(uv[1], uv[3]), uv[5]) - std_min(std_min(uv[1], uv[3]), uv[5])) * m0.fh * ls;
            bigMicro = ex > 1.01f || ey > 1.01f;
        }
        const MicroTri sub = micro_triangle(uv, tileInItem, level - TILE_LOG4);
        r = region_rect<ModeDynamic>(P, sub, maxAbs);
        // (a tile that IS its work item -- level 6 here, level 5 in the 1024-tile queue -- was asked both questions by triage_items, with the same arguments,
        //  and is on the active list because the answer was no: configs[1], 10 192 such tiles, 22 -> 9 us)
        constexpr bool kSameTests = ((OMMX_RC_LEVELS & 1) != 0) == ((OMMX_RC_LEVELS & 2) != 0);
        const bool asked = kSameTests && level == TILE_LOG4;
        if (P.useCoarse && !asked) st = region_state<ModeDynamic>(P, sub, maxAbs, no_window());
        // a tile in a mixed neighbourhood that the level curve does not reach (region_curve.h): settled like a uniform one
        if (st < 0 && region_curve_applies(P) && !A.degenerate[item]) {
            const DevMip& m = P.mips[0];
            const RcShape shape = rc_shape(uv, m.fw, m.fh, m.w, m.h, level);
            fatItem = shape.ok != 0 && shape.fat != 0;   // (into the record of an open tile: the persistent launch's single-texel pass asks per micro-triangle)
            if ((OMMX_RC_LEVELS & 2) && !asked) st = region_curve_state<ModeDynamic>(P, P.texIsFp32 != 0, shape, sub, maxAbs);
        }
    }
    // ---- open tiles: wave-compacted append, section by section (a wave sees one section, two at a range boundary, early items apart) ----
    // (the tiles of early items go to the staging list with their range in bits 16..21 of word 1; early_tiles_scatter orders them by range)
    const bool open = live && st < 0;
    const uint32_t key = sec * 2u + (isEarly ? 1u : 0u);
    unsigned long long todo = __ballot(open);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
        const unsigned long long ob = __ballot(open && key == k0) & todo;
        todo &= ~ob;
        uint32_t wbase = 0;
        if ((int)lane == leader) {
            const uint32_t n = (uint32_t)__popcll(ob), s0 = k0 >> 1;
            if (k0 & 1u) { wbase = atomicAdd(queueCtl + kCtlEarlyStaged, n); atomicAdd(sectionTails + 2u * s0 + 1u, n); }
            else wbase = S.cut[s0] + atomicAdd(sectionTails + stride * s0, n);
        }
        wbase = __shfl(wbase, leader);
        if (open && key == k0) {
            uint4* rec = ((k0 & 1u) ? earlyStage : queue) + (size_t)kTileRecordWords * (wbase + __popcll(ob & ((1ull << lane) - 1ull)));
            const unsigned long long dst = (unsigned long long)(A.states + A.stateOfs[item] + (size_t)tileInItem * tileBytes);
            rec[0] = make_uint4(item | (A.degenerate[item] ? 0x40000000u : 0u) | (r.ok ? 0x80000000u : 0u), tileInItem | (level << 24) | (bigMicro ? 0x80000000u : 0u) | (fatItem ? kTileFat : 0u) | ((k0 & 1u) ? (k0 >> 1) << 16 : 0u),
                                (uint32_t)r.sx | ((uint32_t)r.sy << 16), (uint32_t)r.ex | ((uint32_t)r.ey << 16));
            rec[1] = make_uint4(__float_as_uint(uvv[0]), __float_as_uint(uvv[1]), __float_as_uint(uvv[2]), __float_as_uint(uvv[3]));
            rec[2] = make_uint4(__float_as_uint(uvv[4]), __float_as_uint(uvv[5]), (uint32_t)dst, (uint32_t)(dst >> 32));
        }
    }
    // ---- settled tiles: constant states, written by the whole wave ----
    if (live && st >= 0) {
        atomicOr(&A.stateMask[item], 1u << st);
        if (P.wantKnownCount && st < 2) atomicAdd(&A.knownCount[item], (uint32_t)TILE);
    }
    unsigned long long sb = __ballot(live && st >= 0);
    const unsigned long long dstMine = (live && st >= 0) ? (unsigned long long)(A.states + A.stateOfs[item] + (size_t)tileInItem * tileBytes) : 0ull;
    while (sb) {
        const int src = __ffsll((long long)sb) - 1;
        sb &= sb - 1ull;
        const unsigned long long d = ((unsigned long long)(uint32_t)__shfl((int)(dstMine >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)dstMine, src);
        const uint32_t s = (uint32_t)__shfl(st, src);
        uint32_t v = 0;
        for (uint32_t b = 0; b < 32u; b += bits) v |= s << b;
        for (uint32_t o = lane * 16u; o < tileBytes; o += 64u * 16u) *(uint4*)((uint8_t*)d + o) = make_uint4(v, v, v, v);
    }
}

// ------------------------------------------------------------------------------------------------
// Group triage (round 5).  Every OPEN tile is asked the two hierarchical questions once more per 64-micro-triangle group (its level-(N - 3) sub-triangles): the
// summed-area table (region_state_ex: settled / wholly open / undecided) and, for what the table leaves open, the curve-free-region test (region_curve.h:
// settled / edge-free).  Until round 4 ONE wave of the tile's workgroup did that inside the persistent classify_tiles launch while the other three waited at
// a barrier, and the test's registers (RcShape, RcTex, the cell loop) were live across the whole kernel -- 160 bytes of scratch per lane.  Here it is a dense
// pass of its own: lane = (open tile, group), no barriers, no LDS; the verdict bytes land in the tile's record, which the persistent workgroup reads anyway.
// ------------------------------------------------------------------------------------------------
// Chunks (round 5).  The groups that stay open are a quarter of a tile's 64 on average, so a workgroup of the persistent launch would spend its fixed per-tile
// work (record, texel window, five barriers, tables) on ~20 groups.  triage_groups therefore goes over the queue in windows of kChunkWindow adjacent records
// (tiles of one work item are adjacent: triage_tiles appends them wave by wave) and joins records of the same item whose open groups fit into 64 slots:
// the first becomes the chunk's HEAD -- bits 12..18 of its word [0].y say which of the following records belong to it, its rectangle becomes the union --
// the others are marked kTileDead like the tiles without any open group (whoever pops them has nothing to do) and carry their first slot in those bits.
constexpr uint32_t kChunkWindow = 8, kChunkMaskShift = 12, kChunkMaskBits = 0x7Fu;
#ifndef OMMX_TRIAGE_WAVES
#define OMMX_TRIAGE_WAVES 5   // (96 VGPRs, no scratch; 4 / 5 waves: 0.85 -> 0.75 ms at c2; 6 waves spill 60 bytes)
#endif
template <bool FP32, class MD, int TILE>
__global__ __launch_bounds__(256, OMMX_TRIAGE_WAVES) void triage_groups(ClassifyParams P, ItemArrays A, uint4* __restrict__ queue, const uint32_t* __restrict__ queueCtl, uint32_t numSections, uint32_t window)
{
    constexpr uint32_t GROUPS = (uint32_t)TILE / 64u, PER_BLOCK = 256u / GROUPS;   // 64 groups: a wave per tile; 16 groups: four tiles per wave
    // records a wave takes in a row: kChunkWindow for 4096-tiles -- 1 when every item is ONE tile (level 6 only; 1024-tiles always): nothing to join there, and a wave
    // that walks 8 records one after the other is 8 x the latency of a small bake's triage (configs[1]: 60 -> 15 us)
    const uint32_t SPAN = GROUPS == 64u ? window : 1u;
    const uint32_t g = threadIdx.x % GROUPS, slot = threadIdx.x / GROUPS;
    const bool coarse = P.useCoarse != 0;
    const bool fastFine = P.filterLinear != 0 && P.mipCount == 1 && !P.noFine && P.altKernel == 0;
    const bool curveOn = (OMMX_RC_LEVELS & 4) && region_curve_applies(P);
    const uint32_t bits = (uint32_t)P.format, groupBytes = 8u * bits;   // 64 micro-triangles x bits
    for (uint32_t sec = 0; sec < numSections; ++sec) {
        const uint32_t base = queueCtl[kSecBases + sec], tail = queueCtl[kSecTails + sec];
        for (uint32_t w0 = (blockIdx.x * PER_BLOCK + slot) * SPAN; w0 < tail; w0 += gridDim.x * PER_BLOCK * SPAN) {
            // the chunk being built (wave-uniform; 4096-tiles only): head record, its item, open groups so far, follower mask, union rectangle
            uint32_t hRec = 0xFFFFFFFFu, hItem = 0, hOpen = 0, hMask = 0, hY = 0, hSx = 0, hSy = 0, hEx = 0, hEy = 0; bool hOk = false;
            auto flush = [&]() {
                if (hRec != 0xFFFFFFFFu && hMask != 0u && g == 0u) {
                    uint32_t* hw = (uint32_t*)(queue + (size_t)kTileRecordWords * (base + hRec));
                    hw[0] = (hw[0] & 0x7FFFFFFFu) | (hOk ? 0x80000000u : 0u); hw[1] = hY | (hMask << kChunkMaskShift); hw[2] = hSx | (hSy << 16); hw[3] = hEx | (hEy << 16);
                }
                hRec = 0xFFFFFFFFu;
            };
            for (uint32_t j = 0; j < SPAN; ++j) {
            const uint32_t r = w0 + j;
            if (r >= tail) break;   // (uniform over the lanes that share the records)
            uint4* rec = queue + (size_t)kTileRecordWords * (base + r);
            const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            float uv[6] = { __uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x), __uint_as_float(r2.y) };
            const uint32_t level = (r0.y >> 24) & 0x7Fu, first = (r0.y & 0xFFFu) * (GROUPS);   // first group of the tile in the item's level-(N - 3) enumeration
            const float maxAbs = item_max_abs(uv);
            const bool degenerate = ((r0.x >> 30) & 1u) != 0u;
            const bool fast = fastFine && !degenerate && maxAbs <= 16384.f && (r0.y >> 31) == 0u;   // (= classify_tiles' uFast: only then can it use the edge-free verdict)
            const MicroTri gsub = micro_triangle(uv, first + g, level - 3u);
            int gs = coarse ? region_state_ex<MD>(P, gsub, maxAbs, no_window()) : kRegionUnknown;
            if (gs < 0 && curveOn && !degenerate) {
                const DevMip& m0 = P.mips[0];
                const RcShape shape = rc_shape(uv, m0.fw, m0.fh, m0.w, m0.h, level);
                const int cs = region_curve_state_impl(rc_tex<MD>(P, FP32), shape, gsub.lo.x, gsub.lo.y, gsub.hi.x, gsub.hi.y, maxAbs);
                gs = (cs >= 0 || (cs <= -kRegionEdgeFreeBase && fast)) ? cs : gs;
            }
            ((uint8_t*)(rec + 3))[g] = (uint8_t)group_verdict_byte(gs);
            // A settled group is final right here: its constant packed states go to the tile's block, its state into the item's mask / known count (the
            // persistent workgroup packs the open groups only).  A tile whose groups are ALL settled -- 14 % of the open tiles of the metric configuration --
            // is marked dead in its record: the persistent launch passes it without touching the window, a barrier or the states.
            if (gs >= 0) {
                uint8_t* dst = (uint8_t*)(((unsigned long long)r2.w << 32) | r2.z) + (size_t)g * groupBytes;
                uint32_t v = 0;
                for (uint32_t b = 0; b < 32u; b += bits) v |= (uint32_t)gs << b;
                if (bits == 2u) *(uint4*)dst = make_uint4(v, v, v, v); else *(uint2*)dst = make_uint2(v, v);
            }
            // per tile (= GROUPS consecutive lanes of the wave): which states its settled groups have, how many of them are known (T / O)
            const uint32_t lane = threadIdx.x & 63u, sh = lane & ~(GROUPS - 1u);   // (GROUPS == 64: the whole wave, sh == 0)
            const unsigned long long tileLanes = GROUPS == 64u ? ~0ull : (((1ull << GROUPS) - 1ull) << sh);
            uint32_t mask = 0;
            #pragma unroll
            for (int st = 0; st < 4; ++st) if (__ballot(gs == st) & tileLanes) mask |= 1u << st;
            const uint32_t known = (uint32_t)__popcll(__ballot(gs == 0 || gs == 1) & tileLanes) * 64u;
            const uint32_t open = (uint32_t)__popcll(__ballot(gs < 0) & tileLanes);
            const uint32_t item = r0.x & 0x3FFFFFFFu;
            // join: this record follows the chunk's head when it is of the same item and its open groups still fit
            bool follower = false; uint32_t followerStart = 0;
            if (GROUPS == 64u && open != 0u) {
                const uint32_t ry = uniform_u32(r0.y), rx = uniform_u32(r0.x), rz = uniform_u32(r0.z), rw = uniform_u32(r0.w), ritem = rx & 0x3FFFFFFFu;
                if (hRec != 0xFFFFFFFFu && ritem == hItem && hOpen + open <= 64u) {
                    follower = true; followerStart = hOpen;   // (its open groups take the slots behind those of the members before it)
                    hMask |= 1u << (r - hRec - 1u); hOpen += open; hOk = hOk && (rx >> 31) != 0u;
                    const uint32_t sx = rz & 0xFFFFu, sy = rz >> 16, ex = rw & 0xFFFFu, ey = rw >> 16;
                    hSx = sx < hSx ? sx : hSx; hSy = sy < hSy ? sy : hSy; hEx = ex > hEx ? ex : hEx; hEy = ey > hEy ? ey : hEy;
                } else {
                    flush();
                    hRec = r; hItem = ritem; hOpen = open; hMask = 0; hY = ry; hOk = (rx >> 31) != 0u; hSx = rz & 0xFFFFu; hSy = rz >> 16; hEx = rw & 0xFFFFu; hEy = rw >> 16;
                }
            }
            if (g == 0u) {
                if (mask) atomicOr(&A.stateMask[item], mask);
                if (P.wantKnownCount && known) atomicAdd(&A.knownCount[item], known);
                if (open == 0u || follower) ((uint32_t*)rec)[1] = r0.y | kTileDead | (followerStart << kChunkMaskShift);   // (bits 12..18 of a follower: its first slot)
            }
            }
            flush();
        }
    }
}

constexpr int WIN = 32; // largest LDS texel window edge

// SLICED: 4^level >= TILE, the tile is a slice of ONE work item (block-uniform item data, LDS texel/SAT window).
// !SLICED: the tile holds TILE / 4^level whole items.
// Waves per SIMD of the persistent launch, measured again every time the kernel's instruction mix moved: round 1 asked for 7 (72 VGPRs, one spilled dword:
// 52.3 -> 50.0 ms against 6), round 2 kept 7 (36.8 / 33.1 / 40.9 ms at 6 / 7 / 8); since curve_excluded() took the edge tests out of 88 % of the
// micro-triangles 6 waves (80 VGPRs, 25 instead of 34 spilled, 2.8 instead of 7.1 GB of HBM traffic per launch) win: 26.4 / 24.5 / 25.1 / 29.9 ms at
// 5 / 6 / 7 / 8 on the metric configuration, 131.0 / 122.5 / 127.2 ms on configs[4].
#ifndef OMMX_CLASSIFY_WAVES
#define OMMX_CLASSIFY_WAVES 6
#endif
// TILE micro-triangles per workgroup: 4096 for levels >= 6, 1024 below (a level-5 item is exactly one 1024-tile)
constexpr uint32_t kDeferredState = 0xFEu;   // s_state code of a micro-triangle that classify_generic() will classify
// The kernel's first argument read again from the kernarg segment (scalar loads through the scalar cache) behind an empty asm the optimizer cannot look through:
// a phase that starts with it re-loads the few fields it uses instead of the kernel holding every field any phase uses in SGPRs from its first line to its
// last -- which do not exist: the allocator parks them in VGPR lanes (74 in the instantiation that defers the texel walks) and every use is a v_readlane.
#ifndef OMMX_FRESH_PARAMS
#define OMMX_FRESH_PARAMS 1
#endif
__device__ __forceinline__ const ClassifyParams* fresh_kernarg_params()
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();   // (a pointer into the constant address space; ClassifyParams is the first argument of the kernels that call this)
    asm volatile("" : "+s"(p));
    return (const ClassifyParams*)p;
}
template <bool FP32, bool SLICED, int TILE, class MD, bool DEFER = false>
__global__ __launch_bounds__(BLOCK, OMMX_CLASSIFY_WAVES) void classify_tiles(ClassifyParams Pk, ItemArrays A, const uint32_t* __restrict__ itemIds,
                                                        uint32_t numItems, uint32_t levelArg, uint64_t numTiles,
                                                        const uint4* __restrict__ tileQueue, uint32_t* __restrict__ queueCtl, uint32_t numSections, GenericQueue G)
{
    __shared__ unsigned long long s_gbase;       // DEFER: first entry of this tile's reservation in the generic queue
    __shared__ __attribute__((aligned(16))) uint8_t s_state[TILE];
    __shared__ uint16_t s_queue[TILE];
    __shared__ int      s_group[TILE / GROUP];
    // sliced tiles: the 64-group SLOTS 0 .. s_gcount-1 of the workgroup hold the OPEN groups of the chunk's tiles (triage_groups settled the others), slot s =
    // group s_gid[s] of the work item's level-(N - 3) enumeration; s_group[s] = its verdict (< 0), s_gdec[s] its bird-curve decode
    __shared__ uint32_t s_gid[SLICED ? TILE / GROUP : 1];
    __shared__ uint16_t s_olist[TILE / GROUP];   // the slots that are all-open (taken as whole waves by the single-texel pass)
    __shared__ uint32_t s_gcount, s_ocount;
    __shared__ uint32_t s_qcount, s_ecount;      // queue fill counts (front / back of s_queue)
    __shared__ uint32_t s_mask, s_known;
    __shared__ uint32_t s_pending, s_fine;       // single-texel pass: micro-triangles left for the generic pass / level-line statistic
    __shared__ uint32_t s_next;                  // sliced: next tile-queue position of this (persistent) workgroup
    // tid is re-"defined" (an empty asm the optimizer cannot look through) at the top of every tile and phase: otherwise LICM hoists every cheap value derived
    // from it -- LDS addresses, lane masks, wave indices, a dozen of them -- out of the persistent tile loop and keeps them alive across all phases, i.e. in
    // scratch (round 4: 18 scratch stores in front of the loop, reloads in every phase).  Recomputing them costs one or two instructions each.
    uint32_t tid = threadIdx.x;
#if OMMX_FRESH_PARAMS
    // (only the instantiation that defers the texel walks: there it takes the last 20 bytes of scratch away, 4.65 -> 4.34 ms on the asset-shaped bake; the
    //  hot instantiation of the metric configuration has no scratch to lose and runs 7.41 -> 7.50 ms with the reloads)
    constexpr bool kFresh = DEFER || OMMX_FRESH_PARAMS == 2;   // (2: A/B builds, every instantiation)
    const ClassifyParams* Pp = kFresh ? fresh_kernarg_params() : &Pk;
#define P (*Pp)
#define OMMX_FRESH_P() do { if (kFresh) Pp = fresh_kernarg_params(); } while (0)
#else
#define P Pk
#define OMMX_FRESH_P() ((void)0)
#endif
#ifndef OMMX_NO_TID_LAUNDER
#define OMMX_FRESH_TID() do { asm volatile("" : "+v"(tid)); OMMX_FRESH_P(); } while (0)
#else
#define OMMX_FRESH_TID() OMMX_FRESH_P()
#endif
    // DEFER: hand `n` queued micro-triangles (s_queue[0 .. n)) to the generic queue; false = no room, the caller walks them itself.  Block-uniform.
    auto defer_generic = [&](uint32_t n, uint32_t itemWord, uint32_t level) -> bool {
        if (!DEFER || !SLICED || n == 0u) return false;
        OMMX_FRESH_TID();
        if (tid == 0) {
            // One atomicAdd per tile and no way back: the count only grows (an add that is taken back lets a concurrent workgroup fail spuriously or succeed
            // at a base the final count no longer covers; a compare-and-swap loop on one word collapses under 1536 workgroups -- measured: 7 us per tile).  A
            // reservation that does not fit is given up, and the part of it that lies inside the queue is filled with null entries below.
            s_gbase = atomicAdd(G.count, (unsigned long long)n);   // (64 bits: the reservations of a large bake that overflows the queue must not wrap)
        }
        __syncthreads();
        const unsigned long long gb = s_gbase;
        if (gb + n > (unsigned long long)G.capacity) {
            for (unsigned long long q = gb + tid; q < (unsigned long long)G.capacity; q += BLOCK) G.entries[q] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);   // (classify_generic skips these)
            return false;
        }
        for (uint32_t q = tid; q < n; q += BLOCK) {
            const uint32_t i = s_queue[q];
            G.entries[gb + q] = make_uint2(itemWord & 0x7FFFFFFFu, (level << 24) | (s_gid[i >> 6] * 64u + (i & 63u)));
            s_state[i] = (uint8_t)kDeferredState;
        }
        return true;
    };
    __shared__ uint32_t s_sec, s_nsec, s_retired; // sliced: section of the current tile / of the next one, tiles of the current section this workgroup has finished
    __shared__ uint32_t s_gdec[SLICED ? TILE / GROUP : 1];   // sliced: bird-curve decode of each 64-group of the tile (classify_device.h: BirdGroup)
    __shared__ uint8_t  s_btab[SLICED ? 256 : 1];            // ... and the 4 x 64 table of the low decode bits, filled once per workgroup
    __shared__ uint32_t s_shape[4];                          // sliced: rc_shape() of the chunk's work item (rhoX, rhoY, ok & fat): fine_single_texel's corner test
    __shared__ float    s_wtex[SLICED ? WIN * WIN : 1];
    __shared__ uint32_t s_wsat[SLICED ? (WIN + 1) * (WIN + 1) : 1];
    constexpr uint32_t TILE_LOG4 = TILE == 4096 ? 6u : 5u; // the tile is the level-(N - TILE_LOG4) sub-triangle of its item
    // SLICED: persistent workgroups drain the queue of open tiles that triage_tiles filled (all levels >= log4 TILE in one launch);
    // the position of the NEXT tile is fetched while the current one is being classified.
    // The queue is cut into sections (TileSections; queueCtl: SectionCtl words per section), drained one after the other by the SAME launch.  A streamed
    // bake (ommCpuBake) has a section per range of work items: a workgroup that leaves a section adds the tiles it finished there to the section's `done`
    // count behind a device-scope release, and `done == tail` tells the placement stream (tail_kernels.hip: stream_wait_section) that every block of
    // that range is in memory -- the finished blocks of one range travel to the host while the same launch classifies the next ranges.
    uint32_t qpos = 0;
    // next record of this workgroup (tid 0): the head of the first section that still has one
    auto next_record = [&](uint32_t& sec) -> uint32_t {
        for (; sec < numSections; ++sec) {
            const uint32_t p = atomicAdd(queueCtl + kSecHeads + sec, 1u);
            if (p < queueCtl[kSecTails + sec]) return queueCtl[kSecBases + sec] + p;
        }
        return 0xFFFFFFFFu;
    };
    if (SLICED) {
        s_btab[tid] = (uint8_t)bird_table_entry(tid >> 6, tid & 63u);   // (BLOCK == 256 entries)
        if (tid == 0) { uint32_t sec = 0; s_next = next_record(sec); s_sec = sec; s_nsec = sec; s_retired = 0; s_ocount = 0; }
        __syncthreads();
        qpos = uniform_u32(s_next);
    }
  for (;;) {
    OMMX_FRESH_TID();
    uint32_t level = levelArg, tile = 0;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u), rec1 = rec, rec2 = rec;
    uint32_t nextPos = 0;
    if (SLICED) {
        if (qpos == 0xFFFFFFFFu) return;
        const uint4* rp = tileQueue + (size_t)kTileRecordWords * qpos;
        rec = rp[0]; rec1 = rp[1]; rec2 = rp[2];   // three independent loads: one round trip for item, rectangle, UVs and output address
        rec.x = uniform_u32(rec.x); rec.y = uniform_u32(rec.y); rec.z = uniform_u32(rec.z); rec.w = uniform_u32(rec.w);
        if (tid == 0) { uint32_t sec = s_nsec; nextPos = next_record(sec); s_nsec = sec; }   // consumed at the end of this tile
        level = (rec.y >> 24) & 0x7Fu;
        // a sliced tile implies 4^level >= TILE; without the hint clang hoists micro_triangle()'s level-0 branch (three loop-invariant
        // vertices) out of the phase-1/2 loops and keeps them in VGPRs for the whole kernel
        __builtin_assume(level >= TILE_LOG4);
    } else {
        // 2-D grid: the AQL dispatch packet counts work-items per dimension in 32 bits, so x alone tops out at 2^24 tiles
        const uint64_t tile64 = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
        if (tile64 >= numTiles) return;
        tile = (uint32_t)tile64;
    }
    const uint32_t M = 1u << (2 * level);
    const uint32_t itemsPerTile = SLICED ? 1u : TILE / M;
    const uint32_t firstItem = SLICED ? 0u : tile * itemsPerTile;
    const uint32_t headTile = SLICED ? (rec.y & 0xFFFu) : 0u;        // tile (in its item) of the chunk's head record
    const uint32_t followers = SLICED ? (rec.y >> kChunkMaskShift) & kChunkMaskBits : 0u;   // which of the next records belong to the chunk (triage_groups)
    const bool dead = SLICED && (rec.y & kTileDead) != 0u;          // (block-uniform) nothing to do here: no open group, or a follower of an earlier head
    uint32_t itemsHere = SLICED ? 1u : numItems - firstItem;
    if (itemsHere > itemsPerTile) itemsHere = itemsPerTile;
    const uint32_t count = SLICED ? (uint32_t)TILE : itemsHere * M;   // micro-triangles in this tile
    const bool coarse = P.useCoarse != 0;

    // block-uniform item data of a sliced tile
    uint32_t uItem = 0; float uUv[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }; float uMaxAbs = 0.f; bool uDegenerate = false, uFast = false;
    TexWindow W = no_window();
    if (tid == 0) { s_qcount = 0; s_mask = 0; s_known = 0; s_pending = 0; s_fine = 0; }   // (s_ocount: see the end of the tile -- phase 0 adds to it before its first barrier)
    // the single-texel fast pass (fine_single_texel) covers Linear filtering of one mip on non-degenerate items; everything else is generic
    const bool fastFine = SLICED && P.filterLinear != 0 && P.mipCount == 1 && !P.noFine && P.altKernel == 0;
    // micro-triangle i (0 .. TILE) of a sliced tile through the split bird decode: group word + table entry instead of the full decode
    auto tile_micro_triangle = [&](uint32_t i) {
        BirdGroup bg; bg.word = s_gdec[SLICED ? i >> 6 : 0u];
        return micro_triangle_grouped(uUv, bg, (uint32_t)s_btab[SLICED ? ((bg.word >> 24) & 3u) * 64u + (i & 63u) : 0u], level);
    };
    if (SLICED) uItem = rec.x & 0x3FFFFFFFu;
    if (!dead) {
    if (SLICED) {
        uUv[0] = uniform_f32(__uint_as_float(rec1.x)); uUv[1] = uniform_f32(__uint_as_float(rec1.y)); uUv[2] = uniform_f32(__uint_as_float(rec1.z));
        uUv[3] = uniform_f32(__uint_as_float(rec1.w)); uUv[4] = uniform_f32(__uint_as_float(rec2.x)); uUv[5] = uniform_f32(__uint_as_float(rec2.y));
        uMaxAbs = uniform_f32(item_max_abs(uUv));
        uDegenerate = ((rec.x >> 30) & 1u) != 0u;
        // fine_single_texel's FINITE precondition: |uv| <= 16384 bounds every pixel coordinate by 2^30 (size <= 65536) and excludes NaN;
        // items outside it (and degenerate ones) take the generic path
        uFast = fastFine && !uDegenerate && uMaxAbs <= 16384.f && (rec.y >> 31) == 0u;   // (bit 31: no micro-triangle of the item fits in one texel, triage_tiles)
        {
            OMMX_FRESH_TID();
            // ---- phase 0b: LDS window = every texel / SAT entry this tile can touch ----
            // (the tile's texel rectangle was computed by triage_tiles: region_rect of its sub-triangle)
            TexRect r; r.sx = (int)(rec.z & 0xFFFFu); r.sy = (int)(rec.z >> 16); r.ex = (int)(rec.w & 0xFFFFu); r.ey = (int)(rec.w >> 16); r.ok = (rec.x >> 31) != 0u;
            const int ww = r.ex - r.sx + 1, wh = r.ey - r.sy + 1;
            bool windowOk = false;
#ifdef OMMX_EXP_NO_WINDOW
            if (false) {
#else
            if (r.ok && ww <= WIN && wh <= WIN) {
#endif
                const DevMip& m0 = P.mips[0];
                // 64 x 4 thread grid over the window (no integer division by the run-time width): column = lane, one row per wave and pass
                const int cx = (int)(tid & 63u);
                for (int cy = (int)(tid >> 6); cy <= wh; cy += 4) {
                    const int x = r.sx + cx, y = r.sy + cy;
                    if (cx < ww && cy < wh) {
                        const size_t idx = (size_t)x + (size_t)y * (size_t)m0.w;
                        s_wtex[cx + cy * ww] = FP32 ? ((const float*)m0.texels)[idx] : (float)((const uint8_t*)m0.texels)[idx] * (1.f / 255.f);
                    }
                    if (coarse && m0.sat && cx <= ww) // SAT entry (x-1, y-1); row / column -1 of the table is zero
                        s_wsat[cx + cy * (ww + 1)] = (x >= 1 && y >= 1) ? m0.sat[(size_t)(x - 1) + (size_t)(y - 1) * (size_t)m0.w] : 0u;
                }
                windowOk = true;
            }
            if (tid == 0 && uFast) {   // (read by phase 2a, behind the barriers of this phase; the corner bound itself was decided by triage_tiles: kTileFat)
                const DevMip& m0 = P.mips[0];
                float ux, uy, rhoX, rhoY;
                rc_rho(uUv, m0.fw, m0.fh, &ux, &uy, &rhoX, &rhoY);
                s_shape[0] = __float_as_uint(rhoX); s_shape[1] = __float_as_uint(rhoY); s_shape[2] = (rec.y & kTileFat) ? 1u : 0u;
            }
            // ---- phase 0c: the slot tables.  Wave w takes the chunk's member records w and w + 4 (member 0 = the head, then the records its follower mask
            //      names): lane = group of the member's tile, verdict byte from the record; the member's open groups take consecutive slots from the start
            //      that triage_groups left in the follower's record ----
            constexpr uint32_t GPT = (uint32_t)TILE / GROUP;   // groups per tile
            const uint32_t lane = tid & 63u, wv = tid >> 6;
            #pragma unroll
            for (uint32_t h = 0; h < 2u; ++h) {
                const uint32_t mi = wv + 4u * h;   // member index
                // record of member mi: the head itself, or head + 1 + (position of the mi-th set bit of the follower mask)
                uint32_t fm = followers, ofs = 0; bool have = mi == 0u;
                for (uint32_t k = 1; k <= mi && fm; ++k) { const uint32_t bit = (uint32_t)__ffs((int)fm) - 1u; fm &= fm - 1u; if (k == mi) { ofs = bit + 1u; have = true; } }
                if (!have) continue;   // (wave-uniform)
                const uint4* mrec = tileQueue + (size_t)kTileRecordWords * (qpos + ofs);
                const uint32_t my = uniform_u32(((const uint32_t*)mrec)[1]);
                const uint32_t mtile = my & 0xFFFu, start = mi == 0u ? 0u : (my >> kChunkMaskShift) & kChunkMaskBits;
                const int gs = lane < GPT ? group_verdict((uint32_t)((const uint8_t*)(mrec + 3))[lane]) : 0;
                const unsigned long long open = __ballot(lane < GPT && gs < 0);
                const bool mine = ((open >> lane) & 1ull) != 0ull;
                const uint32_t sl = start + (uint32_t)__popcll(open & ((1ull << lane) - 1ull));
                if (mine) {
                    const uint32_t gid = mtile * GPT + lane;
                    s_group[sl] = gs; s_gid[sl] = gid; s_gdec[sl] = bird_group(gid, level - 3).word;
                }
                // the all-open slots, compacted in any order (the single-texel pass takes them as whole waves)
                const unsigned long long ao = __ballot(mine && gs == kRegionAllOpen);
                if (ao) {
                    uint32_t ob = 0;
                    if (lane == 0u) ob = atomicAdd(&s_ocount, (uint32_t)__popcll(ao));
                    ob = __shfl(ob, 0);
                    if (mine && gs == kRegionAllOpen) s_olist[ob + (uint32_t)__popcll(ao & ((1ull << lane) - 1ull))] = (uint16_t)sl;
                }
                if (fm == 0u && lane == 0u) s_gcount = start + (uint32_t)__popcll(open);   // (the last member: no follower behind it)
            }
            __syncthreads();
            if (windowOk) { const DevMip& m0 = P.mips[0]; W.tex = (lds_float*)s_wtex; W.sat = (lds_u32*)s_wsat; W.base = m0.texels; W.sx = r.sx; W.sy = r.sy; W.w = ww; W.h = wh; } // (SAT part is only read when coarse is on)
        }
    } else {
        __syncthreads();
        if (tid < (uint32_t)(TILE / GROUP)) {
            int gs = -1;
            const uint32_t i0 = tid * GROUP;
            if (coarse && level >= 3 && i0 < count) { // 64 consecutive micro-triangles = one level-(N-3) sub-triangle of one item
                const float* uvp = A.uv + 6ull * itemIds[firstItem + (i0 >> (2 * level))];
                gs = region_state_ex<MD>(P, micro_triangle(uvp, (i0 & (M - 1u)) >> 6, level - 3), item_max_abs(uvp), W);
            }
            s_group[tid] = gs;
        }
        __syncthreads();
    }
    {
        OMMX_FRESH_TID();
        // ---- phase 1: per-micro-triangle coarse test in the unsettled groups ----
        // one wave = one 64-group; gs (wave-uniform) is < 0 here: kRegionAllOpen or kRegionUnknown
        auto phase1_group = [&](uint32_t i, int gs) {
            bool unresolved = false;
            if (gs <= -kRegionEdgeFreeBase) {
                // the group lies in one cell whose level curve cannot reach it (region_curve.h), but the work item is too thin / too small against the rounding of
                // its vertices for the corner bound: every micro-triangle has the state of that side unless PointInTriangle puts one of the cell's wrong-side corners
                // inside it -- four cross-product tests per lane instead of the whole single-texel pass; the (rare) exceptions are queued for that pass
                const uint32_t code = (uint32_t)(-gs - kRegionEdgeFreeBase);
                if (i < count) {
                    const DevMip& m0 = P.mips[0];
                    const MicroTri t = tile_micro_triangle(i);
                    const float pfx = __builtin_floorf(t.p0.x * m0.fw - 0.5f) + 0.5f, pfy = __builtin_floorf(t.p0.y * m0.fh - 0.5f) + 0.5f;   // the cell of the group = of every vertex in it
                    const float ipx = pfx * m0.rw, ipy = pfy * m0.rh;
                    const bool in0 = point_in_triangle_flat(t, ipx, ipy), in1 = point_in_triangle_flat(t, ipx, ipy + m0.rh);
                    const bool in2 = point_in_triangle_flat(t, ipx + m0.rw, ipy + m0.rh), in3 = point_in_triangle_flat(t, ipx + m0.rw, ipy);
                    const uint32_t in = (in0 ? 1u : 0u) | (in1 ? 2u : 0u) | (in2 ? 4u : 0u) | (in3 ? 8u : 0u);
                    unresolved = (in & code & 15u) != 0u;
                    s_state[i] = (uint8_t)((code & 16u) ? P.stateGT : P.stateLE);
                }
            } else
            if (gs == kRegionAllOpen) { // the whole group is unresolved by construction: no per-micro-triangle SAT test
                if (uFast) return;   // phase 2a takes the group as a whole, it needs no queue entries
                unresolved = i < count;   // (phase 2 writes the state of every queued micro-triangle)
                if (P.noFine && i < count) s_state[i] = 3;
            } else
            if (i < count) {
                int st = -1;
                if (coarse) {
                    // (items under the single-texel pass's FINITE precondition take the straight-line form of the same test: 28.2 -> 27.1 ms)
                    if (SLICED) st = uFast ? coarse_state_finite<MD>(P, tile_micro_triangle(i), W) : coarse_state<MD>(P, tile_micro_triangle(i), W);
                    else st = coarse_state<MD>(P, micro_triangle(A.uv + 6ull * itemIds[firstItem + (i >> (2 * level))], i & (M - 1u), level), W);
                }
                // the reference's fine pass re-classifies everything still "UnknownOpaque" (bake_cpu_impl.cpp:861)
                unresolved = (st < 0) || (st == 3) || !P.filterLinear;
                s_state[i] = (uint8_t)(st < 0 ? 3 : st);
            }
            if (P.noFine) unresolved = false;   // (DisableFineClassification: nothing is queued, unresolved micro-triangles stay UnknownOpaque)
            const unsigned long long vote = __ballot(unresolved);
            if (vote) {
                const uint32_t lane = tid & 63u;
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_qcount, (uint32_t)__popcll(vote));
                wbase = __shfl(wbase, 0);
                if (unresolved) s_queue[wbase + __popcll(vote & ((1ull << lane) - 1ull))] = (uint16_t)i;
            }
        };
        if (SLICED) { // every slot holds an open group
            const uint32_t gcount = s_gcount;
            for (uint32_t g = tid >> 6; g < gcount; g += BLOCK / 64) phase1_group(g * 64u + (tid & 63u), s_group[g]);
        } else {
            for (uint32_t i = tid; i < ((count + 63u) & ~63u); i += BLOCK) {
                const int gs = s_group[i >> 6];      // wave-uniform: a wave is exactly one group
                if (gs >= 0) { if (i < count) s_state[i] = (uint8_t)gs; continue; }
                phase1_group(i, gs);
            }
        }
        __syncthreads();

        // ---- phase 2: fine, dense over the queue ----
        OMMX_FRESH_TID();
        const uint32_t qn = s_qcount;
        if (tid == 0 && qn) atomicAdd(A.fineCount + (size_t)((blockIdx.x + 7u * blockIdx.y) & (kFineSlots - 1)) * kFineStride, (unsigned long long)qn);
        if (uFast) {
            // ---- phase 2a: straight-line single-texel pass.  Work units of 64 lanes: the all-open groups (one wave = one group, no queue entries), then
            //      the queued micro-triangles of the other open groups.  The three edge tests are replaced by curve_excluded() (classify_device.h), which
            //      settles 88 % of the micro-triangles; what is left is compacted into s_queue: micro-triangles outside the single-texel pattern (0xFF)
            //      from the front, for the generic pass 2c; those that need their edge tests (kNeedsEdges) from the back, for pass 2b ----
            const uint32_t gcount = s_gcount, ocount = s_ocount;
            uint32_t qn2 = 0, en = 0, pend = 0;
            const uint32_t units = ocount + ((qn + 63u) >> 6);
            for (uint32_t k = tid >> 6; k < units; k += BLOCK / 64) {   // (k is wave-uniform)
                uint32_t i; bool live = true;
                if (k < ocount) i = (uint32_t)s_olist[k] * 64u + (tid & 63u);
                else { const uint32_t q = (k - ocount) * 64u + (tid & 63u); live = q < qn; i = live ? (uint32_t)s_queue[q] : 0u; }
                if (live) {
                    const int st = fine_single_texel<FP32, MD>(P, tile_micro_triangle(i), W, (const lds_u32*)s_shape);   // state | kNeedsEdges + hints | -1
                    s_state[i] = (uint8_t)(st < 0 ? 0xFF : st);
                    pend |= st < 0 ? 1u : ((st & kNeedsEdges) ? 2u : 0u);
                }
            }
            if (tid == 0 && ocount) s_fine = ocount * 64u;
            const unsigned long long anyGeneric = __ballot((pend & 1u) != 0), anyEdges = __ballot((pend & 2u) != 0);
            if ((tid & 63u) == 0 && (anyGeneric | anyEdges)) atomicOr(&s_pending, (anyGeneric ? 1u : 0u) | (anyEdges ? 2u : 0u));
            __syncthreads();
            OMMX_FRESH_TID();
            if (s_pending) {   // (block-uniform)
                if (tid == 0) { s_qcount = 0; s_ecount = 0; }
                __syncthreads();
                for (uint32_t k = tid >> 6; k < gcount; k += BLOCK / 64) {
                    const uint32_t i = k * 64u + (tid & 63u);
                    const uint32_t code = s_state[i];
                    const unsigned long long vg = __ballot(code == 0xFFu), ve = __ballot(code >= (uint32_t)kNeedsEdges && code != 0xFFu);
                    if (vg | ve) {
                        const uint32_t lane = tid & 63u;
                        uint32_t bg = 0, be = 0;
                        if (lane == 0) { if (vg) bg = atomicAdd(&s_qcount, (uint32_t)__popcll(vg)); if (ve) be = atomicAdd(&s_ecount, (uint32_t)__popcll(ve)); }
                        bg = __shfl(bg, 0); be = __shfl(be, 0);
                        if (code == 0xFFu) s_queue[bg + __popcll(vg & ((1ull << lane) - 1ull))] = (uint16_t)i;
                        else if (code >= (uint32_t)kNeedsEdges) s_queue[(uint32_t)TILE - 1u - (be + __popcll(ve & ((1ull << lane) - 1ull)))] = (uint16_t)i;
                    }
                }
                __syncthreads();
                qn2 = s_qcount; en = s_ecount;
            }
            // ---- phase 2b: the edge tests of the micro-triangles that curve_excluded() could not settle, densely ----
            OMMX_FRESH_TID();
            for (uint32_t q0 = 0; q0 < en; q0 += BLOCK) {
                const uint32_t q = q0 + tid;
                if (q < en) {
                    const uint32_t i = s_queue[(uint32_t)TILE - 1u - q];
                    s_state[i] = (uint8_t)single_texel_edges<FP32, MD>(P, tile_micro_triangle(i), W, (int)s_state[i]);
                }
            }
            // ---- phase 2c: the generic pass for whatever did not fit the single-texel pattern ----
            OMMX_FRESH_TID();
            if (tid == 0 && s_fine) atomicAdd(A.fineCount + (size_t)((blockIdx.x + 7u * blockIdx.y) & (kFineSlots - 1)) * kFineStride, (unsigned long long)s_fine);
            if (!defer_generic(qn2, rec.x, level))
            for (uint32_t q0 = 0; q0 < qn2; q0 += BLOCK) {
                const uint32_t q = q0 + tid;
                if (q < qn2) {
                    const uint32_t i = s_queue[q];
                    s_state[i] = (uint8_t)fine_state<FP32, MD>(P, tile_micro_triangle(i), uDegenerate, W);
                }
            }
        } else if (!defer_generic(qn, rec.x, level))
        for (uint32_t q0 = 0; q0 < qn; q0 += BLOCK) { // q0 is block-uniform (scalar loop counter): one VGPR less across the level-line pass
            const uint32_t q = q0 + tid;
            if (q < qn) {
                const uint32_t i = s_queue[q];
                const uint32_t u = SLICED ? s_gid[i >> 6] * 64u + (i & 63u) : (i & (M - 1u));
                if (SLICED) {
                    s_state[i] = (uint8_t)fine_state<FP32, MD>(P, micro_triangle(uUv, u, level), uDegenerate, W);
                } else {
                    const uint32_t item = itemIds[firstItem + (i >> (2 * level))];
                    s_state[i] = (uint8_t)fine_state<FP32, MD>(P, micro_triangle(A.uv + 6ull * item, u, level), A.degenerate[item] != 0, W);
                }
            }
        }
        __syncthreads();
    }
    }   // (!dead)

    // ---- phase 3: pack + per-item summary ----
    OMMX_FRESH_TID();
    const uint32_t bits = (uint32_t)P.format;          // 1 or 2 bits per micro-triangle
    const uint32_t perWord = 32u / bits;               // micro-triangles per 32-bit word
    if (SLICED) {
        uint32_t* dst = (uint32_t*)(((unsigned long long)uniform_u32(rec2.w) << 32) | uniform_u32(rec2.z));   // (from the tile record)
        uint32_t localMask = 0, localKnown = 0;   // (tiles settled as a whole never get here: triage_tiles wrote them)
        // the words of the slots (all open groups); the settled groups' words, their states and known counts are final since triage_groups
        const uint32_t wpgLog = bits == 2u ? 2u : 1u, openWords = dead ? 0u : s_gcount << wpgLog;   // 4 (4-state) or 2 (2-state) words per group
        for (uint32_t q = tid; q < openWords; q += BLOCK) {
            const uint32_t w = q;   // word q of the slots' states; it belongs to slot q >> wpgLog, i.e. to group s_gid[..] of the item
            uint32_t v = 0;
            if (!DEFER) {
                // 16 (4-state) or 32 (2-state) state bytes -> one word with 128-bit LDS reads and bit gathers instead of a loop over bytes; which states occur
                // and how many are known (T / O) from the packed word itself
                if (bits == 2u) {
                    const uint4 x = *(const uint4*)&s_state[w * 16u];
                    auto g4 = [](uint32_t d) { d &= 0x03030303u; return (d | (d >> 6) | (d >> 12) | (d >> 18)) & 0xFFu; };
                    v = g4(x.x) | (g4(x.y) << 8) | (g4(x.z) << 16) | (g4(x.w) << 24);
                    const uint32_t lo = v & 0x55555555u, hi = (v >> 1) & 0x55555555u;
                    localMask |= ((0x55555555u & ~lo & ~hi) ? 1u : 0u) | ((lo & ~hi) ? 2u : 0u) | ((hi & ~lo) ? 4u : 0u) | ((lo & hi) ? 8u : 0u);
                    localKnown += (uint32_t)__popc(0x55555555u & ~hi);
                } else {
                    const uint4 x = *(const uint4*)&s_state[w * 32u], y = *(const uint4*)&s_state[w * 32u + 16u];
                    auto g1 = [](uint32_t d) { d &= 0x01010101u; return (d | (d >> 7) | (d >> 14) | (d >> 21)) & 0xFu; };
                    v = g1(x.x) | (g1(x.y) << 4) | (g1(x.z) << 8) | (g1(x.w) << 12) | (g1(y.x) << 16) | (g1(y.y) << 20) | (g1(y.z) << 24) | (g1(y.w) << 28);
                    localMask |= (v != 0xFFFFFFFFu ? 1u : 0u) | (v != 0u ? 2u : 0u);
                    localKnown += 32u;
                }
            } else
            for (uint32_t k = 0; k < perWord; ++k) {
                uint32_t st = s_state[w * perWord + k];
                if (DEFER && st == kDeferredState) continue;   // (its bits stay 0 for classify_generic's atomicOr; mask and known count come from there too)
                v |= st << (k * bits);
                localMask |= 1u << st;
                localKnown += st < 2u;
            }
            dst[(int)((s_gid[q >> wpgLog] - headTile * ((uint32_t)TILE / GROUP)) << wpgLog) + (int)(q & ((1u << wpgLog) - 1u))] = v;   // (a follower's tile may lie in front of the head's: the offset is signed)
        }
        if (localMask) atomicOr(&s_mask, localMask);
        if (P.wantKnownCount && localKnown) atomicAdd(&s_known, localKnown);
        // leaving the section (its queue ran dry while this tile was classified; block-uniform): every wave waits until its own stores have been
        // acknowledged by the L2 (s_waitcnt vmcnt(0); a workgroup barrier alone does not: the waves of a workgroup share their CU's path to memory, so the
        // workgroup-scope release in front of it needs no wait), then ONE device-scope release by thread 0 writes the L2 back -- a write-back per wave
        // measured 0.8 ms more on the metric configuration
        if (s_nsec != s_sec) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
        if (tid == 0) { s_next = nextPos; s_ocount = 0; }   // next tile of this workgroup (requested at the top of the loop); published with the barrier below
                                                          // (s_ocount was last read in phase 2a; the next tile's phase 0 adds to it in front of its first barrier)
        __syncthreads();
        if (tid == 0) {
            if (s_mask) atomicOr(&A.stateMask[uItem], s_mask);
            if (P.wantKnownCount && s_known) atomicAdd(&A.knownCount[uItem], s_known);
            // leaving the section: every store of this workgroup's tiles in it is ordered before the count (the fences above, this thread's release)
            const uint32_t sec = s_sec, retired = s_retired + 1u;
            if (s_nsec != sec) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(queueCtl + kSecDone + sec, retired, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                s_sec = s_nsec; s_retired = 0;
            } else s_retired = retired;
        }
        // Every LDS array of this tile was last read before that barrier, so the next tile starts right here: no further synchronisation.
        qpos = uniform_u32(s_next);
        continue;
    } else if (M >= perWord) {
        const uint32_t words = count / perWord;
        for (uint32_t w = tid; w < words; w += BLOCK) {
            uint32_t v = 0, localMask = 0, localKnown = 0;
            for (uint32_t k = 0; k < perWord; ++k) {
                const uint32_t st = s_state[w * perWord + k];
                v |= st << (k * bits);
                localMask |= 1u << st;
                localKnown += st < 2u;
            }
            const uint32_t i0 = w * perWord;
            const uint32_t item = itemIds[firstItem + (i0 >> (2 * level))];
            *(uint32_t*)(A.states + A.stateOfs[item] + (size_t)((i0 & (M - 1u)) / perWord) * 4u) = v;
            if (M == perWord) { A.stateMask[item] = localMask; if (P.wantKnownCount) A.knownCount[item] = localKnown; }
            else { atomicOr(&A.stateMask[item], localMask); if (P.wantKnownCount) atomicAdd(&A.knownCount[item], localKnown); }
        }
    } else {
        // items smaller than one word (level 0/1, and level 2 in 2-state): one lane per item, byte stores
        for (uint32_t k = tid; k < itemsHere; k += BLOCK) {
            const uint32_t item = itemIds[firstItem + k];
            uint32_t v = 0, mask = 0, known = 0;
            for (uint32_t j = 0; j < M; ++j) {
                const uint32_t st = s_state[k * M + j];
                v |= st << (j * bits);
                mask |= 1u << st;
                known += st < 2u;
            }
            uint32_t nbytes = (M * bits) >> 3; if (nbytes < 1u) nbytes = 1u;
            uint8_t* dst = A.states + A.stateOfs[item];
            for (uint32_t bI = 0; bI < nbytes; ++bI) dst[bI] = (uint8_t)(v >> (8u * bI));
            A.stateMask[item] = mask;
            if (P.wantKnownCount) A.knownCount[item] = known;
        }
    }
    return;   // (!SLICED: one tile per workgroup)
  }
#undef P
#undef OMMX_FRESH_P
}


// ------------------------------------------------------------------------------------------------
// Deferred generic pass.  Micro-triangles that span several texels (asset-sized triangles) walk the texels of their raster box (conservative
// raster + level-line kernel / nearest vote per texel: fine_state).  Inside classify_tiles one lane walks one box in lockstep with 63 others:
// neighbouring boxes differ in size by orders of magnitude, two thirds of the walks end at their first mixed texel after a handful of visits
// and the rest visit every texel under the triangle, so a fifth of the lanes does useful work (profiles/r03_v3_cards_pmc.md).  Here the walks
// are queue entries and a lane that finishes one takes the next (generic_walks below); every texel of a box is visited at most
// once, in row-major order, with the serial loop's early exit: same state.  Mip chains, the alternative kernel and degenerate items take the
// serial fine_state().  The state is ORed into the packed word the persistent launch left 0; item mask / known count are folded per wave.
// (Round 3's form -- 12 visits one lane per walk, then eight lanes per unfinished walk -- took 36.9 ms where this one takes 28.8.)
// ------------------------------------------------------------------------------------------------
struct RasterBox { EdgeEq e0, e1, e2; int minx, miny, xend, yend; uint32_t w, cnt; };   // texels [minx, xend) x [miny, yend), row-major index k < cnt
struct TexelCursor { int x, y; };
__device__ __forceinline__ TexelCursor cursor_at(const RasterBox& B, uint32_t k) { TexelCursor c; c.x = B.minx + (int)(k % B.w); c.y = B.miny + (int)(k / B.w); return c; }
__device__ __forceinline__ bool cursor_live(const RasterBox& B, const TexelCursor& c) { return c.y < B.yend; }
__device__ __forceinline__ void cursor_step(const RasterBox& B, TexelCursor& c) { if (++c.x == B.xend) { c.x = B.minx; ++c.y; } }
__device__ __forceinline__ RasterBox raster_box(const DevMip& m, const MicroTri& t, float off)
{
    // same set-up as raster_micro_triangle (classify_device.h): winding, raster-space vertices, box
    const double ax = (double)(t.p2.x - t.p0.x), ay = (double)(t.p2.y - t.p0.y);
    const double bx = (double)(t.p1.x - t.p0.x), by = (double)(t.p1.y - t.p0.y);
    const bool ccw = (ax * by - bx * ay) < 0;
    V2 a = mk2(t.p0.x * m.fw + off, t.p0.y * m.fh + off);
    const V2 b = mk2(t.p1.x * m.fw + off, t.p1.y * m.fh + off);
    V2 c = mk2(t.p2.x * m.fw + off, t.p2.y * m.fh + off);
    if (!ccw) { V2 s = a; a = c; c = s; }
    const float lox = std_min(std_min(a.x, b.x), c.x), loy = std_min(std_min(a.y, b.y), c.y);
    const float hix = std_max(std_max(a.x, b.x), c.x), hiy = std_max(std_max(a.y, b.y), c.y);
    const int minx = cvt_trunc_x86(__builtin_floorf(lox)), miny = cvt_trunc_x86(__builtin_floorf(loy));
    const int maxx = cvt_trunc_x86(__builtin_ceilf(hix)), maxy = cvt_trunc_x86(__builtin_ceilf(hiy));
    RasterBox B; B.e0 = edge_eq(a, b); B.e1 = edge_eq(b, c); B.e2 = edge_eq(c, a); B.minx = minx; B.miny = miny;
    const long long w = (long long)maxx - (long long)minx, h = (long long)maxy - (long long)miny;
    const unsigned long long cnt64 = (w > 0 && h > 0) ? (unsigned long long)w * (unsigned long long)h : 0ull;
    // (boxes beyond 2^32 texels cannot be walked in any case -- the serial loop would not end either --: they count as empty)
    const bool ok = cnt64 != 0ull && cnt64 <= 0xFFFFFFF0ull;
    B.w = ok ? (uint32_t)w : 1u; B.cnt = ok ? (uint32_t)cnt64 : 0u; B.xend = ok ? maxx : minx + 1; B.yend = ok ? maxy : miny;
    return B;
}
// does texel k of the box (row-major) lie under the (conservative) triangle?  (the cheap half of a visit: a lane skips to its next covered texel in a loop
// of these, so that the expensive half below runs with every lane that still has one -- half the texels of a box are not under its triangle)
__device__ __forceinline__ bool texel_under(const RasterBox& B, const TexelCursor& c)
{
    const float sx = (float)c.x, sy = (float)c.y;
    return eval_cons(B.e0, sx, sy) < 0.f && eval_cons(B.e1, sx, sy) < 0.f && eval_cons(B.e2, sx, sy) < 0.f;
}
// a wave's classified entries: state ORed into the packed word the persistent launch left 0; item mask / known count with one atomic per item and wave
// (the entries of a wave mostly share their item)
// returns the number of live entries of the wave (wave-uniform; the caller adds them up for the pass's statistic)
__device__ __forceinline__ uint32_t generic_commit(const ClassifyParams& P, const ItemArrays& A, bool live, uint32_t item, uint32_t index, int state)
{
    const uint32_t lane = threadIdx.x & 63u, bits = (uint32_t)P.format;
    if (live) {
        const uint64_t bit = (uint64_t)index * bits;
        atomicOr((uint32_t*)(A.states + A.stateOfs[item]) + (bit >> 5), (uint32_t)state << (uint32_t)(bit & 31u));
    }
    unsigned long long todo = __ballot(live);
    const uint32_t committed = (uint32_t)__popcll(todo);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t it0 = (uint32_t)__shfl((int)item, leader);
        const unsigned long long same = __ballot(live && item == it0) & todo;
        todo &= ~same;
        uint32_t mask = 0;
        #pragma unroll
        for (int s = 0; s < 4; ++s) if (__ballot(live && item == it0 && state == s) & same) mask |= 1u << s;
        const uint32_t known = (uint32_t)__popcll(__ballot(live && item == it0 && state < 2) & same);
        if ((int)lane == leader) { atomicOr(&A.stateMask[it0], mask); if (P.wantKnownCount && known) atomicAdd(&A.knownCount[it0], known); }
    }
    return committed;
}

// ---- the walks ----
// A wave that runs 64 walks in lockstep has two thirds of its lanes finished after two visits, waiting for its longest walk.  Here a lane that finishes
// gets the next queued micro-triangle: a wave pulls chunks of OMMX_GENERIC_CHUNK entries from a cursor next to the queue's count word, and whenever
// OMMX_GENERIC_REFILL lanes are idle they commit their states together (generic_commit folds the item masks per wave), take the next entries and set them
// up (micro-triangle, centre vote, raster box) in one round.  Same visits in the same order per walk as the serial loop, with its early exit: same state.
// Rows are left at the first texel that is not under the triangle after one that was: the texels under the conservative triangle form an interval in
// every row (each edge function is monotone in x, in fp32 too).
// KIND 0, the level-line kernel (Linear filter): most cells of a long walk are cheap -- flat (no edge can cross a constant patch: the reference votes by the
// first texel) and with all four texels on a side the walk has already seen; when only (any vote above, any vote below) counts, such a cell changes nothing.
// Every lane advances to its next cell that needs work (corner votes: its texels are not all on a seen side; edge tests: it is not flat), a few rounds of
// fetch + compare, and then the wave does the corner votes and the three edge tests for those cells together.  KIND 1, Nearest: every covered texel votes
// with its sample.
#ifndef OMMX_GENERIC_REFILL
#define OMMX_GENERIC_REFILL 16   // measured on the cards workload: 8 / 16 / 32 = 30.2 / 29.0 / 31.1 ms
#endif
#ifndef OMMX_GENERIC_CHUNK
#define OMMX_GENERIC_CHUNK 1024u
#endif
#ifndef OMMX_GENERIC_ADVANCE
#define OMMX_GENERIC_ADVANCE 4   // rounds of advancing per round of cell work (1 / 4 / 8: 28.6 / 28.8 / 31.2 ms)
#endif
template <bool FP32, int KIND, class MD>
__device__ __forceinline__ void generic_walks(const ClassifyParams& P, const ItemArrays& A, const GenericQueue& G, uint32_t n)
{
    const DevMip& m = P.mips[0];
    const uint32_t lane = threadIdx.x & 63u;
    const bool countsMatter = P.promotion == 0;
    const float off = KIND == 0 ? -0.5f : 0.f;
    unsigned long long* const cursorWord = G.count + 1;
    uint32_t chunkNext = 0, chunkEnd = 0;   // (wave-uniform) the chunk of the queue this wave works on
    bool drained = false;                   // (wave-uniform) no entries left in the queue
    uint32_t classified = 0;                // (wave-uniform) entries this wave has committed: one atomic per wave at the end (count[2], a statistic)
    bool have = false, result = false;      // this lane: holds an unfinished walk / a finished one that is not committed yet
    bool rowSeen = false;                   // a texel of the cursor's row was under the triangle
    uint32_t item = 0, levelWord = 0, above = 0, below = 0;
    int direct = -1;                        // state of a degenerate item's micro-triangle (serial fine_state), or -1
    MicroTri t; t.p0 = t.p1 = t.p2 = mk2(0.f, 0.f);
    RasterBox B = raster_box(m, t, off);
    TexelCursor c = cursor_at(B, 0u);
    // to the next texel under the triangle; false: the box is exhausted
    auto next_covered = [&]() -> bool {
        while (cursor_live(B, c) && !texel_under(B, c)) { if (rowSeen) { c.x = B.minx; ++c.y; rowSeen = false; } else cursor_step(B, c); }
        return cursor_live(B, c);
    };
    auto step = [&]() { rowSeen = true; cursor_step(B, c); if (c.x == B.minx) rowSeen = false; };   // (c.x == minx: the step wrapped into the next row)
    for (;;) {
        const unsigned long long busy = __ballot(have);
        const uint32_t idle = 64u - (uint32_t)__popcll(busy);
        if ((!drained && idle >= (uint32_t)OMMX_GENERIC_REFILL) || busy == 0ull) {
            // ---- commit what is finished, hand out the next entries ----
            classified += generic_commit(P, A, result, item, levelWord & 0xFFFFFFu, direct >= 0 ? direct : state_from_coverage(P, above, below));
            result = false;
            if (drained) break;   // (busy == 0)
            if (chunkNext == chunkEnd) {
                unsigned long long start = 0;
                if (lane == 0u) start = atomicAdd(cursorWord, (unsigned long long)OMMX_GENERIC_CHUNK);
                start = (unsigned long long)__shfl((long long)start, 0);
                if (start >= (unsigned long long)n) drained = true;
                else { chunkNext = (uint32_t)start; chunkEnd = start + OMMX_GENERIC_CHUNK < (unsigned long long)n ? (uint32_t)start + OMMX_GENERIC_CHUNK : n; }
            }
            if (!drained) {
                const uint32_t avail = chunkEnd - chunkNext, take = idle < avail ? idle : avail;
                const uint32_t rank = (uint32_t)__popcll(~busy & ((1ull << lane) - 1ull));
                bool get = !have && rank < take;
                uint2 ent = get ? G.entries[chunkNext + rank] : make_uint2(0u, 0u);
                chunkNext += take;
                if (ent.x == 0xFFFFFFFFu) get = false;   // (null entry: the inside part of a reservation that did not fit)
                const bool degenerate = get && ((ent.x >> 30) & 1u) != 0u;
                if (get) { item = ent.x & 0x3FFFFFFFu; levelWord = ent.y; above = 0; below = 0; direct = -1; }
                if (__ballot(degenerate) != 0ull) {   // (rare: degenerate items take the serial form)
                    if (degenerate) { direct = fine_state<FP32, MD>(P, micro_triangle(A.uv + 6ull * item, levelWord & 0xFFFFFFu, levelWord >> 24), true, no_window()); result = true; get = false; }
                }
                if (get) {
                    t = micro_triangle(A.uv + 6ull * item, levelWord & 0xFFFFFFu, levelWord >> 24);
                    if (KIND == 0) vote(P.cutoff < bilinear<FP32, MD, true>(P, m, t.p0, no_window()), above, below);
                    B = raster_box(m, t, off);
                    c = cursor_at(B, 0u); rowSeen = false;
                    have = true;
                }
            }
            continue;
        }
        if (KIND == 1) {   // ---- Nearest: one visit ----
            if (have) {
                if (!next_covered()) { have = false; result = true; }
                else {
                    nearest_texel<FP32, MD>(P, m, c.x, c.y, above, below, no_window());
                    step();
                    if (!countsMatter && above != 0 && below != 0) { have = false; result = true; }
                }
            }
            continue;
        }
        // ---- level line: every walking lane advances to its next cell that needs work ----
        bool cell = false, corners = false;
        float ha = 0.f, hb = 0.f, hc = 0.f, hd = 0.f, pfx = 0.f, pfy = 0.f;
        uint32_t obits = 0;
        for (int round = 0; round < OMMX_GENERIC_ADVANCE; ++round) {
            if (have && !cell) {
                if (!next_covered()) { have = false; result = true; }
                else {
                    pfx = (float)c.x + 0.5f; pfy = (float)c.y + 0.5f;
                    float gx, gy, gz, gw;   // 00, 01, 11, 10
                    fetch_cell<FP32, MD, true>(P, m, MD::pow2(P), c.x, c.y, no_window(), gx, gy, gz, gw);
                    const bool o0 = P.cutoff < gx, o1 = P.cutoff < gy, o2 = P.cutoff < gz, o3 = P.cutoff < gw;
                    hb = gw - gx; hc = gy - gx; hd = gx + gz - gy - gw; ha = gx - P.cutoff;
                    const bool flat = near_zero(hb, 1e-6f) & near_zero(hc, 1e-6f) & near_zero(hd, 1e-6f);
                    const bool seen = !countsMatter & (o0 == o1) & (o1 == o2) & (o2 == o3) & (o0 ? above != 0 : below != 0);
                    // a cell that is not flat needs its three edge tests -- unless the level curve provably stays out of it (cell_excluded, classify_device.h: audited
                    // by the oracle next to every edge test): most cells of a long walk are such cells, all four texels on a side the walk has seen, in the smooth
                    // flank of an alpha edge.  They are passed like the flat ones, so the edge tests below run for the cells that need them.
                    bool edgesNeeded = !flat;
#ifndef OMMX_NO_CELL_EXCLUSION
                    if (edgesNeeded) {
                        const V2 q0 = mk2(m.fw * t.p0.x - pfx, m.fh * t.p0.y - pfy), q1 = mk2(m.fw * t.p1.x - pfx, m.fh * t.p1.y - pfy), q2 = mk2(m.fw * t.p2.x - pfx, m.fh * t.p2.y - pfy);
                        edgesNeeded = !cell_excluded(q0, q1, q2, ha, hb, hc, hd);
                    }
#endif
                    corners = !seen;
                    cell = !seen | edgesNeeded;
                    obits = (o0 ? 1u : 0u) | (o1 ? 2u : 0u) | (o2 ? 4u : 0u) | (o3 ? 8u : 0u) | (flat ? 16u : 0u) | (edgesNeeded ? 32u : 0u);
                    step();
                }
            }
            if (__ballot(have && !cell) == 0ull || __popcll(__ballot(cell)) >= 48) break;
        }
        // ---- the cells that need work: bake_kernels_cpu.h:241-399 ----
        if (cell) {
            const bool o0 = (obits & 1u) != 0u, o1 = (obits & 2u) != 0u, o2 = (obits & 4u) != 0u, o3 = (obits & 8u) != 0u, flat = (obits & 16u) != 0u, edgesNeeded = (obits & 32u) != 0u;
            bool both = false;
            if (corners) {
                const float ipx = pfx * m.rw, ipy = pfy * m.rh;
                const bool in0 = point_in_triangle_flat(t, ipx, ipy), in1 = point_in_triangle_flat(t, ipx, ipy + m.rh);
                const bool in2 = point_in_triangle_flat(t, ipx + m.rw, ipy + m.rh), in3 = point_in_triangle_flat(t, ipx + m.rw, ipy);
                const bool isO = (in0 & o0) | (in1 & o1) | (in2 & o2) | (in3 & o3), isT = (in0 & !o0) | (in1 & !o1) | (in2 & !o2) | (in3 & !o3);
                above += isO ? 1u : 0u; below += isT ? 1u : 0u;
                both = isO & isT;
                if (flat & !both) vote(o0, above, below);
            }
            if (edgesNeeded & !both & (countsMatter | !(above != 0 && below != 0))) {
                const V2 q0 = mk2(m.fw * t.p0.x - pfx, m.fh * t.p0.y - pfy), q1 = mk2(m.fw * t.p1.x - pfx, m.fh * t.p1.y - pfy), q2 = mk2(m.fw * t.p2.x - pfx, m.fh * t.p2.y - pfy);
                const bool x0 = edge_crosses_level_curve(q0, q1, ha, hb, hc, hd), x1 = edge_crosses_level_curve(q1, q2, ha, hb, hc, hd), x2 = edge_crosses_level_curve(q2, q0, ha, hb, hc, hd);
                if (x0 | x1 | x2) { above += 1; below += 1; }
            }
            if (!countsMatter && above != 0 && below != 0) { have = false; result = true; }
        }
    }
    if (lane == 0u && classified) atomicAdd(G.count + 2, (unsigned long long)classified);
}


// ---- the dense form of the walks (round 5) ----
// The walks above keep one lane per micro-triangle for everything, so a wave's instructions are issued for a third of its lanes: the skip to the next covered
// texel runs for the one lane at the start of a wide row, the corner votes for the dozen lanes whose cell is not on a seen side, the edge tests for the cells
// that need them.  Here a lane OWNS a micro-triangle (its record lies in the wave's LDS: edge functions, box, vertices, cursor, vote counters) and WORKS on
// whatever visit is next: every round the active owners offer OMMX_GENERIC_QUOTA texels of their boxes each, the wave tests them 64 at a time (texel_under's
// three edge functions), the covered ones go into a ring of visits; whenever the ring holds 64, the wave fetches 64 cells and does their votes; cells that need
// their edge tests go into a second ring and are done 64 at a time as well.  The votes of a micro-triangle are sums (bake_kernels_cpu.h:241-399 adds to two
// counters per texel; GetStateFromCoverage reads the sums), so the order in which its texels are visited does not matter, and when only (any above, any below)
// counts -- every promotion but Nearest -- the visits of a micro-triangle that is already mixed are dropped wherever they are: the serial loop's early exit.
// An owner finishes when its box is exhausted (or it is mixed) AND none of its visits is in a ring (a counter per owner), so a record is never reused under a
// visit.  No workgroup barrier: the structures are per wave, a wave's LDS operations execute in order.
#ifndef OMMX_GENERIC_DENSE
#define OMMX_GENERIC_DENSE 1
#endif
#ifndef OMMX_GENERIC_QUOTA_LOG2
#define OMMX_GENERIC_QUOTA_LOG2 2   // texels an owner offers per round: 2 / 4 / 8 = 19.3 / 18.2 / 19.5 ms on the cards workload
#endif
#ifndef OMMX_GENERIC_DENSE_CHUNK
#define OMMX_GENERIC_DENSE_CHUNK 256u   // queue entries a wave takes from the cursor at a time: 1024 / 256 / 128 = 18.2 / 16.9 / 17.0 ms (the last chunk of a wave is the launch's tail)
#endif
#ifndef OMMX_GENERIC_DENSE_REFILL
#define OMMX_GENERIC_DENSE_REFILL 24   // idle owners that trigger a refill: 16 / 24 / 32 = 18.46 / 18.2 / 18.2 ms
#endif
constexpr uint32_t GD_REC = 20u;      // dwords per owner record: 0-8 three edges (nx, ny, c); 9 above, 10 below, 11 visits in flight; 12-17 p0, p1, p2; 18 minx, 19 miny (the rest of the box and the cursor stay in the owner's registers)
constexpr uint32_t GD_RING = 128u;    // entries per ring (a ring holds < 64 before 64 are pushed)
constexpr uint32_t GD_WAVE_DWORDS = 64u * GD_REC + 2u * GD_RING + 16u;   // records, two rings of one word per visit (owner | dx << 6 | dy << 19, relative to the box), the slot map
constexpr int GD_MAX_EXTENT = 8192;   // boxes wider or higher than this (13 bits per offset) are offered a piece of a row at a time, each piece after the visits of the one before
#define OMMX_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
template <bool FP32, int KIND, class MD>
__device__ __forceinline__ void generic_dense(const ClassifyParams& P, const ItemArrays& A, const GenericQueue& G, uint32_t n, uint32_t* __restrict__ L)
{
    constexpr uint32_t Q = 1u << OMMX_GENERIC_QUOTA_LOG2, MASK = GD_RING - 1u;
    const DevMip& m = P.mips[0];
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below_me = (1ull << lane) - 1ull;
    const bool countsMatter = P.promotion == 0;
    const float off = KIND == 0 ? -0.5f : 0.f;
    uint32_t* const rec = L;
    uint32_t* const cRing = L + 64u * GD_REC; uint32_t* const eRing = cRing + GD_RING;
    uint8_t* const slotMap = (uint8_t*)(eRing + GD_RING);
    uint32_t* const mine = rec + lane * GD_REC;
    unsigned long long* const cursorWord = G.count + 1;
    uint32_t chunkNext = 0, chunkEnd = 0;
    bool drained = false;
    uint32_t classified = 0;
    uint32_t cHead = 0, cCount = 0, eHead = 0, eCount = 0;   // (wave-uniform) the two rings
#ifdef OMMX_GD_STATS
    uint32_t st[12] = { 0 };
#define GD_STAT(i, v) st[i] += (v)
#else
#define GD_STAT(i, v) do { } while (0)
#endif
    // the owner's side of this lane
    bool have = false, result = false;
    uint32_t item = 0, levelWord = 0, fAbove = 0, fBelow = 0;
    int direct = -1, cx = 0, cy = 0, minx = 0, xend = 1, yend = 0;
    int vminx = 0, vminy = 0, vxend = 1, vyend = 0;   // the part of the box on offer (see `big`)
    bool big = false;

    // ---- 64 visits (or what is left): the cell of a covered texel and its votes; cells that need their edge tests move on to the second ring ----
    auto cell_stage = [&](uint32_t cnt) {
        const bool on = lane < cnt;
        const uint32_t idx = (cHead + lane) & MASK;
        const uint32_t word = on ? cRing[idx] : 0u, slot = word & 63u;
        cHead = (cHead + cnt) & MASK; cCount -= cnt;
        uint32_t* const r = rec + slot * GD_REC;
        const uint4 v1 = *(const uint4*)(r + 16);   // p2, minx, miny
        const int x = (int)v1.z + (int)((word >> 6) & 8191u), y = (int)v1.w + (int)(word >> 19);
        const uint2 ab = on ? make_uint2(r[9], r[10]) : make_uint2(0u, 0u);
        const uint32_t above = ab.x, below = ab.y;
        const bool live = on && (countsMatter || !(above != 0 && below != 0));
        GD_STAT(4, 1u); GD_STAT(5, (uint32_t)__popcll(__ballot(live)));
        uint32_t da = 0, db = 0;
        bool toEdge = false;
        float ha = 0.f, hb = 0.f, hc = 0.f, hd = 0.f;
        if (live) {
            if (KIND == 1) nearest_texel<FP32, MD>(P, m, x, y, da, db, no_window());
            else {
                const uint4 v0 = *(const uint4*)(r + 12);
                MicroTri t; t.p0 = mk2(__uint_as_float(v0.x), __uint_as_float(v0.y)); t.p1 = mk2(__uint_as_float(v0.z), __uint_as_float(v0.w)); t.p2 = mk2(__uint_as_float(v1.x), __uint_as_float(v1.y));
                finish_tri(t);
                const float pfx = (float)x + 0.5f, pfy = (float)y + 0.5f;
                float gx, gy, gz, gw;   // 00, 01, 11, 10
                fetch_cell<FP32, MD, true>(P, m, MD::pow2(P), x, y, no_window(), gx, gy, gz, gw);
                const bool o0 = P.cutoff < gx, o1 = P.cutoff < gy, o2 = P.cutoff < gz, o3 = P.cutoff < gw;
                hb = gw - gx; hc = gy - gx; hd = gx + gz - gy - gw; ha = gx - P.cutoff;
                const bool flat = near_zero(hb, 1e-6f) & near_zero(hc, 1e-6f) & near_zero(hd, 1e-6f);
                const bool seen = !countsMatter & (o0 == o1) & (o1 == o2) & (o2 == o3) & (o0 ? above != 0 : below != 0);
                bool edgesNeeded = !flat;
#ifndef OMMX_NO_CELL_EXCLUSION
                if (edgesNeeded) {
                    const V2 q0 = mk2(m.fw * t.p0.x - pfx, m.fh * t.p0.y - pfy), q1 = mk2(m.fw * t.p1.x - pfx, m.fh * t.p1.y - pfy), q2 = mk2(m.fw * t.p2.x - pfx, m.fh * t.p2.y - pfy);
                    edgesNeeded = !cell_excluded(q0, q1, q2, ha, hb, hc, hd);
                }
#endif
                bool both = false;
                if (!seen) {
                    const float ipx = pfx * m.rw, ipy = pfy * m.rh;
                    const bool in0 = point_in_triangle_flat(t, ipx, ipy), in1 = point_in_triangle_flat(t, ipx, ipy + m.rh);
                    const bool in2 = point_in_triangle_flat(t, ipx + m.rw, ipy + m.rh), in3 = point_in_triangle_flat(t, ipx + m.rw, ipy);
                    const bool isO = (in0 & o0) | (in1 & o1) | (in2 & o2) | (in3 & o3), isT = (in0 & !o0) | (in1 & !o1) | (in2 & !o2) | (in3 & !o3);
                    da += isO ? 1u : 0u; db += isT ? 1u : 0u;
                    both = isO & isT;
                    if (flat & !both) vote(o0, da, db);
                }
                toEdge = edgesNeeded & !both & (countsMatter | !((above + da) != 0 && (below + db) != 0));
            }
        }
        if (da) atomicAdd(r + 9, da);
        if (db) atomicAdd(r + 10, db);
        const unsigned long long em = __ballot(toEdge);
        if (toEdge) {
            const uint32_t e = (eHead + eCount + (uint32_t)__popcll(em & below_me)) & MASK;
            eRing[e] = word;   // (the cell is fetched again there: 16 bytes of ring per entry cost a wave per SIMD)
        }
        eCount += (uint32_t)__popcll(em); GD_STAT(6, (uint32_t)__popcll(em));
        if (on && !toEdge) atomicSub(r + 11, 1u);
        OMMX_WAVE_SYNC();
    };
    // ---- 64 cells' edge tests (bake_kernels_cpu.h:241-399: an edge of the micro-triangle that crosses the level curve inside the cell votes for both sides) ----
    auto edge_stage = [&](uint32_t cnt) {
        const bool on = lane < cnt;
        const uint32_t idx = (eHead + lane) & MASK;
        const uint32_t word = on ? eRing[idx] : 0u, slot = word & 63u;
        eHead = (eHead + cnt) & MASK; eCount -= cnt;
        uint32_t* const r = rec + slot * GD_REC;
        const uint2 ab = on ? make_uint2(r[9], r[10]) : make_uint2(0u, 0u);
        GD_STAT(7, 1u); GD_STAT(8, (uint32_t)__popcll(__ballot(on && (countsMatter || !(ab.x != 0 && ab.y != 0)))));
        if (on && (countsMatter || !(ab.x != 0 && ab.y != 0))) {
            const uint4 v0 = *(const uint4*)(r + 12), v1 = *(const uint4*)(r + 16);
            const int x = (int)v1.z + (int)((word >> 6) & 8191u), y = (int)v1.w + (int)(word >> 19);
            const float pfx = (float)x + 0.5f, pfy = (float)y + 0.5f;
            float gx, gy, gz, gw;   // 00, 01, 11, 10
            fetch_cell<FP32, MD, true>(P, m, MD::pow2(P), x, y, no_window(), gx, gy, gz, gw);
            const float hb = gw - gx, hc = gy - gx, hd = gx + gz - gy - gw, ha = gx - P.cutoff;
            const V2 q0 = mk2(m.fw * __uint_as_float(v0.x) - pfx, m.fh * __uint_as_float(v0.y) - pfy), q1 = mk2(m.fw * __uint_as_float(v0.z) - pfx, m.fh * __uint_as_float(v0.w) - pfy),
                     q2 = mk2(m.fw * __uint_as_float(v1.x) - pfx, m.fh * __uint_as_float(v1.y) - pfy);
            const bool x0 = edge_crosses_level_curve(q0, q1, ha, hb, hc, hd), x1 = edge_crosses_level_curve(q1, q2, ha, hb, hc, hd), x2 = edge_crosses_level_curve(q2, q0, ha, hb, hc, hd);
            if (x0 | x1 | x2) { atomicAdd(r + 9, 1u); atomicAdd(r + 10, 1u); }
        }
        if (on) atomicSub(r + 11, 1u);
        OMMX_WAVE_SYNC();
    };

    for (;;) {
        // ---- owners: finished? ----
        bool exhausted = false, mixed = false, waiting = false;
        if (have) {
            const uint4 w = *(const uint4*)(mine + 8);
            const uint2 ab = make_uint2(w.y, w.z);
            const uint32_t pending = w.w;
            mixed = !countsMatter && ab.x != 0 && ab.y != 0;
            exhausted = cy >= yend;
            if (pending == 0u && (exhausted | mixed)) { have = false; result = true; fAbove = ab.x; fBelow = ab.y; }
            waiting = big && pending != 0u;
        }
        const unsigned long long busy = __ballot(have);
        const uint32_t idle = 64u - (uint32_t)__popcll(busy);
        if ((!drained && idle >= (uint32_t)OMMX_GENERIC_DENSE_REFILL) || busy == 0ull) {
            // ---- commit what is finished, hand out the next entries ----
            classified += generic_commit(P, A, result, item, levelWord & 0xFFFFFFu, direct >= 0 ? direct : state_from_coverage(P, fAbove, fBelow));
            result = false;
            if (drained) break;   // (busy == 0)
            if (chunkNext == chunkEnd) {
                unsigned long long start = 0;
                if (lane == 0u) start = atomicAdd(cursorWord, (unsigned long long)OMMX_GENERIC_DENSE_CHUNK);
                start = (unsigned long long)__shfl((long long)start, 0);
                if (start >= (unsigned long long)n) drained = true;
                else { chunkNext = (uint32_t)start; chunkEnd = start + OMMX_GENERIC_DENSE_CHUNK < (unsigned long long)n ? (uint32_t)start + OMMX_GENERIC_DENSE_CHUNK : n; }
            }
            if (!drained) {
                const uint32_t avail = chunkEnd - chunkNext, take = idle < avail ? idle : avail;
                const uint32_t rank = (uint32_t)__popcll(~busy & below_me);
                bool get = !have && rank < take;
                uint2 ent = get ? G.entries[chunkNext + rank] : make_uint2(0u, 0u);
                chunkNext += take;
                if (ent.x == 0xFFFFFFFFu) get = false;   // (null entry: the inside part of a reservation that did not fit)
                const bool degenerate = get && ((ent.x >> 30) & 1u) != 0u;
                if (get) { item = ent.x & 0x3FFFFFFFu; levelWord = ent.y; direct = -1; }
                if (__ballot(degenerate) != 0ull) {   // (rare: degenerate items take the serial form)
                    if (degenerate) { direct = fine_state<FP32, MD>(P, micro_triangle(A.uv + 6ull * item, levelWord & 0xFFFFFFu, levelWord >> 24), true, no_window()); result = true; get = false; }
                }
                GD_STAT(9, 1u); GD_STAT(10, (uint32_t)__popcll(__ballot(get)));
                if (get) {
                    const MicroTri t = micro_triangle(A.uv + 6ull * item, levelWord & 0xFFFFFFu, levelWord >> 24);
                    uint32_t a0 = 0, b0 = 0;
                    if (KIND == 0) vote(P.cutoff < bilinear<FP32, MD, true>(P, m, t.p0, no_window()), a0, b0);
                    const RasterBox B = raster_box(m, t, off);
                    minx = B.minx; xend = B.xend; yend = B.yend; cx = B.minx; cy = B.miny;
                    // (the part of the box the workers see: all of it, unless its offsets do not fit the rings' 13 bits -- a micro-triangle more than 8192 texels across)
                    big = (long long)B.xend - (long long)B.minx > (long long)GD_MAX_EXTENT || (long long)B.yend - (long long)B.miny > (long long)GD_MAX_EXTENT;
                    vminx = B.minx; vminy = B.miny; vxend = B.xend; vyend = B.yend;
                    *(uint4*)(mine + 0) = make_uint4(__float_as_uint(B.e0.nx), __float_as_uint(B.e0.ny), __float_as_uint(B.e0.c), __float_as_uint(B.e1.nx));
                    *(uint4*)(mine + 4) = make_uint4(__float_as_uint(B.e1.ny), __float_as_uint(B.e1.c), __float_as_uint(B.e2.nx), __float_as_uint(B.e2.ny));
                    *(uint4*)(mine + 8) = make_uint4(__float_as_uint(B.e2.c), a0, b0, 0u);
                    *(uint4*)(mine + 12) = make_uint4(__float_as_uint(t.p0.x), __float_as_uint(t.p0.y), __float_as_uint(t.p1.x), __float_as_uint(t.p1.y));
                    *(uint4*)(mine + 16) = make_uint4(__float_as_uint(t.p2.x), __float_as_uint(t.p2.y), (uint32_t)vminx, (uint32_t)vminy);
                    have = true;
                }
            }
            OMMX_WAVE_SYNC();
            continue;
        }
        // ---- every active owner offers the next Q texels of its box; 64 of them are tested per pass, the covered ones become visits ----
        const bool active = have && !exhausted && !mixed && !waiting;
        const unsigned long long emask = __ballot(active);
        const uint32_t nAct = (uint32_t)__popcll(emask);
        if (nAct == 0u) {   // nothing to enumerate: what the waiting owners wait for is in the rings
            GD_STAT(11, 1u);
            while (cCount) { cell_stage(cCount < 64u ? cCount : 64u); if (eCount >= 64u) edge_stage(64u); }
            while (eCount) edge_stage(eCount < 64u ? eCount : 64u);
            continue;
        }
        if (active) slotMap[(uint32_t)__popcll(emask & below_me)] = (uint8_t)lane;
        if (active && big) {   // (rare) the next piece: the rest of the cursor's row, at most Q texels; the ring words of its visits count from the cursor
            vminx = cx; vminy = cy; vxend = (long long)xend - (long long)cx > (long long)Q ? cx + (int)Q : xend; vyend = cy + 1;
            *(uint2*)(mine + 18) = make_uint2((uint32_t)vminx, (uint32_t)vminy);
        }
        OMMX_WAVE_SYNC();
        const uint32_t passes = (nAct * Q + 63u) >> 6;
        GD_STAT(0, 1u); GD_STAT(1, passes);
        for (uint32_t p = 0; p < passes; ++p) {
            const uint32_t rk = p * (64u >> OMMX_GENERIC_QUOTA_LOG2) + (lane >> OMMX_GENERIC_QUOTA_LOG2);
            const bool valid = rk < nAct;
            const uint32_t slot = valid ? (uint32_t)slotMap[rk] : 0u;
            uint32_t* const r = rec + slot * GD_REC;
            bool covered = false;
            int x = 0, y = 0;
            // (box and cursor come from the owner's registers)
            const int bminx = __shfl(vminx, (int)slot), bminy = __shfl(vminy, (int)slot), bxend = __shfl(vxend, (int)slot), byend = __shfl(vyend, (int)slot);
            x = __shfl(cx, (int)slot); y = __shfl(cy, (int)slot);
            if (valid) {
                const uint4 w0 = *(const uint4*)(r + 0), w1 = *(const uint4*)(r + 4); const uint32_t w2x = r[8];
                const uint32_t d = lane & (Q - 1u);
                #pragma unroll
                for (uint32_t s = 0; s + 1u < Q; ++s) if (s < d) { if (++x == bxend) { x = bminx; ++y; } }
                if (y < byend) {
                    EdgeEq e0, e1, e2;
                    e0.nx = __uint_as_float(w0.x); e0.ny = __uint_as_float(w0.y); e0.c = __uint_as_float(w0.z); e0.bias = 0.f;
                    e1.nx = __uint_as_float(w0.w); e1.ny = __uint_as_float(w1.x); e1.c = __uint_as_float(w1.y); e1.bias = 0.f;
                    e2.nx = __uint_as_float(w1.z); e2.ny = __uint_as_float(w1.w); e2.c = __uint_as_float(w2x); e2.bias = 0.f;
                    const float sx = (float)x, sy = (float)y;
                    covered = (eval_cons(e0, sx, sy) < 0.f) & (eval_cons(e1, sx, sy) < 0.f) & (eval_cons(e2, sx, sy) < 0.f);
                }
            }
            const unsigned long long cm = __ballot(covered);
            GD_STAT(2, (uint32_t)__popcll(__ballot(valid && y < byend))); GD_STAT(3, (uint32_t)__popcll(cm));
            if (covered) {
                const uint32_t c = (cHead + cCount + (uint32_t)__popcll(cm & below_me)) & MASK;
                cRing[c] = slot | ((uint32_t)(x - bminx) << 6) | ((uint32_t)(y - bminy) << 19);
                atomicAdd(r + 11, 1u);
            }
            cCount += (uint32_t)__popcll(cm);
            OMMX_WAVE_SYNC();
            if (cCount >= 64u) { cell_stage(64u); if (eCount >= 64u) edge_stage(64u); }
        }
        if (active) {
            const uint32_t steps = big ? (uint32_t)(vxend - vminx) : Q;
            #pragma unroll
            for (uint32_t s = 0; s < Q; ++s) if (s < steps && cy < yend) { if (++cx == xend) { cx = minx; ++cy; } }
        }
        OMMX_WAVE_SYNC();
    }
    if (lane == 0u && classified) atomicAdd(G.count + 2, (unsigned long long)classified);
#ifdef OMMX_GD_STATS
    if (lane == 0u) for (int i = 0; i < 12; ++i) atomicAdd(G.count + 8 + i, (unsigned long long)st[i]);
#endif
}

#ifndef OMMX_GENERIC_WAVES
#define OMMX_GENERIC_WAVES 6
#endif
template <bool FP32, class MD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OMMX_GENERIC_WAVES, OMMX_GENERIC_WAVES))) void classify_generic(ClassifyParams P, ItemArrays A, GenericQueue G)
{
    const uint32_t n = *G.count < (unsigned long long)G.capacity ? (uint32_t)*G.count : G.capacity;
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, waves = (gridDim.x * 256u) >> 6;
    if (P.mipCount == 1 && !(P.filterLinear && P.altKernel)) {
#if OMMX_GENERIC_DENSE
        __shared__ __attribute__((aligned(16))) uint32_t s_dense[4][GD_WAVE_DWORDS];
        if (P.filterLinear) generic_dense<FP32, 0, MD>(P, A, G, n, s_dense[threadIdx.x >> 6]); else generic_dense<FP32, 1, MD>(P, A, G, n, s_dense[threadIdx.x >> 6]);
#else
        if (P.filterLinear) generic_walks<FP32, 0, MD>(P, A, G, n); else generic_walks<FP32, 1, MD>(P, A, G, n);
#endif
        return;
    }
    uint32_t classified = 0;
    for (uint32_t e0 = wave * 64u; e0 < n; e0 += waves * 64u) {   // (wave-uniform) mip chains / the alternative kernel: the serial form per entry
        const uint32_t e = e0 + lane;
        bool live = e < n;
        uint2 ent = live ? G.entries[e] : make_uint2(0u, 0u);
        if (ent.x == 0xFFFFFFFFu) { live = false; ent = make_uint2(0u, 0u); }   // (null entry: the inside part of a reservation that did not fit)
        const uint32_t item = ent.x & 0x3FFFFFFFu;
        int state = 0;
        if (live) state = fine_state<FP32, MD>(P, micro_triangle(A.uv + 6ull * item, ent.y & 0xFFFFFFu, ent.y >> 24), ((ent.x >> 30) & 1u) != 0u, no_window());
        classified += generic_commit(P, A, live, item, ent.y & 0xFFFFFFu, state);
    }
    if (lane == 0u && classified) atomicAdd(G.count + 2, (unsigned long long)classified);
}

// ---- launches ----
// Items of level >= 6 are cut into 4096-tiles, level-5 items are one 1024-tile; both go through triage_tiles + ONE persistent
// classify_tiles launch per tile size, whatever the mix of levels (the level travels in the tile record).  Items below level 5 are
// packed several to a 1024-tile and keep a plain grid launch per level (their total cost is negligible: <= 256 micro-triangles each).
static uint32_t tiles_per_item(uint32_t level, uint32_t tileLog4) { return 1u << (2u * (level - tileLog4)); }

uint64_t classify_queue_records(const uint32_t count[kNumLevels], bool sections)
{
    uint64_t n = 0;
    for (uint32_t l = 5; l < (uint32_t)kNumLevels; ++l) n += (uint64_t)count[l] * tiles_per_item(l, l >= 6 ? 6u : 5u) * (sections && l >= 6 ? 2u : 1u);
    return n;
}

void classify_plan(const uint32_t first[kNumLevels], const uint32_t count[kNumLevels], uint32_t ranges, ClassifyPlan* plan)
{
    memset(plan, 0, sizeof *plan);
    for (int c = 0; c < 2; ++c) {   // 0: 4096-tiles (levels 6..12, highest first), 1: 1024-tiles (level 5)
        TileLevels& L = c == 0 ? plan->big : plan->small;
        uint64_t total = 0;
        for (int level = c == 0 ? kMaxLevel : 5; level >= (c == 0 ? 6 : 5); --level) {
            if (!count[level]) continue;
            L.level[L.n] = (uint32_t)level; L.first[L.n] = first[level]; L.tileStart[L.n] = (uint32_t)total;
            total += (uint64_t)count[level] * tiles_per_item((uint32_t)level, c == 0 ? 6u : 5u);
            L.n++;
        }
        if (total > 0x7FFFFFFFull) total = 0;   // (that many tiles cannot happen: their packed states would not fit in HBM)
        L.tileStart[L.n] = (uint32_t)total;
        (c == 0 ? plan->totalBig : plan->totalSmall) = total;
    }
    // ranges of the 4096-tile enumeration: K pieces of (about) equal tile counts, cut at work-item boundaries, in the order of the tile enumeration = the order
    // of the final result (highest level first, then the position in that level's active list)
    // (equal ranges: measured against growing and bell-shaped splits, which leave the copy engine idle early or a large last range exposed -- the PCIe copy
    //  is the slower pipe from the first range on, so it wants a steady supply of small pieces)
    if (ranges > kMaxStreamRanges) ranges = kMaxStreamRanges;
    const uint32_t K = plan->totalBig ? (ranges ? ranges : 1u) : 0u;
    TileSections& S = plan->ranges; S.n = K;
    // (round 4: the classification of the metric workload is now faster than the PCIe copy of its result, so the copy engine is the critical pipe and what
    //  counts is how early its FIRST piece is ready: the first ranges are small -- a quarter, a half, three quarters of the others --, the rest equal)
#ifndef OMMX_STREAM_RAMP
#define OMMX_STREAM_RAMP 1
#endif
    double wsum = 0.0, wacc = 0.0;
    auto weight = [&](uint32_t k) { return (OMMX_STREAM_RAMP && K >= 8u && k < 3u) ? 0.25 * (double)(k + 1u) : 1.0; };
    for (uint32_t k = 0; k < K; ++k) wsum += weight(k);
    for (uint32_t k = 1; k <= K; ++k) {
        wacc += weight(k - 1u);
        uint64_t t = k == K ? plan->totalBig : (uint64_t)((double)plan->totalBig * (wacc / wsum));
        if (k < K) {
            const TileLevels& L = plan->big; uint32_t g = 0;
            while (g + 1 < L.n && t >= L.tileStart[g + 1]) ++g;
            const uint32_t per = tiles_per_item(L.level[g], 6u);
            t = L.tileStart[g] + (t - L.tileStart[g]) / per * per;
        }
        S.cut[k] = (uint32_t)t;
    }
}

// the staged tiles of the early items, ordered by range into the queue's second copy: bases of the odd sections (their tails were counted by triage_tiles) ...
__global__ void early_tiles_bases(uint32_t* __restrict__ queueCtl, uint32_t ranges, uint32_t secondCopy)
{
    uint32_t run = secondCopy;
    for (uint32_t k = 0; k < ranges; ++k) { queueCtl[kSecBases + 2u * k + 1u] = run; run += queueCtl[kSecTails + 2u * k + 1u]; }
}
// ... and the records (bits 16..21 of word 1 = range, removed here)
__global__ __launch_bounds__(256) void early_tiles_scatter(const uint4* __restrict__ staged, uint4* __restrict__ queue, uint32_t* __restrict__ queueCtl)
{
    __shared__ uint32_t s_cnt[kMaxStreamRanges], s_base[kMaxStreamRanges];   // (one global atomic per range and 256 records, not per record)
    const uint32_t n = queueCtl[kCtlEarlyStaged];
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        if (threadIdx.x < kMaxStreamRanges) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        uint4 r0 = make_uint4(0u, 0u, 0u, 0u), r1 = r0, r2 = r0; uint32_t k = 0, local = 0;
        if (i < n) {
            r0 = staged[(size_t)kTileRecordWords * i]; r1 = staged[(size_t)kTileRecordWords * i + 1]; r2 = staged[(size_t)kTileRecordWords * i + 2];
            k = (r0.y >> 16) & (kMaxStreamRanges - 1u);
            local = atomicAdd(&s_cnt[k], 1u);
        }
        __syncthreads();
        if (threadIdx.x < kMaxStreamRanges && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(queueCtl + kCtlEarlyFill + threadIdx.x, s_cnt[threadIdx.x]);
        __syncthreads();
        if (i < n) {
            uint4* dst = queue + (size_t)kTileRecordWords * (queueCtl[kSecBases + 2u * k + 1u] + s_base[k] + local);
            dst[0] = make_uint4(r0.x, r0.y & 0xFF00FFFFu, r0.z, r0.w); dst[1] = r1; dst[2] = r2;
        }
        __syncthreads();
    }
}

template <bool FP32, class MD>
static void launch_classify_md(const ClassifyParams& P, const ItemArrays& A, const uint32_t* activeIds, const uint32_t first[kNumLevels],
                               const uint32_t count[kNumLevels], uint4* queue, uint32_t* queueCtl, uint32_t numCUs, const ClassifyChunks& chunks, hipStream_t stream)
{
    // queueCtl: the 4096-tile queue's section words (kSecTails / kSecHeads / kSecBases / kSecDone), then the same four words of the 1024-tile queue (one section)
    uint32_t* ctl1024 = queueCtl + kCtl1024;
    ClassifyPlan plan; classify_plan(first, count, chunks.count, &plan);
    const TileSections& S = plan.ranges; const uint32_t K = S.n;
    const bool paired = K > 1u;   // (two sections per range; the queue holds a second copy of the levels >= 6 for the odd ones: classify_queue_records)
    const bool deferred = chunks.generic.entries != nullptr && chunks.after == nullptr;
    GenericQueue noGeneric; memset(&noGeneric, 0, sizeof noGeneric);
    uint4* q1024 = queue + (plan.totalBig * (paired ? 2u : 1u)) * kTileRecordWords;
    TileSections one; memset(&one, 0, sizeof one); one.n = 1;
    // ---- sliced items, step 1: tile triage of both tile sizes (settled tiles are final after it, open ones are queued) ----
    if (plan.totalBig) {
        hipLaunchKernelGGL((triage_tiles<4096>), dim3((uint32_t)((plan.totalBig + 255u) / 256u)), dim3(256), 0, stream, P, A, activeIds, plan.big, queue, queueCtl, S,
                           paired ? chunks.early : (const uint8_t*)nullptr, chunks.earlyLead, (uint4*)chunks.earlyStage);
        if (paired && chunks.early) {
            hipLaunchKernelGGL(early_tiles_bases, dim3(1), dim3(1), 0, stream, queueCtl, K, (uint32_t)plan.totalBig);
            hipLaunchKernelGGL(early_tiles_scatter, dim3(1024), dim3(256), 0, stream, (const uint4*)chunks.earlyStage, queue, queueCtl);
        }
    }
    if (plan.totalSmall)
        hipLaunchKernelGGL((triage_tiles<1024>), dim3((uint32_t)((plan.totalSmall + 255u) / 256u)), dim3(256), 0, stream, P, A, activeIds, plan.small, q1024, ctl1024, one,
                           (const uint8_t*)nullptr, (const uint32_t*)nullptr, (uint4*)nullptr);
    // ---- sliced items, step 1b: group triage of the open tiles (verdict bytes into the records; grid-stride over the sections' device-side counts) ----
    {
        const uint32_t cap = numCUs * 16u;   // workgroups (a wave per 4096-tile / four 1024-tiles per wave); the queue's fill is only known on the device
        if (plan.totalSmall) {
            const uint64_t need = (plan.totalSmall + 15u) / 16u;
            hipLaunchKernelGGL((triage_groups<FP32, MD, 1024>), dim3((uint32_t)(need < cap ? need : cap)), dim3(256), 0, stream, P, A, q1024, (const uint32_t*)ctl1024, 1u, 1u);
        }
        if (plan.totalBig) {
            const uint32_t window = plan.big.level[0] == 6u ? 1u : kChunkWindow;   // (levels highest first: 6 on top = one tile per item everywhere)
            const uint64_t need = (plan.totalBig + 4u * window - 1u) / (4u * window);
            hipLaunchKernelGGL((triage_groups<FP32, MD, 4096>), dim3((uint32_t)(need < cap ? need : cap)), dim3(256), 0, stream, P, A, queue, (const uint32_t*)queueCtl, paired ? 2u * K : K, window);
        }
    }
    // ---- small items: one launch per level ----
    for (uint32_t level = 0; level < 5u; ++level) {
        if (!count[level]) continue;
        const uint64_t M = 1ull << (2 * level);
        const uint64_t tiles = ((uint64_t)count[level] * M + 1023u) / 1024u;
        const uint32_t gx = tiles < (1u << 20) ? (uint32_t)tiles : (1u << 20), gy = (uint32_t)((tiles + gx - 1) / gx);
        hipLaunchKernelGGL((classify_tiles<FP32, false, 1024, MD>), dim3(gx, gy), dim3(BLOCK), 0, stream, P, A, activeIds + first[level], count[level], level, tiles,
                           (const uint4*)nullptr, (uint32_t*)nullptr, 0u, noGeneric);
    }
    // ---- sliced items, step 2: one persistent launch per queue; every CU holds OMMX_CLASSIFY_WAVES workgroups of 4 waves (one per SIMD) ----
    // A streamed bake runs its placement kernels NEXT TO the persistent launch.  They are small (<= 25 VGPRs, <= 17 KB of LDS: they fit beside six of these
    // workgroups on a CU), but the grid must stay below what the chip can hold: with every slot requested (or 16 fewer) the dispatcher keeps workgroups of
    // this launch pending and lets nothing else in -- measured with device time stamps: the one-lane wait kernel of the first range then starts 24 ms late,
    // when the first of these workgroups exits.  Half a workgroup per CU less than full is the measured optimum (metric configuration, ms per ommCpuBake, by
    // workgroups held back: 0 / 16: 52.7, 64: 34.6 .. 37.7 (unstable), 96: 34.9, 128: 34.8, 192: 35.4, 256: 35.4; configs[4]: 64: 145, 128: 148, 256: 155).
#ifndef OMMX_STREAM_HOLDBACK   // workgroups a streamed bake's persistent launch leaves free, in eighths of the CU count (4 = half a workgroup per CU)
#define OMMX_STREAM_HOLDBACK 4
#endif
    const uint64_t want = (uint64_t)numCUs * OMMX_CLASSIFY_WAVES - (chunks.after ? numCUs * OMMX_STREAM_HOLDBACK / 8u : 0u);
    if (plan.totalSmall) {
        const dim3 cg((uint32_t)(plan.totalSmall < want ? plan.totalSmall : want)), cb(BLOCK);
        if (deferred) hipLaunchKernelGGL((classify_tiles<FP32, true, 1024, MD, true>), cg, cb, 0, stream, P, A, (const uint32_t*)nullptr, 0u, 0u, (uint64_t)0, (const uint4*)q1024, ctl1024, 1u, chunks.generic);
        else hipLaunchKernelGGL((classify_tiles<FP32, true, 1024, MD>), cg, cb, 0, stream, P, A, (const uint32_t*)nullptr, 0u, 0u, (uint64_t)0, (const uint4*)q1024, ctl1024, 1u, noGeneric);
    }
    if (chunks.mark) chunks.mark(chunks.user);   // (everything but the persistent launch of the levels >= 6 is enqueued)
    if (plan.totalBig) {
        const dim3 cg((uint32_t)(plan.totalBig < want ? plan.totalBig : want)), cb(BLOCK);
        if (deferred) hipLaunchKernelGGL((classify_tiles<FP32, true, 4096, MD, true>), cg, cb, 0, stream, P, A, (const uint32_t*)nullptr, 0u, 0u, (uint64_t)0, (const uint4*)queue, queueCtl,
                                         paired ? 2u * K : K, chunks.generic);
        else hipLaunchKernelGGL((classify_tiles<FP32, true, 4096, MD>), cg, cb, 0, stream, P, A, (const uint32_t*)nullptr, 0u, 0u, (uint64_t)0, (const uint4*)queue, queueCtl,
                                paired ? 2u * K : K, noGeneric);
    }
    // ---- deferred generic pass: the micro-triangles of several texels that the persistent launches queued instead of walking ----
    if (deferred) {
        if (chunks.markGeneric) chunks.markGeneric(chunks.user);
        hipLaunchKernelGGL((classify_generic<FP32, MD>), dim3(numCUs * (OMMX_GENERIC_DENSE ? (uint32_t)OMMX_GENERIC_WAVES : 8u)), dim3(256), 0, stream, P, A, chunks.generic);
    }
    for (uint32_t k = 0; k < K; ++k) {
        if (chunks.after) {   // the work items of this range, as segments of the per-level active lists, in the order of the final result
            ClassifySegment segs[kNumLevels]; uint32_t ns = 0;
            const TileLevels& L = plan.big;
            for (uint32_t g = 0; g < L.n; ++g) {
                const uint32_t lo = S.cut[k] > L.tileStart[g] ? S.cut[k] : L.tileStart[g], hi = S.cut[k + 1] < L.tileStart[g + 1] ? S.cut[k + 1] : L.tileStart[g + 1];
                if (lo >= hi) continue;
                const uint32_t per = tiles_per_item(L.level[g], 6u);
                segs[ns].level = L.level[g]; segs[ns].first = L.first[g] + (lo - L.tileStart[g]) / per; segs[ns].count = (hi - lo) / per; ns++;
            }
            chunks.after(chunks.user, k, segs, ns, false);
        }
    }
    if (chunks.after) {   // the lower levels come last in the result (descending level): one more call, after the last big range
        ClassifySegment segs[kNumLevels]; uint32_t ns = 0;
        for (int level = 5; level >= 0; --level) if (count[level]) { segs[ns].level = (uint32_t)level; segs[ns].first = first[level]; segs[ns].count = count[level]; ns++; }
        chunks.after(chunks.user, K, segs, ns, true);
    }
}

hipError_t launch_classify(const ClassifyParams& P, const ItemArrays& A, const uint32_t* activeIds, const uint32_t first[kNumLevels], const uint32_t count[kNumLevels],
                           void* queue, uint32_t* queueCtl, uint32_t numCUs, hipStream_t stream, const ClassifyChunks* chunksIn)
{
    ClassifyChunks chunks; memset(&chunks, 0, sizeof chunks); chunks.count = 1;
    if (chunksIn) chunks = *chunksIn;
    if (chunks.count > kMaxStreamRanges) chunks.count = kMaxStreamRanges;
    uint64_t any = 0; for (int l = 0; l < kNumLevels; ++l) any += count[l];
    if (!any) return hipSuccess;
    hipError_t e = hipMemsetAsync(queueCtl, 0, kClassifyCtlWords * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    // the two address-mode/pow2 pairs that real assets use get their own instantiation (the reference has one per pair); the rest is dynamic
    const bool wrapP2 = P.addrMode == 0 && P.pow2Dispatch, clampP2 = P.addrMode == 2 && P.pow2Dispatch;
    uint4* q = (uint4*)queue;
#ifdef OMMX_ONLY_HOT   // (A/B and resource-usage builds: one instantiation set -- UNORM8 texels, Wrap addressing, power-of-two size -- compiles in a sixth of the time)
    if (P.texIsFp32 || !wrapP2) return hipErrorNotSupported;
    launch_classify_md<false, ModeStatic<0, 1>>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
#elif defined(OMMX_ONLY_U8)   // (A/B builds for the bench configurations: UNORM8 texels, power-of-two size, Wrap or Clamp)
    if (P.texIsFp32 || !(wrapP2 || clampP2)) return hipErrorNotSupported;
    if (wrapP2) launch_classify_md<false, ModeStatic<0, 1>>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
    else launch_classify_md<false, ModeStatic<2, 1>>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
#else
    if (P.texIsFp32) {
        if (wrapP2) launch_classify_md<true, ModeStatic<0, 1>>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
        else if (clampP2) launch_classify_md<true, ModeStatic<2, 1>>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
        else launch_classify_md<true, ModeDynamic>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
    } else {
        if (wrapP2) launch_classify_md<false, ModeStatic<0, 1>>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
        else if (clampP2) launch_classify_md<false, ModeStatic<2, 1>>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
        else launch_classify_md<false, ModeDynamic>(P, A, activeIds, first, count, q, queueCtl, numCUs, chunks, stream);
    }
#endif
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// XXH64 (seed 42) over the item's 3-state byte stream: one byte per micro-triangle, UT folded into
// UO (bake_cpu_impl.cpp:374-377,1038-1040).  The bytes are never materialised: each lane expands
// its 8-byte lane of every 32-byte stripe from the packed states on the fly.
//
// XXH64 keeps four independent accumulators, each a strictly sequential chain over the stripes, so
// the parallelism is (items) x (4 accumulators): a wave carries 16 items, lane = (item, accumulator).
// ------------------------------------------------------------------------------------------------
constexpr uint64_t XP1 = 0x9E3779B185EBCA87ULL, XP2 = 0xC2B2AE3D27D4EB4FULL, XP3 = 0x165667B19E3779F9ULL,
                   XP4 = 0x85EBCA77C2B2AE63ULL, XP5 = 0x27D4EB2F165667C5ULL;
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xxh_round(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
__device__ __forceinline__ uint64_t xxh_merge(uint64_t h, uint64_t v) { h ^= xxh_round(0, v); return h * XP1 + XP4; }
__device__ __forceinline__ uint64_t xxh_avalanche(uint64_t h) { h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32; return h; }

// 8 consecutive micro-triangle states -> 8 bytes (little endian), UT(2) -> UO(3).  Bit-spreading instead of a per-state loop:
// four 2-bit (or 1-bit) fields of a byte are moved to the low bits of four bytes with two shift/or/and steps.
__device__ __forceinline__ uint32_t spread4x2(uint32_t b) // b: 8 bits = 4 states -> 4 bytes, UT folded into UO
{
    uint32_t y = (b | (b << 12)) & 0x000F000Fu;
    y = (y | (y << 6)) & 0x03030303u;
    return y | ((y >> 1) & 0x01010101u); // 2 -> 3, 3 stays 3, 0/1 unchanged
}
__device__ __forceinline__ uint32_t spread4x1(uint32_t n) // n: 4 bits = 4 states -> 4 bytes
{
    return (n | (n << 7) | (n << 14) | (n << 21)) & 0x01010101u;
}
__device__ __forceinline__ uint64_t expand8(uint32_t packed, uint32_t bits)
{
    if (bits == 2) return (uint64_t)spread4x2(packed & 0xffu) | ((uint64_t)spread4x2((packed >> 8) & 0xffu) << 32);
    return (uint64_t)spread4x1(packed & 0xfu) | ((uint64_t)spread4x1((packed >> 4) & 0xfu) << 32);
}

__device__ __forceinline__ uint32_t state_at(const uint8_t* p, uint32_t u, uint32_t bits)
{
    if (bits == 2) { uint32_t s = (p[u >> 2] >> ((u & 3u) << 1)) & 3u; return s == 2u ? 3u : s; }
    return (p[u >> 3] >> (u & 7u)) & 1u;
}

__global__ __launch_bounds__(256) void digest_items(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs,
                                                    const uint32_t* __restrict__ itemIds, uint32_t numItems, uint32_t level, uint32_t bits,
                                                    uint64_t* __restrict__ digests, const uint8_t* __restrict__ only, int want)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t it = gid >> 2, acc = gid & 3u;
    bool live = it < numItems;
    const uint32_t item = live ? itemIds[it] : 0u;
    if (live && only && (only[item] != 0) != (want != 0)) live = false;   // (streamed bakes: the items of one class only)
    const uint8_t* p = states + (live ? stateOfs[item] : 0ull);
    const uint32_t M = 1u << (2 * level);     // stream length in bytes
    const uint64_t seed = 42;
    uint64_t h;
    if (M >= 32u) {
        uint64_t v = acc == 0 ? seed + XP1 + XP2 : (acc == 1 ? seed + XP2 : (acc == 2 ? seed : seed - XP1));
        const uint32_t stripes = M >> 5;
        if (live) {
            if (bits == 2) {
                // stripe s = micro-triangles [32s, 32s+32) = packed bytes [8s, 8s+8); this lane takes bytes [8s+2acc, +2)
                const uint16_t* q = (const uint16_t*)p + acc;
                for (uint32_t s = 0; s < stripes; ++s) v = xxh_round(v, expand8((uint32_t)q[4 * s], 2));
            } else {
                const uint8_t* q = p + acc; // packed bytes [4s, 4s+4), one byte per accumulator lane
                for (uint32_t s = 0; s < stripes; ++s) v = xxh_round(v, expand8((uint32_t)q[4 * s], 1));
            }
        }
        // combine the four accumulators of this item (lanes 4k..4k+3)
        const uint32_t lane = threadIdx.x & 63u, l0 = lane & ~3u;
        const uint64_t v1 = __shfl(v, l0), v2 = __shfl(v, l0 + 1), v3 = __shfl(v, l0 + 2), v4 = __shfl(v, l0 + 3);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
        h += (uint64_t)M; // M is a multiple of 32: no tail
    } else {
        h = seed + XP5 + (uint64_t)M;
        if (live) {
            uint32_t u = 0;
            for (; u + 8 <= M; u += 8) {
                uint64_t w = 0;
                for (uint32_t k = 0; k < 8; ++k) w |= (uint64_t)state_at(p, u + k, bits) << (8 * k);
                h ^= xxh_round(0, w); h = rotl64(h, 27) * XP1 + XP4;
            }
            if (u + 4 <= M) {
                uint32_t w = 0;
                for (uint32_t k = 0; k < 4; ++k) w |= state_at(p, u + k, bits) << (8 * k);
                h ^= (uint64_t)w * XP1; h = rotl64(h, 23) * XP2 + XP3; u += 4;
            }
            for (; u < M; ++u) { h ^= (uint64_t)state_at(p, u, bits) * XP5; h = rotl64(h, 11) * XP1; }
        }
    }
    h = xxh_avalanche(h);
    if (live && acc == 0) digests[item] = h;
}

// Same digest for items of >= 256 packed bytes (4-state: level >= 5, 2-state: level >= 6).  In digest_items a wave reads 2-byte
// pieces of 16 items that lie 16 KiB apart, and rocprofv3 counted 4.6x the states' bytes in HBM fetches.  Here a workgroup
// (64 items x 4 accumulators) stages 256 packed bytes per item through LDS with 16-byte coalesced loads (row stride 65 words:
// the 64 rows start in different banks), then every lane walks its accumulator's pieces out of LDS.
// The walk over an item is sequential (XXH64's accumulators are chains), so a workgroup's run time is (bytes per item / chunk) round trips to HBM,
// however few items a launch has: with 256-byte chunks a 16 KiB item took 64 round trips (0.7 ms even for a handful of items, and the streamed bake
// digests range by range); items of >= 1 KiB are staged in 1 KiB chunks (65.8 KB of LDS per workgroup), 16 round trips.
// A streamed bake digests range by range, next to the persistent classification launch (29 KB of LDS and one wave slot per SIMD are free on a CU): a
// few thousand items per launch, so the launch takes as long as ONE workgroup -- 16 items x 1 KiB chunks (16 round trips for a 16 KiB item instead of
// 64, at raised wave priority) serve those; the hash lanes are the first 4 x ITEMS threads, all 256 load.
// (the small-workgroup forms run NEXT TO the persistent classification launch of a streamed bake -- up to six workgroups of <= 80 VGPRs per CU --:
//  like every kernel of the placement stream they stay small, 25 VGPRs with rolled loops)
template <int DG_CHUNK, int DG_ITEMS, bool MIXED>
__device__ __forceinline__ void digest_items_lds_body(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs, const DigestLists& D, uint32_t blocksA, uint32_t bits,
                                                      uint64_t* __restrict__ digests, uint32_t blk);
template <int DG_CHUNK, int DG_ITEMS, bool MIXED>
__global__ __launch_bounds__(256) void digest_items_lds(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs, DigestLists D, uint32_t blocksA, uint32_t bits,
                                                        uint64_t* __restrict__ digests)
{
    digest_items_lds_body<DG_CHUNK, DG_ITEMS, MIXED>(states, stateOfs, D, blocksA, bits, digests, blockIdx.x);
}
template <int DG_CHUNK, int DG_ITEMS, bool MIXED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) void digest_items_lds_guest(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs, DigestLists D,
                                                                                                   uint32_t blocksA, uint32_t bits, uint64_t* __restrict__ digests)
{
    digest_items_lds_body<DG_CHUNK, DG_ITEMS, MIXED>(states, stateOfs, D, blocksA, bits, digests, blockIdx.x);
}
template <int DG_CHUNK, int DG_ITEMS, bool MIXED>
__device__ __forceinline__ void digest_items_lds_body(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs, const DigestLists& D, uint32_t blocksA, uint32_t bits,
                                                      uint64_t* __restrict__ digests, uint32_t blk)
{
    constexpr int DG_STRIDE = DG_CHUNK / 4 + 1;
    __shared__ uint32_t s_buf[DG_ITEMS * DG_STRIDE];
    __shared__ const uint8_t* s_ptr[DG_ITEMS];
    __shared__ uint32_t s_bytes[MIXED ? DG_ITEMS : 1], s_maxBytes;
    // XXH64's round multiplies the 8 input bytes by P2 first.  The 8 bytes are the states of 8 micro-triangles, unpacked from two bytes of packed 2-bit states:
    // x = A + (B << 32) with A, B = the spread of one packed byte each, so x * P2 = A * P2 + ((B * P2) << 32) mod 2^64 -- two reads of a 256-entry table of
    // spread(b) * P2 instead of the unpack and four quarter-rate 32-bit multiplies (digest of configs[2]'s 127 k level-8 items: 0.76 -> see DESIGN 5).
    __shared__ uint64_t s_tab[256];
    const uint32_t tid = threadIdx.x;
    if (bits == 2) s_tab[tid & 255u] = (uint64_t)spread4x2(tid & 255u) * XP2;
    const bool listB = MIXED && blk >= blocksA;
    const uint32_t* itemIds = D.ids; uint32_t numItems = D.count, first = blk * DG_ITEMS;
    const uint8_t* only = D.only;
    if (listB) {
        numItems = *D.liveCount < D.capacityB ? *D.liveCount : D.capacityB; itemIds = D.listB + *D.liveStart; first = (blk - blocksA) * DG_ITEMS; only = nullptr;
    }
    if (first >= numItems) return;
    if (DG_ITEMS < 64) __builtin_amdgcn_s_setprio(3);   // (a latency-bound guest next to the classification: its few instructions should not queue behind 6 waves per SIMD)
    const uint32_t il = tid >> 2, acc = tid & 3u, it = first + il;
    const bool hasher = il < (uint32_t)DG_ITEMS;
    bool live = hasher && it < numItems;
    if (live && only && (only[itemIds[it]] != 0) != (D.want != 0)) live = false;   // (streamed bakes: the items of one class only)
    if (MIXED) { if (tid == 0) s_maxBytes = 0; __syncthreads(); }
    uint32_t level = D.level;
    if (tid < (uint32_t)DG_ITEMS) {
        const uint32_t j = first + tid;
        const bool take = j < numItems && !(only && (only[itemIds[j]] != 0) != (D.want != 0));
        s_ptr[tid] = take ? states + stateOfs[itemIds[j]] : nullptr;
        if (MIXED) { const uint32_t l = listB ? (uint32_t)D.itemLevel[itemIds[j < numItems ? j : first]] : level; const uint32_t b = take ? ((1u << (2u * l)) * bits) >> 3 : 0u; s_bytes[tid] = b; atomicMax(&s_maxBytes, b); }
    }
    if (listB && live) level = D.itemLevel[itemIds[it]];
    const uint32_t M = 1u << (2 * level);
    const uint32_t myBytes = (M * bits) >> 3;                 // multiple of DG_CHUNK (the launchers check)
    const uint32_t stripesPerChunk = DG_CHUNK / (4u * bits);  // a 32-byte stripe of the byte stream = 4*bits packed bytes
    const uint64_t seed = 42;
    uint64_t v = acc == 0 ? seed + XP1 + XP2 : (acc == 1 ? seed + XP2 : (acc == 2 ? seed : seed - XP1));
    __syncthreads();
    const uint32_t bytesPerItem = MIXED ? s_maxBytes : myBytes;
    // The full-size form fetches chunk k + 1 into registers while chunk k is hashed: with one workgroup per CU (the few thousand level-10 items of a mixed bake) the
    // latency of the fetch was half of every step.  The guest forms (32 VGPRs) fetch and store in one go.
    constexpr bool PREFETCH = DG_ITEMS == 64 && !MIXED;
    constexpr uint32_t LOADS = DG_ITEMS * (DG_CHUNK / 16) / 256;   // 16-byte pieces per thread and chunk
    uint4 pre[PREFETCH ? LOADS : 1];
    auto fetch = [&](uint32_t chunk) {
        #pragma unroll
        for (uint32_t j = 0; j < LOADS; ++j) {
            const uint32_t k = tid + j * 256u, row = k / (DG_CHUNK / 16), part = k % (DG_CHUNK / 16);
            const uint8_t* src = s_ptr[row];
            pre[PREFETCH ? j : 0] = src ? *(const uint4*)(src + chunk + part * 16u) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if (PREFETCH && bytesPerItem) fetch(0);
    for (uint32_t chunk = 0; chunk < bytesPerItem; chunk += DG_CHUNK) {
        if (PREFETCH) {
            #pragma unroll
            for (uint32_t j = 0; j < LOADS; ++j) {
                const uint32_t k = tid + j * 256u, row = k / (DG_CHUNK / 16), part = k % (DG_CHUNK / 16);
                uint32_t* dst = s_buf + row * DG_STRIDE + part * 4u;
                const uint4 w = pre[PREFETCH ? j : 0];
                dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
            }
            __syncthreads();
            if (chunk + DG_CHUNK < bytesPerItem) fetch(chunk + DG_CHUNK);
        } else {
        #pragma unroll DG_ITEMS < 64 ? 1 : 4   // (the guest forms keep their register count down: see digest_items_lds_guest)
        for (uint32_t k = tid; k < DG_ITEMS * (DG_CHUNK / 16); k += 256) {
            const uint32_t row = k / (DG_CHUNK / 16), part = k % (DG_CHUNK / 16);
            const uint8_t* src = s_ptr[row];
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (src && (!MIXED || chunk < s_bytes[row])) w = *(const uint4*)(src + chunk + part * 16u);
            uint32_t* dst = s_buf + row * DG_STRIDE + part * 4u;
            dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
        }
        __syncthreads();
        }
        if (!hasher || (MIXED && chunk >= myBytes)) { /* a load-only thread, or this item's stream has ended */ }
        else if (bits == 2) {
            const uint16_t* q = (const uint16_t*)(s_buf + il * DG_STRIDE) + acc;
            #pragma unroll DG_ITEMS < 64 ? 2 : 4
            for (uint32_t st = 0; st < stripesPerChunk; ++st) {
                const uint32_t piece = (uint32_t)q[4 * st];
                const uint64_t xp = s_tab[piece & 0xffu] + ((uint64_t)(uint32_t)s_tab[piece >> 8] << 32);
                v = rotl64(v + xp, 31) * XP1;
            }
        } else {
            const uint8_t* q = (const uint8_t*)(s_buf + il * DG_STRIDE) + acc;
            #pragma unroll DG_ITEMS < 64 ? 2 : 4
            for (uint32_t st = 0; st < stripesPerChunk; ++st) v = xxh_round(v, expand8((uint32_t)q[4 * st], 1));
        }
        __syncthreads();
    }
    const uint32_t lane = tid & 63u, l0 = lane & ~3u;
    const uint64_t v1 = __shfl(v, l0), v2 = __shfl(v, l0 + 1), v3 = __shfl(v, l0 + 2), v4 = __shfl(v, l0 + 3);
    uint64_t h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
    h += (uint64_t)M;
    h = xxh_avalanche(h);
    if (live && acc == 0) digests[itemIds[it]] = h;
}

// The same digest for FEW, LONG items: a launch whose workgroups are all resident at once takes as long as one of them, and one of them takes
// (stripes per item) x (the instructions of one XXH64 round in one wave, ~400 cycles: unpack the states, two 64-bit multiplies, add, rotate) -- 5.9 ms
// for the 32 768 stripes of a level-10 item however few there are (6 of the 136 ms of configs[4]; 20 % of an 8-rank share of it), 0.29 ms for a rank's
// share of level-8 items.  Only `v = rotl(v + x, 31) * P1` is a chain; x = unpack(states) * P2 is not.  So a workgroup of 16 items splits the work:
// wave 0 runs the 64 chains (item, accumulator) over x values it reads from LDS -- ~100 cycles per round --, the other three waves produce the x of
// the next 32 stripes and fetch the packed states of the 32 after those, one barrier per 32 stripes:
//     step k:   wave 0: chain over X[k & 1]      waves 1-3: issue loads of chunk k + 2; X[(k + 1) & 1] <- pack[(k + 1) & 1]; pack[k & 1] <- the loaded chunk
// Several levels in ONE launch (round 5): a bake of mixed levels used to run one of these launches per level, one after the other, each as long as its longest
// chain (configs[4]: five launches, 5.9 ms); the chains of different levels are independent, so the workgroups of all of them go into one grid, highest
// level first, and the launch takes as long as the level-10 chain alone.
constexpr int DL_ITEMS = 16, DL_STRIPES = 32;
struct DigestChainLevels { uint32_t n; uint32_t level[kNumLevels], count[kNumLevels], blockStart[kNumLevels + 1]; const uint32_t* ids[kNumLevels]; };
// MANY items of several levels in one launch of the form above (a workgroup holds 64 items of one level; highest level first, so the longest streams start first):
// the levels of a mixed bake used to take one launch each, one after the other, each as long as its longest item.
__global__ __launch_bounds__(256) void digest_items_lds_levels(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs, DigestChainLevels V, uint32_t bits,
                                                               uint64_t* __restrict__ digests)
{
    uint32_t seg = 0;
    while (seg + 1u < V.n && blockIdx.x >= V.blockStart[seg + 1u]) ++seg;
    DigestLists D; memset(&D, 0, sizeof D);
    D.ids = V.ids[seg]; D.count = V.count[seg]; D.level = V.level[seg];
    digest_items_lds_body<256, 64, false>(states, stateOfs, D, 0u, bits, digests, blockIdx.x - V.blockStart[seg]);
}
__global__ __launch_bounds__(256) void digest_items_chain(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs, DigestChainLevels V,
                                                          uint32_t bits, uint64_t* __restrict__ digests)
{
    __shared__ uint32_t s_pack[2][DL_ITEMS][64 + 1];            // packed states of DL_STRIPES stripes per item: 4 * bits bytes per stripe, <= 256 bytes
    __shared__ uint64_t s_x[2][DL_STRIPES][DL_ITEMS * 4];       // x[stripe][item * 4 + accumulator]
    __shared__ const uint8_t* s_ptr[DL_ITEMS];
    uint32_t seg = 0;
    while (seg + 1u < V.n && blockIdx.x >= V.blockStart[seg + 1u]) ++seg;
    const uint32_t* __restrict__ itemIds = V.ids[seg];
    const uint32_t numItems = V.count[seg], level = V.level[seg];
    const uint32_t tid = threadIdx.x, first = (blockIdx.x - V.blockStart[seg]) * DL_ITEMS;
    if (tid < (uint32_t)DL_ITEMS) s_ptr[tid] = first + tid < numItems ? states + stateOfs[itemIds[first + tid]] : nullptr;
    const uint32_t M = 1u << (2 * level);
    const uint32_t chunkBytes = 4u * bits * DL_STRIPES, quads = chunkBytes / 16u;   // 256 or 128 bytes per item and step
    const uint32_t steps = ((M * bits) >> 3) / chunkBytes;                          // (the launcher checks: a multiple, >= 2)
    const bool chain = tid < 64u;
    const uint32_t lt = tid - 64u;                                                  // loader / producer index 0 .. 191
    uint4 r0 = make_uint4(0u, 0u, 0u, 0u), r1 = r0;
    auto fetch = [&](uint32_t step) {
        const uint32_t k0 = lt, k1 = lt + 192u;
        if (k0 < DL_ITEMS * quads) { const uint8_t* src = s_ptr[k0 / quads]; r0 = src ? *(const uint4*)(src + (size_t)step * chunkBytes + (k0 % quads) * 16u) : make_uint4(0u, 0u, 0u, 0u); }
        if (k1 < DL_ITEMS * quads) { const uint8_t* src = s_ptr[k1 / quads]; r1 = src ? *(const uint4*)(src + (size_t)step * chunkBytes + (k1 % quads) * 16u) : make_uint4(0u, 0u, 0u, 0u); }
    };
    auto stash = [&](uint32_t buf) {
        const uint32_t k0 = lt, k1 = lt + 192u;
        if (k0 < DL_ITEMS * quads) { uint32_t* d = &s_pack[buf][k0 / quads][(k0 % quads) * 4u]; d[0] = r0.x; d[1] = r0.y; d[2] = r0.z; d[3] = r0.w; }
        if (k1 < DL_ITEMS * quads) { uint32_t* d = &s_pack[buf][k1 / quads][(k1 % quads) * 4u]; d[0] = r1.x; d[1] = r1.y; d[2] = r1.z; d[3] = r1.w; }
    };
    auto produce = [&](uint32_t buf) {
        for (uint32_t e = lt; e < DL_STRIPES * DL_ITEMS * 4u; e += 192u) {
            const uint32_t st = e >> 6, la = e & 63u, il = la >> 2, acc = la & 3u;
            const uint32_t piece = bits == 2 ? (uint32_t)((const uint16_t*)s_pack[buf][il])[4u * st + acc] : (uint32_t)((const uint8_t*)s_pack[buf][il])[4u * st + acc];
            s_x[buf][st][la] = expand8(piece, bits) * XP2;
        }
    };
    __syncthreads();
    if (!chain) { fetch(0); stash(0); }
    __syncthreads();
    if (!chain) { if (steps > 1u) fetch(1); produce(0); if (steps > 1u) stash(1); }
    __syncthreads();
    const uint32_t acc = tid & 3u;
    const uint64_t seed = 42;
    uint64_t v = acc == 0 ? seed + XP1 + XP2 : (acc == 1 ? seed + XP2 : (acc == 2 ? seed : seed - XP1));
    for (uint32_t k = 0; k < steps; ++k) {
        if (chain) {
            #pragma unroll 8
            for (uint32_t st = 0; st < (uint32_t)DL_STRIPES; ++st) { v += s_x[k & 1u][st][tid]; v = rotl64(v, 31); v *= XP1; }
        } else {
            if (k + 2u < steps) fetch(k + 2u);
            if (k + 1u < steps) produce((k + 1u) & 1u);
            if (k + 2u < steps) stash(k & 1u);
        }
        __syncthreads();
    }
    if (!chain) return;
    const uint32_t l0 = tid & ~3u;
    const uint64_t v1 = __shfl(v, l0), v2 = __shfl(v, l0 + 1), v3 = __shfl(v, l0 + 2), v4 = __shfl(v, l0 + 3);
    uint64_t h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
    h += (uint64_t)M;
    h = xxh_avalanche(h);
    const uint32_t it = first + (tid >> 2);
    if (it < numItems && acc == 0) digests[itemIds[it]] = h;
}

// few, long items (every workgroup resident at once: 16 384 items fill the chip, twice that still gains): the split form above
// FEW long items (a workgroup-step of the split form takes ~12 us whatever the number of workgroups up to 3 per CU; with the table-driven round of the plain form a lane
// alone needs ~110 cycles per stripe, so the split form pays only while the plain form would leave most of the chip idle)
#ifndef OMMX_DIGEST_CHAIN_MAX_ITEMS
#define OMMX_DIGEST_CHAIN_MAX_ITEMS 2048u
#endif
static bool digest_wants_chain(uint32_t numItems, uint32_t level, uint32_t bits) { return numItems != 0 && ((((1u << (2 * level)) * bits) >> 3) >= 1024u) && numItems <= OMMX_DIGEST_CHAIN_MAX_ITEMS; }
static bool digest_wants_lds(uint32_t numItems, uint32_t level, uint32_t bits) { return numItems != 0 && ((((1u << (2 * level)) * bits) >> 3) >= 256u) && !digest_wants_chain(numItems, level, bits); }
// the active items of ALL levels (level l: ids first[l] .. + count[l] of activeIds): the levels a form suits in one launch of it, the small levels one launch each
void launch_digest_levels(const uint8_t* states, const uint64_t* stateOfs, const uint32_t* activeIds, const uint32_t first[kNumLevels], const uint32_t count[kNumLevels],
                          uint32_t bits, uint64_t* digests, hipStream_t stream)
{
    DigestChainLevels V; memset(&V, 0, sizeof V);
    for (int l = kNumLevels - 1; l >= 0; --l) {   // (highest level first: the longest chains start first)
        if (!digest_wants_chain(count[l], (uint32_t)l, bits)) continue;
        V.level[V.n] = (uint32_t)l; V.count[V.n] = count[l]; V.ids[V.n] = activeIds + first[l];
        V.blockStart[V.n + 1u] = V.blockStart[V.n] + (count[l] + DL_ITEMS - 1u) / DL_ITEMS; V.n++;
    }
    if (V.n) hipLaunchKernelGGL(digest_items_chain, dim3(V.blockStart[V.n]), dim3(256), 0, stream, states, stateOfs, V, bits, digests);
    memset(&V, 0, sizeof V);
    for (int l = kNumLevels - 1; l >= 0; --l) {
        if (!digest_wants_lds(count[l], (uint32_t)l, bits)) continue;
        V.level[V.n] = (uint32_t)l; V.count[V.n] = count[l]; V.ids[V.n] = activeIds + first[l];
        V.blockStart[V.n + 1u] = V.blockStart[V.n] + (count[l] + 63u) / 64u; V.n++;
    }
    if (V.n) hipLaunchKernelGGL(digest_items_lds_levels, dim3(V.blockStart[V.n]), dim3(256), 0, stream, states, stateOfs, V, bits, digests);
    for (int l = 0; l < kNumLevels; ++l)
        if (count[l] && !digest_wants_chain(count[l], (uint32_t)l, bits) && !digest_wants_lds(count[l], (uint32_t)l, bits)) launch_digest(states, stateOfs, activeIds + first[l], count[l], (uint32_t)l, bits, digests, stream);
}
void launch_digest(const uint8_t* states, const uint64_t* stateOfs, const uint32_t* itemIds, uint32_t numItems, uint32_t level, uint32_t bits,
                   uint64_t* digests, hipStream_t stream, const uint8_t* only, int want)
{
    if (numItems == 0) return;
    const uint32_t bytesPerItem = ((1u << (2 * level)) * bits) >> 3;
    DigestLists D; memset(&D, 0, sizeof D); D.ids = itemIds; D.count = numItems; D.level = level; D.only = only; D.want = want;
    // (256-byte chunks x 64 items: a full-size launch is faster with them than with 1 KB chunks -- 0.77 vs 1.12 ms for 127 k items, more workgroups per CU
    //  hide the latency)
    if (digest_wants_chain(numItems, level, bits) && only == nullptr) {
        DigestChainLevels V; memset(&V, 0, sizeof V);
        V.n = 1; V.level[0] = level; V.count[0] = numItems; V.ids[0] = itemIds; V.blockStart[0] = 0; V.blockStart[1] = (numItems + DL_ITEMS - 1u) / DL_ITEMS;
        hipLaunchKernelGGL(digest_items_chain, dim3(V.blockStart[1]), dim3(256), 0, stream, states, stateOfs, V, bits, digests);
        return;
    }
    if (bytesPerItem >= 256u) { // (powers of two: a multiple of the chunk size)
        hipLaunchKernelGGL((digest_items_lds<256, 64, false>), dim3((numItems + 63u) / 64u), dim3(256), 0, stream, states, stateOfs, D, 0u, bits, digests);
        return;
    }
    const uint32_t threads = numItems * 4u;
    hipLaunchKernelGGL(digest_items, dim3((threads + 255u) / 256u), dim3(256), 0, stream, states, stateOfs, itemIds, numItems, level, bits, digests, only, want);
}

// streamed bakes: list A (a range's own items of one level >= 6, filtered) and list B (a device-side slice of the early list) in ONE small-workgroup launch
void launch_digest_lists(const uint8_t* states, const uint64_t* stateOfs, const DigestLists& D, uint32_t bits, uint64_t* digests, hipStream_t stream)
{
    if (D.count == 0 && D.capacityB == 0) return;
    if (bits == 2) {   // every item has >= 1 KiB of packed states
        const uint32_t a = (D.count + 15u) / 16u, b = (D.capacityB + 15u) / 16u;
        hipLaunchKernelGGL((digest_items_lds_guest<1024, 16, true>), dim3(a + b), dim3(256), 0, stream, states, stateOfs, D, a, bits, digests);
    } else {           // >= 512 bytes
        const uint32_t a = (D.count + 31u) / 32u, b = (D.capacityB + 31u) / 32u;
        hipLaunchKernelGGL((digest_items_lds_guest<512, 32, true>), dim3(a + b), dim3(256), 0, stream, states, stateOfs, D, a, bits, digests);
    }
}

// ------------------------------------------------------------------------------------------------
// Summed-area table of (alpha > cutoff) (texture_impl.cpp:191-220): indicator + row scan, then a
// column scan.  uint32 sums are exact, so any summation order gives the reference's table.
// ------------------------------------------------------------------------------------------------
template <bool FP32>
__global__ __launch_bounds__(256) void sat_rows(const void* __restrict__ texels, uint32_t* __restrict__ sat, int w, int h, float cutoff)
{
    // one wave per row, 64-texel chunks with a carried running sum
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= h) return;
    uint32_t carry = 0;
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        uint32_t v = 0;
        if (x < w) {
            const size_t idx = (size_t)x + (size_t)row * (size_t)w;
            const float a = FP32 ? ((const float*)texels)[idx] : (float)((const uint8_t*)texels)[idx] * (1.f / 255.f);
            v = a > cutoff ? 1u : 0u;
        }
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t n = __shfl_up(v, d); if (lane >= d) v += n; }
        v += carry;
        if (x < w) sat[(size_t)x + (size_t)row * (size_t)w] = v;
        carry = __shfl(v, 63);
    }
}

// column pass in three steps so that all of the chip works on it (a thread per column alone is 4096 threads walking 4096 rows: 1 ms at 4K):
//   sat_cols_block    thread = (column, block of SAT_ROWS rows): running sum inside the block, block total -> partial[block][column]
//   sat_cols_carry    thread = column: exclusive scan of that column's block totals (h / SAT_ROWS values)
//   sat_cols_add      thread = (column, block): add the carried total of the blocks above
constexpr int SAT_ROWS = 64;
__global__ __launch_bounds__(256) void sat_cols_block(uint32_t* __restrict__ sat, uint32_t* __restrict__ partial, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (x >= w) return;
    const int y0 = k * SAT_ROWS, y1 = y0 + SAT_ROWS < h ? y0 + SAT_ROWS : h;
    uint32_t run = 0;
    for (int y = y0; y < y1; ++y) { const size_t i = (size_t)x + (size_t)y * (size_t)w; run += sat[i]; sat[i] = run; }
    partial[(size_t)k * (size_t)w + (size_t)x] = run;
}
__global__ __launch_bounds__(256) void sat_cols_carry(uint32_t* __restrict__ partial, int w, int blocks)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    uint32_t run = 0;
    for (int k = 0; k < blocks; ++k) { const size_t i = (size_t)k * (size_t)w + (size_t)x; const uint32_t v = partial[i]; partial[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void sat_cols_add(uint32_t* __restrict__ sat, const uint32_t* __restrict__ partial, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (x >= w || k == 0) return;
    const uint32_t carry = partial[(size_t)k * (size_t)w + (size_t)x];
    const int y0 = k * SAT_ROWS, y1 = y0 + SAT_ROWS < h ? y0 + SAT_ROWS : h;
    for (int y = y0; y < y1; ++y) sat[(size_t)x + (size_t)y * (size_t)w] += carry;
}

size_t sat_scratch_bytes(int w, int h) { return sizeof(uint32_t) * (size_t)w * (size_t)((h + SAT_ROWS - 1) / SAT_ROWS); }

void launch_sat_build(const void* texels, int fp32, uint32_t* sat, uint32_t* scratch, int w, int h, float cutoff, hipStream_t stream)
{
    if (fp32) hipLaunchKernelGGL(sat_rows<true>, dim3((h + 3) / 4), dim3(256), 0, stream, texels, sat, w, h, cutoff);
    else      hipLaunchKernelGGL(sat_rows<false>, dim3((h + 3) / 4), dim3(256), 0, stream, texels, sat, w, h, cutoff);
    const int blocks = (h + SAT_ROWS - 1) / SAT_ROWS;
    const dim3 grid((w + 255) / 256, blocks);
    hipLaunchKernelGGL(sat_cols_block, grid, dim3(256), 0, stream, sat, scratch, w, h);
    hipLaunchKernelGGL(sat_cols_carry, dim3((w + 255) / 256), dim3(256), 0, stream, scratch, w, blocks);
    hipLaunchKernelGGL(sat_cols_add, grid, dim3(256), 0, stream, sat, (const uint32_t*)scratch, w, h);
}

// ------------------------------------------------------------------------------------------------
// Tail: gather the surviving OMMs into the final arrayData order, write descriptors and the index buffer.
// ------------------------------------------------------------------------------------------------
// one workgroup per emitted OMM: 16-byte vector copy of its packed states (sizes are powers of two); OMMs of items that
// were settled by triage (only emitted when special indices are disabled) are written as their constant pattern
__global__ __launch_bounds__(256) void tail_gather_omms(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs,
                                                        const uint8_t* __restrict__ active, const uint32_t* __restrict__ stateMask,
                                                        const uint8_t* __restrict__ level, int bits, int storeBits,
                                                        const uint32_t* __restrict__ order, const uint32_t* __restrict__ dstOfs,
                                                        const uint32_t* __restrict__ sizes, uint32_t numOmms, uint8_t* __restrict__ arrayData,
                                                        uint8_t* __restrict__ unitCodes, uint32_t* __restrict__ blockRawCounts, uint2* __restrict__ descs)
{
    for (uint32_t j = blockIdx.x; j < numOmms; j += gridDim.x) {
        const uint32_t item = order[j];
        // ommCpuOpacityMicromapDesc { u32 offset; u16 subdivisionLevel; u16 format; } (was a launch of its own: tail_descs)
        if (descs && threadIdx.x == 0) descs[j] = make_uint2(dstOfs[j], (uint32_t)level[item] | ((uint32_t)bits << 16));
        uint8_t* dst = arrayData + dstOfs[j];
        const uint32_t n = sizes[j];
        if (unitCodes) {
            // The compressed result transfer (omm_host.cpp): the exchange codec's code of every 16-byte unit and the number of raw units per 256-unit block are
            // produced HERE, where the unit is in a register anyway, instead of by a pass of their own over the finished array.  The caller guarantees: every OMM is
            // a multiple of 16 bytes (no level below 3; 4 in 2-state) and storeBits == bits.
            const bool uniform = !active[item];
            uint32_t pat = 0;
            if (uniform) {
                const uint32_t st = (uint32_t)(31 - __clz((int)stateMask[item]));
                for (uint32_t b = 0; b < 8u; b += (uint32_t)bits) pat |= st << b;
                pat *= 0x01010101u;
            }
            const uint4* s4 = (const uint4*)(states + (uniform ? 0ull : stateOfs[item])); uint4* d4 = (uint4*)dst;
            const uint32_t units = n / 16u, u0 = dstOfs[j] / 16u;
            for (uint32_t k0 = 0; k0 < units; k0 += blockDim.x) {
                const uint32_t k = k0 + threadIdx.x;
                const bool live = k < units;
                uint4 v = make_uint4(pat, pat, pat, pat);
                if (live && !uniform) v = s4[k];
                const uint32_t code = codec_unit_code(v.x, v.y, v.z, v.w);
                if (live) { d4[k] = v; unitCodes[u0 + k] = (uint8_t)code; }
                // raw units per codec block: the units of a wave are consecutive, so they lie in at most two blocks
                const uint32_t blk = (u0 + k) / kCodecBlockUnits;
                unsigned long long todo = __ballot(live && code == 4u);
                while (todo) {
                    const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
                    const uint32_t b0 = (uint32_t)__shfl((int)blk, (int)leader);
                    const unsigned long long same = __ballot(live && code == 4u && blk == b0) & todo;
                    if ((threadIdx.x & 63u) == leader) atomicAdd(&blockRawCounts[b0], (uint32_t)__popcll(same));
                    todo &= ~same;
                }
            }
            continue;
        }
        if (!active[item]) {
            const uint32_t st = (uint32_t)(31 - __clz((int)stateMask[item]));
            uint32_t usedBits = (1u << (2u * level[item])) * (uint32_t)bits; if (usedBits > 8u) usedBits = 8u;
            uint32_t pat = 0;
            for (uint32_t b = 0; b < usedBits; b += (uint32_t)bits) pat |= st << b;
            for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) dst[k] = (uint8_t)pat;
            continue;
        }
        const uint8_t* src = states + stateOfs[item];
        if (storeBits != bits) {   // 2-bit states -> the reference's 1-bit packing of a 2-state bake: byte |= (uint8_t)(state << (i & 7)), states 0 / 1 / 3
            const uint32_t M = 1u << (2u * level[item]);
            for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
                uint32_t v = 0;
                for (uint32_t j = 0; j < 8u && 8u * k + j < M; ++j) { const uint32_t u = 8u * k + j; v |= ((uint32_t)(src[u >> 2] >> ((u & 3u) << 1)) & 3u) << j; }
                dst[k] = (uint8_t)v;
            }
            continue;
        }
        if (n >= 16u) {
            const uint4* s4 = (const uint4*)src; uint4* d4 = (uint4*)dst;
            for (uint32_t k = threadIdx.x; k < n / 16u; k += blockDim.x) d4[k] = s4[k];
        } else {
            if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
        }
    }
}

void launch_gather_omms(const uint8_t* states, const uint64_t* stateOfs, const uint8_t* active, const uint32_t* stateMask, const uint8_t* level, int bits, int storeBits,
                        const uint32_t* order, const uint32_t* dstOfs, const uint32_t* sizes, uint32_t numOmms, uint8_t* arrayData, hipStream_t stream,
                        uint8_t* unitCodes, uint32_t* blockRawCounts, void* descs)
{
    if (numOmms == 0) return;
    const uint32_t grid = numOmms < 65536u * 4u ? numOmms : 65536u * 4u;
    if (storeBits != bits || !blockRawCounts) unitCodes = nullptr;
    hipLaunchKernelGGL(tail_gather_omms, dim3(grid), dim3(256), 0, stream, states, stateOfs, active, stateMask, level, bits, storeBits, order, dstOfs, sizes, numOmms, arrayData,
                       unitCodes, blockRawCounts, (uint2*)descs);
}

// index buffer: triangle -> unique work item -> dedup representative -> special index or descriptor slot
// (bake_cpu_impl.cpp:1856-1870)
__global__ __launch_bounds__(256) void tail_write_indices(const int32_t* __restrict__ triToItem, const uint32_t* __restrict__ rep,
                                                          const int32_t* __restrict__ itemValue, uint32_t numTris, int32_t unresolved,
                                                          int32_t* __restrict__ out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= numTris) return;
    const int32_t it = triToItem[t];
    out[t] = it < 0 ? unresolved : itemValue[rep[it]];
}

void launch_write_indices(const int32_t* triToItem, const uint32_t* rep, const int32_t* itemValue, uint32_t numTris, int32_t unresolved,
                          int32_t* out, hipStream_t stream)
{
    if (numTris == 0) return;
    hipLaunchKernelGGL(tail_write_indices, dim3((numTris + 255u) / 256u), dim3(256), 0, stream, triToItem, rep, itemValue, numTris, unresolved, out);
}

} // namespace ommx
