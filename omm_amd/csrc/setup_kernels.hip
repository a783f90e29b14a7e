// setup_kernels.hip -- SetupWorkItems (bake_cpu_impl.cpp:589-660) on the device.
//
// The reference walks the triangles serially: fetch UVs, pick the subdivision level, drop invalid (NaN/Inf) triangles,
// and merge triangles with the same 64-bit work-item id (vm_id.h: the reference's hash chain over (UV triangle, level, format), which its map
// trusts -- triangles whose ids collide are ONE work item there, and so they are here) into one work item owned by the FIRST occurrence.
// Device form: one lane per triangle + a hash build over the ids (hash_build.h):
//   * slot value = smallest triangle index with the key  = first occurrence
//   * exclusive scan over "is first occurrence" in triangle order = work-item numbering in the reference's order
//   * stable counting split of the items by level = the per-level launch lists
// (Rounds 1 - 5 keyed the table by an own hash of the tuple, compared the tuples of merged triangles and redid the setup on the host when they
// differed: tuple equality, which is what the reference computes only as long as its ids do not collide.  tests/golden/vmid_collisions.json holds inputs where they do.)
// The one libm-dependent piece -- log2f in the edge heuristic for degenerate triangles under dynamic subdivision
// (bake_cpu_impl.cpp:511-528) -- is not evaluated here: such triangles are reported in `pending` for the host.
#include <hip/hip_runtime.h>
#include <string.h>

#include "hash_build.h"
#include "vm_id.h"
#include "scan_lookback.h"
#include <stdint.h>
#include "bake_types.h"
#include "bake_kernels.h"

namespace ommx {

#define SETUP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

__device__ __forceinline__ float half_bits_to_float(uint32_t h) // glm::unpackHalf2x16 element (IEEE binary16 incl. denormals)
{
    const uint32_t sign = (h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { e = 1; while (!(m & 0x400u)) { m <<= 1; e--; } m &= 0x3ffu; bits = sign | ((e + 112u) << 23) | (m << 13); }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    return __uint_as_float(bits);
}

// x86: uint(float) goes through cvttss2si r64 and keeps the low 32 bits (bake_cpu_impl.cpp:502)
__device__ __forceinline__ uint32_t cvt_u32_x86(float f)
{
    return (f >= -9223372036854775808.f && f < 9223372036854775808.f) ? (uint32_t)(long long)f : 0u;
}

__global__ __launch_bounds__(256) void setup_fetch(SetupParams S, float* __restrict__ triUv, uint8_t* __restrict__ triLevel,
                                                   uint8_t* __restrict__ triFlags, uint64_t* __restrict__ hashKeys,
                                                   SetupCounters* __restrict__ counters, uint32_t* __restrict__ pendingList,
                                                   uint32_t* __restrict__ ones, uint32_t onesWords, uint32_t* __restrict__ zero, uint32_t zeroWords,
                                                   uint32_t* __restrict__ zero2, uint32_t zero2Words, uint32_t* __restrict__ zero3, uint32_t zero3Words,
                                                   float* __restrict__ triArea)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    // fills that later launches need (each was a fill launch or two of its own): the UV-dedup hash table to all ones, the caller's words to zero
    for (uint32_t w = t; w < onesWords; w += gridDim.x * blockDim.x) ones[w] = 0xFFFFFFFFu;
    for (uint32_t w = t; w < zeroWords; w += gridDim.x * blockDim.x) zero[w] = 0u;
    for (uint32_t w = t; w < zero2Words; w += gridDim.x * blockDim.x) zero2[w] = 0u;
    for (uint32_t w = t; w < zero3Words; w += gridDim.x * blockDim.x) zero3[w] = 0u;   // (setup_dedup_lookup's scan states)
    if (t >= S.numTris) return;
    // ---- FetchUVTriangle (util/geometry.h:191-239) ----
    uint32_t idx[3];
    const size_t o = 3ull * t;
    #pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (S.indexFormat == 2) idx[k] = ((const uint8_t*)S.indices)[o + k];
        else if (S.indexFormat == 0) idx[k] = ((const uint16_t*)S.indices)[o + k];
        else idx[k] = ((const uint32_t*)S.indices)[o + k];
    }
    float p[6];
    #pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint8_t* base = (const uint8_t*)S.texCoords + (size_t)S.stride * idx[k];
        if (S.uvFormat == 2) { // stride is caller-defined: assemble from bytes when it is not 4-aligned
            if (((uintptr_t)base & 3u) == 0) { p[2 * k] = ((const float*)base)[0]; p[2 * k + 1] = ((const float*)base)[1]; }
            else { uint32_t a = 0, b = 0; for (int q = 0; q < 4; ++q) { a |= (uint32_t)base[q] << (8 * q); b |= (uint32_t)base[4 + q] << (8 * q); } p[2 * k] = __uint_as_float(a); p[2 * k + 1] = __uint_as_float(b); }
        } else {
            uint32_t v = 0; for (int q = 0; q < 4; ++q) v |= (uint32_t)base[q] << (8 * q);
            if (S.uvFormat == 0) { p[2 * k] = (float)(v & 0xffffu) * 1.5259021896696421759314870504694e-5f; p[2 * k + 1] = (float)(v >> 16) * 1.5259021896696421759314870504694e-5f; }
            else { p[2 * k] = half_bits_to_float(v & 0xffffu); p[2 * k + 1] = half_bits_to_float(v >> 16); }
        }
    }
    bool invalid = false;
    #pragma unroll
    for (int k = 0; k < 6; ++k) invalid |= !(__builtin_fabsf(p[k]) < __builtin_inff()); // NaN or Inf (util/geometry.h:37-42)
    // util/geometry.h:44-47
    const float area0 = 0.5f * __builtin_fabsf(p[0] * (p[3] - p[5]) + p[2] * (p[5] - p[1]) + p[4] * (p[1] - p[3]));
    const bool degenerate = (double)area0 < 1e-9;
    invalid |= S.degenerateInvalid && degenerate;   // GetIsInvalid (bake_cpu_impl.cpp:563-575): without the level-line kernel degenerate triangles are not baked at all
    // ---- GetSubdivisionLevelForPrimitive (bake_cpu_impl.cpp:542-560) ----
    uint32_t level; bool pending = false;
    if (S.perTriLevels && S.perTriLevels[t] <= 12) level = S.perTriLevels[t];
    else if (S.dynScale > 0.f) {
        if (degenerate || S.edgeHeuristic) { level = 0; pending = !invalid; }
        else { // ComputeAreaHeuristic (:470-509)
            const float fw = (float)(uint32_t)S.texW, fh = (float)(uint32_t)S.texH;
            const float ax = p[0] * fw, ay = p[1] * fh, bx = p[2] * fw, by = p[3] * fh, cx = p[4] * fw, cy = p[5] * fh;
            const float v0x = cx - ax, v0y = cy - ay, v1x = bx - ax, v1y = by - ay;
            const float nx = v0y * 0.f - v1y * 0.f, ny = 0.f * v1x - 0.f * v0x, nz = v0x * v1y - v1x * v0y;
            const float pixelArea = 0.5f * __builtin_sqrtf(nx * nx + ny * ny + nz * nz);
            uint32_t v = cvt_u32_x86(pixelArea / (S.dynScale * S.dynScale));
            v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
            uint32_t r = (v & 0xAAAAAAAAu) != 0;
            r |= (uint32_t)((v & 0xFFFF0000u) != 0) << 4; r |= (uint32_t)((v & 0xFF00FF00u) != 0) << 3;
            r |= (uint32_t)((v & 0xF0F0F0F0u) != 0) << 2; r |= (uint32_t)((v & 0xCCCCCCCCu) != 0) << 1;
            level = r >> 1; if (level > (uint32_t)S.globalLevel) level = (uint32_t)S.globalLevel;
        }
    } else level = (uint32_t)S.globalLevel;

    #pragma unroll
    for (int k = 0; k < 6; ++k) triUv[6ull * t + k] = p[k];
    triLevel[t] = (uint8_t)level;
    triFlags[t] = (uint8_t)((invalid ? 1u : 0u) | (degenerate ? 2u : 0u) | (pending ? 4u : 0u));
    if (invalid) atomicAdd(&counters->numDisabled, 1u);
    if (pending) { const uint32_t slot = atomicAdd(&counters->numPending, 1u); pendingList[slot] = t; }
    // the reference's work-item id (vm_id.h).  Invalid triangles and bakes without duplicate detection never look at the table (setup_dedup_*).
    hashKeys[t] = (invalid || S.disableDedup) ? (uint64_t)t : vm_id(p, (int32_t)level, S.format);
    // UV-space area of the input triangle (bake_cpu_impl.cpp:1904-1915, GetArea2D util/geometry.h:141-149): the side channel of ommDebugGetStats2's
    // knownAreaMetric.  Triangles that own no work item (NaN / Inf coordinates) keep area 0 like the reference's value-initialised vector.
    if (triArea) {
        float a = 0.f;
        if (!invalid) {
            const float v0x = p[4] - p[0], v0y = p[5] - p[1], v1x = p[2] - p[0], v1y = p[3] - p[1];
            const float nx = v0y * 0.f - v1y * 0.f, ny = 0.f * v1x - 0.f * v0x, nz = v0x * v1y - v1x * v0y;   // glm::cross(float3(v0, 0), float3(v1, 0))
            a = 0.5f * __builtin_sqrtf(nx * nx + ny * ny + nz * nz);
        }
        triArea[t] = a;
    }
}

// after the host has filled in the levels of the pending triangles: recompute their hash keys
__global__ __launch_bounds__(256) void setup_rehash_pending(SetupParams S, const float* __restrict__ triUv, const uint8_t* __restrict__ triLevel,
                                                            const uint32_t* __restrict__ pendingList, uint32_t numPending, uint64_t* __restrict__ hashKeys)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= numPending) return;
    const uint32_t t = pendingList[k];
    if (S.disableDedup) return;
    float p[6];
    for (int q = 0; q < 6; ++q) p[q] = triUv[6ull * t + q];
    hashKeys[t] = vm_id(p, (int32_t)triLevel[t], S.format);
}

// ---- UV-triangle dedup as a hash build (hash_build.h): firstTri[t] = smallest triangle index with triangle t's 64-bit key ----
// (keys: setup_fetch; invalid triangles and disableDedup bakes have unique keys by construction and skip the table)
__global__ __launch_bounds__(256) void setup_dedup_insert(const uint64_t* __restrict__ hashKeys, const uint8_t* __restrict__ triFlags, uint32_t n, HashTable table)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t buckets[256];
    const bool live = t < n && !(triFlags[t] & 1u);
    hash_put_min_block(table, live, live ? hashKeys[t] : 0ull, t, buckets);
}

// first occurrence and item flag: a triangle whose id equals an earlier triangle's belongs to that triangle's work item (bake_cpu_impl.cpp:633-649)
// ... and (round 5) the item numbers: the exclusive scan of the item flags in the same launch -- single pass with decoupled look-back (a tile publishes its
// count with flag 1, collects the counts of the tiles before it until one with an inclusive prefix, flag 2; tiles handed out by ticket; state word =
// flag << 62 | count; `scanState`: ticket word, then the states from word 64 on -- zeroed by setup_fetch).  It was a rocPRIM scan of two launches.
__global__ __launch_bounds__(1024) void setup_dedup_lookup(const uint64_t* __restrict__ hashKeys, const uint8_t* __restrict__ triFlags, uint32_t n, HashTable table,
                                                          int disableDedup, const float* __restrict__ triUv, const uint8_t* __restrict__ triLevel,
                                                          uint32_t* __restrict__ firstTri, uint32_t* __restrict__ isItem, uint32_t* __restrict__ itemOfTri,
                                                          uint32_t* __restrict__ scanState, SetupCounters* __restrict__ counters)
{
    __shared__ uint32_t s_tile, s_wave[16], s_excl;
    if (threadIdx.x == 0) s_tile = atomicAdd(scanState, 1u);
    __syncthreads();
    const uint32_t tile = s_tile, t = tile * 1024u + threadIdx.x;
    if (tile * 1024u >= n) return;   // (uniform per workgroup)
    uint32_t item = 0;
    if (t < n) {
        const bool invalid = (triFlags[t] & 1u) != 0;
        uint32_t f = t;
        if (!invalid && !disableDedup) {
            f = hash_get(table, hashKeys[t], t);
        }
        firstTri[t] = f;
        item = (f == t && !invalid) ? 1u : 0u;
        isItem[t] = item;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(item != 0u);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t waveBase = 0, total = 0;
    for (uint32_t w = 0; w < 16u; ++w) { const uint32_t c = s_wave[w]; if (w < wave) waveBase += c; total += c; }
    if (wave == 0) { const unsigned long long excl = lookback_exclusive((unsigned long long*)(scanState + 64), tile, total); if (lane == 0) s_excl = (uint32_t)excl; }
    __syncthreads();
    if (t < n) itemOfTri[t] = s_excl + waveBase + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(256) void setup_emit_items(SetupParams S, const float* __restrict__ triUv, const uint8_t* __restrict__ triLevel,
                                                        const uint8_t* __restrict__ triFlags, const uint32_t* __restrict__ firstTri,
                                                        const uint32_t* __restrict__ isItem, const uint32_t* __restrict__ itemOfTri,
                                                        float* __restrict__ itemUv, uint8_t* __restrict__ itemLevel, uint8_t* __restrict__ itemDegenerate,
                                                        int32_t* __restrict__ triToItem, SetupCounters* __restrict__ counters)
{
    __shared__ uint32_t s_hist[kNumLevels];
    __shared__ unsigned long long s_work;
    if (threadIdx.x < kNumLevels) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_work = 0;
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < S.numTris) {
        if (t == S.numTris - 1) counters->numItems = itemOfTri[t] + isItem[t];
        triToItem[t] = (triFlags[t] & 1u) ? -1 : (int32_t)itemOfTri[firstTri[t]];
        if (isItem[t]) {
            const uint32_t i = itemOfTri[t];
            #pragma unroll
            for (int k = 0; k < 6; ++k) itemUv[6ull * i + k] = triUv[6ull * t + k];
            const uint32_t lvl = triLevel[t];
            itemLevel[i] = (uint8_t)lvl; itemDegenerate[i] = (triFlags[t] >> 1) & 1u;
            atomicAdd(&s_hist[lvl], 1u);
            if (S.wantWorkload) { // ComputeWorkloadSize (bake_cpu_impl.cpp:662-680)
                const float* p = triUv + 6ull * t;
                const float lox = (p[2] < p[0] ? p[2] : p[0]), lox2 = (p[4] < lox ? p[4] : lox), loy = (p[3] < p[1] ? p[3] : p[1]), loy2 = (p[5] < loy ? p[5] : loy);
                const float hix = (p[0] < p[2] ? p[2] : p[0]), hix2 = (hix < p[4] ? p[4] : hix), hiy = (p[1] < p[3] ? p[3] : p[1]), hiy2 = (hiy < p[5] ? p[5] : hiy);
                const float dx = (hix2 - lox2) * (float)S.texW, dy = (hiy2 - loy2) * (float)S.texH;
                const int ax = (dx >= -2147483648.f && dx < 2147483648.f) ? (int)dx : (int)0x80000000;
                const int ay = (dy >= -2147483648.f && dy < 2147483648.f) ? (int)dy : (int)0x80000000;
                atomicAdd(&s_work, (unsigned long long)(long long)(int)((uint32_t)ax * (uint32_t)ay));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kNumLevels && s_hist[threadIdx.x]) atomicAdd(&counters->levelCount[threadIdx.x], s_hist[threadIdx.x]);
    if (threadIdx.x == 0 && s_work) atomicAdd((unsigned long long*)&counters->workload, s_work);
}

// ---- items grouped by level (stable): a counting split over the 13 levels instead of a sort ----
// Items are numbered in triangle order; itemIds lists them level by level, ascending inside a level.  Chunks of kSplitChunk consecutive
// items: (1) per-chunk counts per level, (2) one small scan gives every chunk its write position per level (behind levelStart[l]),
// (3) the chunks scatter their items, ranked inside the chunk with wave ballots in item order.
constexpr uint32_t kSplitChunk = 4096;
__device__ __forceinline__ uint32_t level_of_slot(const uint8_t* __restrict__ itemLevel, uint32_t i, uint32_t numItems) { return i < numItems ? (uint32_t)itemLevel[i] : 0xFFu; }

__global__ __launch_bounds__(256) void setup_split_count(const uint8_t* __restrict__ itemLevel, const SetupCounters* __restrict__ counters, uint32_t* __restrict__ chunkCount)
{
    __shared__ uint32_t h[kNumLevels];
    if (threadIdx.x < kNumLevels) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t numItems = counters->numItems, base = blockIdx.x * kSplitChunk;
    if (base < numItems) {
        for (uint32_t j = 0; j < kSplitChunk; j += 256) {
            const uint32_t lvl = level_of_slot(itemLevel, base + j + threadIdx.x, numItems);
            unsigned long long todo = __ballot(lvl != 0xFFu);
            while (todo) {   // one LDS atomic per (wave, level)
                const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
                const uint32_t l0 = (uint32_t)__shfl((int)lvl, (int)leader);
                const unsigned long long same = __ballot(lvl == l0);
                if ((threadIdx.x & 63u) == leader) atomicAdd(&h[l0], (uint32_t)__popcll(same));
                todo &= ~same;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kNumLevels) chunkCount[blockIdx.x * kNumLevels + threadIdx.x] = h[threadIdx.x];
}

// levelStart[] from the level histogram, then chunkCount[c][l] -> first write position of chunk c in level l.  One wave per level, 64
// chunks per step (wave prefix sum).
__global__ __launch_bounds__(1024) void setup_split_scan(SetupCounters* __restrict__ counters, uint32_t* __restrict__ chunkCount, uint32_t numChunks)
{
    __shared__ uint32_t start[kNumLevels + 1];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int l = 0; l < kNumLevels; ++l) { start[l] = run; counters->levelStart[l] = run; run += counters->levelCount[l]; }
        start[kNumLevels] = run; counters->levelStart[kNumLevels] = run;
    }
    __syncthreads();
    const uint32_t l = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (l >= (uint32_t)kNumLevels) return;
    uint32_t run = start[l];
    for (uint32_t c0 = 0; c0 < numChunks; c0 += 64) {
        const uint32_t c = c0 + lane;
        const uint32_t v = c < numChunks ? chunkCount[c * kNumLevels + l] : 0u;
        uint32_t incl = v;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += o; }
        if (c < numChunks) chunkCount[c * kNumLevels + l] = run + incl - v;
        run += (uint32_t)__shfl((int)incl, 63);
    }
}

__global__ __launch_bounds__(256) void setup_split_scatter(const uint8_t* __restrict__ itemLevel, const SetupCounters* __restrict__ counters,
                                                           const uint32_t* __restrict__ chunkStart, uint32_t* __restrict__ itemIds)
{
    __shared__ uint32_t pos[kNumLevels];            // next write position of this chunk per level
    __shared__ uint32_t waveCount[4][kNumLevels];   // items of each level held by each wave in the current round
    const uint32_t numItems = counters->numItems, base = blockIdx.x * kSplitChunk;
    if (base >= numItems) return;   // (block-uniform)
    if (threadIdx.x < kNumLevels) pos[threadIdx.x] = chunkStart[blockIdx.x * kNumLevels + threadIdx.x];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t j = 0; j < kSplitChunk; j += 256) {
        if (threadIdx.x < 4 * kNumLevels) (&waveCount[0][0])[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t i = base + j + threadIdx.x;
        const uint32_t lvl = level_of_slot(itemLevel, i, numItems);
        uint32_t rank = 0;
        unsigned long long todo = __ballot(lvl != 0xFFu);
        while (todo) {
            const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t l0 = (uint32_t)__shfl((int)lvl, (int)leader);
            const unsigned long long same = __ballot(lvl == l0);
            if (lvl == l0) rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            if (lane == leader) waveCount[wave][l0] = (uint32_t)__popcll(same);
            todo &= ~same;
        }
        __syncthreads();
        if (lvl != 0xFFu) {
            uint32_t before = 0;
            for (uint32_t w = 0; w < wave; ++w) before += waveCount[w][lvl];
            itemIds[pos[lvl] + before + rank] = i;
        }
        __syncthreads();
        if (threadIdx.x < kNumLevels) pos[threadIdx.x] += waveCount[0][threadIdx.x] + waveCount[1][threadIdx.x] + waveCount[2][threadIdx.x] + waveCount[3][threadIdx.x];
        __syncthreads();
    }
}

static uint32_t setup_scan_words(uint32_t numTris) { return 64u + 2u * ((numTris + 1023u) / 1024u + 1u); }

size_t setup_scratch_bytes(uint32_t numTris)
{
    const size_t n = numTris ? numTris : 1;
    const size_t m = (size_t)setup_scan_words((uint32_t)n) * 4;   // setup_dedup_lookup's scan states
    const size_t p256 = 256;
    auto pad = [&](size_t v) { return (v + p256 - 1) / p256 * p256; };
    //           triUv        keys         firstTri,isItem,itemOfTri  level,flags   pending      chunk counts  hash table
    return pad(n * 24) + pad(n * 8) + 3 * pad(n * 4) + 2 * pad(n) + pad(n * 4) + pad(n * 4) + pad(hash_table_bytes(hash_table_slots(n))) + pad(m) + 1024;
}

// phase A: fetch + levels + hash keys.  The caller then looks at counters->numPending (after its sync) and, if non-zero,
// fixes those levels and calls setup_rehash().  phase B: dedup + item emission + level grouping.
struct SetupScratch {
    float* triUv; uint64_t* keysA; uint32_t *firstTri, *isItem, *itemOfTri, *pending, *lkeysA;
    uint8_t *triLevel, *triFlags; void* hash; void* tmp; size_t tmpBytes;
};
static SetupScratch carve_setup(void* base, size_t bytes, uint32_t numTris)
{
    const size_t n = numTris ? numTris : 1;
    auto pad = [](size_t v) { return (v + 255) / 256 * 256; };
    SetupScratch s; uint8_t* p = (uint8_t*)base;
    s.triUv = (float*)p; p += pad(n * 24);
    s.keysA = (uint64_t*)p; p += pad(n * 8);
    s.firstTri = (uint32_t*)p; p += pad(n * 4);
    s.isItem = (uint32_t*)p; p += pad(n * 4); s.itemOfTri = (uint32_t*)p; p += pad(n * 4);
    s.triLevel = p; p += pad(n); s.triFlags = p; p += pad(n);
    s.pending = (uint32_t*)p; p += pad(n * 4);
    s.lkeysA = (uint32_t*)p; p += pad(n * 4);
    s.hash = p; p += pad(hash_table_bytes(hash_table_slots(n)));
    s.tmp = p; s.tmpBytes = bytes - (size_t)(p - (uint8_t*)base);
    return s;
}

hipError_t run_setup_fetch(const SetupParams& S, void* scratch, size_t scratchBytes, SetupCounters* counters, uint32_t* zero, uint32_t zeroWords, uint32_t* zero2, uint32_t zero2Words,
                           float* triArea, hipStream_t stream)
{
    // (`counters` arrives zeroed: the bake's first transfer carries the block)
    if (S.numTris == 0) {
        if (zeroWords) SETUP_CHECK(hipMemsetAsync(zero, 0, (size_t)zeroWords * 4, stream));
        if (zero2Words) SETUP_CHECK(hipMemsetAsync(zero2, 0, (size_t)zero2Words * 4, stream));
        return hipSuccess;
    }
    if (scratchBytes < setup_scratch_bytes(S.numTris)) return hipErrorInvalidValue;
    SetupScratch s = carve_setup(scratch, scratchBytes, S.numTris);
    const uint32_t onesWords = S.disableDedup ? 0u : (uint32_t)(hash_table_bytes(hash_table_slots(S.numTris)) / 4);
    hipLaunchKernelGGL(setup_fetch, dim3((S.numTris + 255u) / 256u), dim3(256), 0, stream, S, s.triUv, s.triLevel, s.triFlags, s.keysA, counters, s.pending,
                       (uint32_t*)s.hash, onesWords, zero, zeroWords, zero2, zero2Words,
                       (uint32_t*)s.tmp, setup_scan_words(S.numTris), triArea);
    return hipGetLastError();
}

// host-computed levels for the pending (degenerate, dynamic) triangles: levels[k] belongs to triangle pendingHost[k]
// pending = triangles whose level needs glibc's log2f (degenerate triangles under dynamic subdivision).  Usually a handful, but a
// mesh of tiny triangles can have tens of thousands: their UVs and levels move in ONE transfer each way.
__global__ __launch_bounds__(256) void setup_gather_pending(const float* __restrict__ triUv, const uint32_t* __restrict__ pending, uint32_t n, float* __restrict__ out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* src = triUv + 6ull * pending[k];
    #pragma unroll
    for (int j = 0; j < 6; ++j) out[6ull * k + j] = src[j];
}
__global__ __launch_bounds__(256) void setup_scatter_levels(const uint32_t* __restrict__ pending, const uint8_t* __restrict__ levels, uint32_t n, uint8_t* __restrict__ triLevel)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) triLevel[pending[k]] = levels[k];
}

// (tmp: device memory of the caller, >= 24 bytes per pending triangle, shared with copy_pending_to_host -- a hipMalloc / hipFree pair here cost a bake with
//  pending triangles 0.7 ms each)
hipError_t run_setup_fix_pending(const SetupParams& S, void* scratch, size_t scratchBytes, const uint32_t* pendingTris /*host*/, const uint8_t* levels /*host*/,
                                 uint32_t numPending, void* tmp, hipStream_t stream)
{
    if (numPending == 0) return hipSuccess;
    SetupScratch s = carve_setup(scratch, scratchBytes, S.numTris);
    uint8_t* dLevels = (uint8_t*)tmp;
    hipError_t e = hipMemcpyAsync(dLevels, levels, numPending, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s.pending, pendingTris, (size_t)numPending * 4, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        const dim3 grid((numPending + 255u) / 256u), block(256);
        hipLaunchKernelGGL(setup_scatter_levels, grid, block, 0, stream, s.pending, dLevels, numPending, s.triLevel);
        hipLaunchKernelGGL(setup_rehash_pending, grid, block, 0, stream, S, s.triUv, s.triLevel, s.pending, numPending, s.keysA);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);   // (the host arrays are the caller's temporaries)
    return e;
}

hipError_t copy_pending_to_host(void* scratch, size_t scratchBytes, uint32_t numTris, uint32_t numPending, uint32_t* pendingTris /*host*/, float* pendingUv /*host, 6 each*/,
                                void* tmp, hipStream_t stream)
{
    if (numPending == 0) return hipSuccess;
    SetupScratch s = carve_setup(scratch, scratchBytes, numTris);
    float* dUv = (float*)tmp;
    hipLaunchKernelGGL(setup_gather_pending, dim3((numPending + 255u) / 256u), dim3(256), 0, stream, s.triUv, s.pending, numPending, dUv);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(pendingTris, s.pending, (size_t)numPending * 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pendingUv, dUv, (size_t)numPending * 24, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    return e;
}

hipError_t run_setup_items(const SetupParams& S, void* scratch, size_t scratchBytes, SetupCounters* counters, float* itemUv, uint8_t* itemLevel,
                           uint8_t* itemDegenerate, int32_t* triToItem, uint32_t* itemIds, hipStream_t stream)
{
    const uint32_t n = S.numTris;
    if (n == 0) return hipSuccess;
    SetupScratch s = carve_setup(scratch, scratchBytes, n);
    const dim3 grid((n + 255u) / 256u), block(256);
    const uint32_t slots = hash_table_slots(n);
    const HashTable table = hash_table_at(s.hash, slots);
    if (!S.disableDedup) {
        hipLaunchKernelGGL(setup_dedup_insert, /* (the table was filled by setup_fetch) */ grid, block, 0, stream, s.keysA, s.triFlags, n, table);
    }
    hipLaunchKernelGGL(setup_dedup_lookup, dim3((n + 1023u) / 1024u), dim3(1024), 0, stream, s.keysA, s.triFlags, n, table, S.disableDedup ? 1 : 0, s.triUv, s.triLevel, s.firstTri, s.isItem, s.itemOfTri,
                       (uint32_t*)s.tmp, counters);
    hipLaunchKernelGGL(setup_emit_items, grid, block, 0, stream, S, s.triUv, s.triLevel, s.triFlags, s.firstTri, s.isItem, s.itemOfTri, itemUv, itemLevel,
                       itemDegenerate, triToItem, counters);
    const uint32_t numChunks = (n + kSplitChunk - 1u) / kSplitChunk;   // (n bounds the item count, which only the device knows here)
    hipLaunchKernelGGL(setup_split_count, dim3(numChunks), block, 0, stream, itemLevel, counters, s.lkeysA);
    hipLaunchKernelGGL(setup_split_scan, dim3(1), dim3(1024), 0, stream, counters, s.lkeysA, numChunks);
    hipLaunchKernelGGL(setup_split_scatter, dim3(numChunks), block, 0, stream, itemLevel, counters, s.lkeysA, itemIds);
    return hipGetLastError();
}

} // namespace ommx
