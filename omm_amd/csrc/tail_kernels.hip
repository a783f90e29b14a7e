// tail_kernels.hip -- device restatement of the reference's serial tail
// (bake_cpu_impl.cpp:1432-1472 promote, :1031-1066 exact dedup, :1690-1705 histograms,
//  :1707-1754 spatial sort, :1756-1920 serialize) with parallel primitives.
//
// The reference's sequential "first occurrence wins" semantics are kept by construction:
//   * exact dedup     = stable radix sort of (digest, item) + segment heads  -> rep[item] = lowest item with that digest
//   * spatial order   = stable radix sort ascending of (key, item), read back to front
//                       == std::sort(std::greater<pair<key,item>>)
// rocPRIM radix sort, scan and reduce are the only library calls.
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include "hash_build.h"
#include "scan_lookback.h"
#include <stdint.h>
#include "bake_types.h"
#include "bake_kernels.h"

namespace ommx {

#define TAIL_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
size_t tail_scratch_bytes(uint32_t numItems, uint32_t numTris);

__device__ __forceinline__ int cvt_trunc_x86_t(float f) { return (f >= -2147483648.f && f < 2147483648.f) ? (int)f : (int)0x80000000; }
__device__ __forceinline__ int clampi_t(int v, int lo, int hi) { return v < lo ? lo : (hi < v ? hi : v); }
__device__ __forceinline__ uint32_t spread16(uint32_t x)
{
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// ---- round 5: the tail as six launches --------------------------------------------------------------------------------------------------------
// A small bake (configs[1]: 100 000 work items, 6 299 OMMs) spends its time in launch overhead: ~5 us per launch whatever the kernel does, and the
// tail was 30 launches (4 fills, iota, reduce, a 12-launch merge sort of ALL items, a scan, ...).  Now:
//   tail_summarize   promote + the fills every later step needs (hash table and key list to all ones, histograms / counters / tile states to zero)
//   dedup_insert     hash build; the LAST workgroup to finish enters the uniform bins (was dedup_insert_bins)
//   tail_emit        look-up + spatial key + compaction of the EMITTED items only into 64-bit keys (key30 << idxBits | item): unique keys, so the
//                    order of the compaction (atomics) does not matter and the sort needs no payload
//   tail_rank_place  (<= kRankMax keys) descriptor slot AND array offset of an OMM by counting: slot = number of greater keys, offset = sum of
//                    their sizes (the level is in the key); 16 lanes share one key's pass over the list -- no sort, no scan
//   or rocPRIM sort of the keys + tail_place (single-pass scan with decoupled look-back over dynamic tile tickets)
//   tail_indices
// The summary the host reads back (OMM count, error flag, arrayData size, OMMs of less than 16 bytes) is written by the last workgroup of the
// placing kernel.

struct TailWork {            // scratch words of one tail run (zeroed by tail_summarize); the host reads the block back in ONE copy
    uint32_t numEmitted, err, ticket, pad_;
    unsigned long long arrayBytes, smallOmms;   // arrayData size; emitted OMMs of less than 16 bytes
};
constexpr uint32_t kRankMax = 16384u;   // keys up to which counting beats sort + scan (quadratic in the number of emitted OMMs)
constexpr uint32_t kRankChunk = 4096u;  // keys staged in LDS at a time
constexpr uint32_t kPlaceTile = 1024u;

__device__ __forceinline__ uint32_t omm_bytes(uint32_t lvl, uint32_t bits) { const uint32_t n = ((1u << (2u * lvl)) * bits) >> 3; return n < 1u ? 1u : n; }

// promote (bake_cpu_impl.cpp:1432-1472) + digest of uniform items from the table + the fills of the whole tail
__global__ __launch_bounds__(256) void tail_summarize(TailInputs in, int32_t* __restrict__ special, uint32_t* __restrict__ ones, uint32_t onesWords,
                                                      uint32_t* __restrict__ zeroA, uint32_t zeroAWords, uint32_t* __restrict__ zeroB, uint32_t zeroBWords)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t w = i; w < onesWords; w += stride) ones[w] = 0xFFFFFFFFu;
    for (uint32_t w = i; w < zeroAWords; w += stride) zeroA[w] = 0u;
    for (uint32_t w = i; w < zeroBWords; w += stride) zeroB[w] = 0u;
    if (i >= in.numItems) return;
    const uint32_t mask = in.stateMask[i];
    const uint32_t level = in.level[i];
    const bool classified = mask != 0u && mask <= 15u;
    // (an unclassified item -- the bake then fails -- is flagged by tail_emit: the error word is among the words this kernel zeroes)
    bool allEqual = classified && (mask & (mask - 1u)) == 0u;
    int common = 31 - __clz((int)mask);
    if (allEqual && in.uniformDigest) {
        const int s3 = common == 2 ? 3 : common;
        in.digests[i] = in.uniformDigest[level * 4u + (uint32_t)s3];
    }
    if (!allEqual && in.rejectionThreshold > 0.f) {
        const float frac = (float)in.knownCount[i] / (float)(1u << (2u * level));
        if (frac < in.rejectionThreshold) { allEqual = true; common = 2; }
    }
    special[i] = (allEqual && !in.disableSpecial) ? (-(int32_t)common - 1) : 0;
}

// ---- exact-duplicate detection as a hash build (DeduplicateExact, bake_cpu_impl.cpp:1031-1066: equality of the XXH64 digest only, the
//      FIRST work item with a digest keeps its block): rep[i] = smallest item index with digest[i] (hash_build.h) ----
// Uniform work items (87 % of the bench workload) carry one of 13 x 4 table digests (tail_summarize): their first occurrence per (level, state)
// is a min-reduction -- LDS bins per workgroup, then at most one gated global atomic per bin and workgroup -- not 870 000 operations on three
// table slots.  The bins enter the table afterwards (the last workgroup to finish does it), so a non-uniform item whose digest happens to equal a
// table digest still merges with it, as digest-only equality demands.
constexpr uint32_t kUniformBins = kNumLevels * 4u;
__global__ __launch_bounds__(256) void dedup_insert(const uint64_t* __restrict__ digests, const uint32_t* __restrict__ stateMask, const uint8_t* __restrict__ level,
                                                    int haveUniformTable, uint32_t n, HashTable table, uint32_t* __restrict__ firstUniform)
{
    __shared__ uint32_t bins[kUniformBins];
    __shared__ uint32_t buckets[256];
    if (threadIdx.x < kUniformBins) bins[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < n;
    if (live && haveUniformTable) {
        const uint32_t m = stateMask[i];
        if (m != 0u && m <= 15u && (m & (m - 1u)) == 0u) {   // same test and same bin as tail_summarize
            const int common = 31 - __clz((int)m);
            atomicMin(&bins[(uint32_t)level[i] * 4u + (uint32_t)(common == 2 ? 3 : common)], i);
            live = false;
        }
    }
    hash_put_min_block(table, live, live ? digests[i] : 0ull, i, buckets);
    __syncthreads();
    if (threadIdx.x < kUniformBins) {
        const uint32_t b = bins[threadIdx.x];
        if (b != 0xFFFFFFFFu && __atomic_load_n(firstUniform + threadIdx.x, __ATOMIC_RELAXED) > b) atomicMin(firstUniform + threadIdx.x, b);
    }
}

// (a launch of its own: letting the last workgroup of dedup_insert do it needs a device-wide fence per workgroup -- an L2 write-back on this chip --,
//  measured 9 -> 35 us for the insert at configs[1])
__global__ __launch_bounds__(64) void dedup_insert_bins(const uint64_t* __restrict__ uniformDigest, const uint32_t* __restrict__ firstUniform, HashTable table)
{
    const uint32_t t = threadIdx.x;
    if (t < kUniformBins && firstUniform[t] != 0xFFFFFFFFu) hash_put_min(table, uniformDigest[t], firstUniform[t]);
}

// 30-bit spatial key of a work item: level << 26 | 26 Morton bits of its centroid on the reference's 8192^2 grid (bake_cpu_impl.cpp:1722-1748; the
// reference's 64-bit key is level << 60 | the same Morton bits: the same order)
__device__ __forceinline__ uint32_t spatial_key30(const float* __restrict__ p, uint32_t level)
{
    const float cx = (p[0] + p[2] + p[4]) / 3.f, cy = (p[1] + p[3] + p[5]) / 3.f;
    const int qx = cvt_trunc_x86_t(8192.f * cx), qy = cvt_trunc_x86_t(8192.f * cy);
    // GetTexCoord<MirrorOnce, non-pow2> on an 8192^2 grid (util/texture.h:84-87)
    const int mx = clampi_t(cvt_trunc_x86_t(__builtin_fabsf((float)qx + 0.5f)), 0, 8191);
    const int my = clampi_t(cvt_trunc_x86_t(__builtin_fabsf((float)qy + 0.5f)), 0, 8191);
    return (level << 26) | (spread16((uint32_t)mx) | (spread16((uint32_t)my) << 1));   // (mx, my < 8192: 26 Morton bits)
}

// look-up of the first occurrence, special indices into itemValue, and the emitted items (first of their digest, not special) as sort keys:
// spatial key (bake_cpu_impl.cpp:1722-1748) << idxBits | item -- the reference sorts pairs (key, item) descending: the same order, read back to front
__global__ __launch_bounds__(256) void tail_emit(TailInputs in, const int32_t* __restrict__ special, HashTable table, uint32_t idxBits, uint32_t bound,
                                                 uint32_t* __restrict__ rep, int32_t* __restrict__ itemValue, unsigned long long* __restrict__ keys,
                                                 TailWork* __restrict__ work)
{
    __shared__ uint32_t s_waveBase[4], s_blockBase;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool emitted = false; unsigned long long key = 0ull;
    if (i < in.numItems) {
        const uint32_t mask = in.stateMask[i];
        if (!(mask != 0u && mask <= 15u)) { atomicOr(in.errorFlag, 1u); atomicOr(&work->err, 1u); }   // (tail_summarize cannot flag it: it zeroes the words)
        const uint32_t r = in.disableDedup ? i : hash_get(table, in.digests[i], i);
        rep[i] = r;
        const int32_t sp = special[i];
        if (sp != 0) itemValue[i] = sp;
        emitted = r == i && sp == 0;
        if (emitted) key = ((unsigned long long)spatial_key30(in.uv + 6ull * i, (uint32_t)in.level[i]) << idxBits) | i;
    }
    const unsigned long long b = __ballot(emitted);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_waveBase[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < 4; ++w) { const uint32_t c = s_waveBase[w]; s_waveBase[w] = run; run += c; }
        s_blockBase = run ? atomicAdd(&work->numEmitted, run) : 0u;
    }
    __syncthreads();
    if (emitted) {
        const uint32_t pos = s_blockBase + s_waveBase[wave] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (pos < bound) keys[pos] = key; else atomicOr(&work->err, 1u);   // (the bound is exact arithmetic on the active count: not reachable)
    }
}

// Descriptor slot and array offset by counting (<= kRankMax keys): an OMM's slot in the descending order is the number of greater keys; the level leads
// the key, so its offset is the bytes of all OMMs of higher levels (level histogram) plus the greater keys of its own level times that level's size.
// The list goes through LDS a chunk at a time (every workgroup stages all of it: 50 KB at configs[1]); 32 lanes share one key's pass over a chunk.
__global__ __launch_bounds__(1024) void tail_rank_place(const unsigned long long* __restrict__ keys, uint32_t idxBits, uint32_t bound, int bits, TailWork* __restrict__ work,
                                                        uint32_t* __restrict__ order, uint32_t* __restrict__ dstOfs, uint32_t* __restrict__ sizes,
                                                        int32_t* __restrict__ itemValue, uint32_t* __restrict__ arrayHist)
{
    __shared__ unsigned long long s_keys[kRankChunk];
    __shared__ uint32_t h[kNumLevels];
    if (threadIdx.x < kNumLevels) h[threadIdx.x] = 0;
    uint32_t E = __atomic_load_n(&work->numEmitted, __ATOMIC_RELAXED); if (E > bound) E = bound;
    const uint32_t e = blockIdx.x * 32u + (threadIdx.x >> 5), sub = threadIdx.x & 31u, lvlShift = idxBits + 26u;
    const unsigned long long mine = e < E ? keys[e] : 0ull;
    uint32_t cnt = 0;
    for (uint32_t c0 = 0; c0 < E; c0 += kRankChunk) {
        __syncthreads();
        const uint32_t m = E - c0 < kRankChunk ? E - c0 : kRankChunk;
        for (uint32_t k = threadIdx.x; k < kRankChunk; k += 1024u) {
            const bool live = k < m;
            const unsigned long long kk = live ? keys[c0 + k] : 0ull;
            if (live) s_keys[k] = kk;
            const uint32_t lvl = live ? (uint32_t)(kk >> lvlShift) : 0xFFu;
            unsigned long long todo = __ballot(live);   // one LDS atomic per (wave, level)
            while (todo) {
                const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
                const uint32_t l0 = (uint32_t)__shfl((int)lvl, (int)leader);
                const unsigned long long same = __ballot(lvl == l0) & todo;
                if ((threadIdx.x & 63u) == leader && l0 < (uint32_t)kNumLevels) atomicAdd(&h[l0], (uint32_t)__popcll(same));
                todo &= ~same;
            }
        }
        __syncthreads();
        if (e < E) {
            #pragma unroll 8
            for (uint32_t k = sub; k < m; k += 32u) cnt += s_keys[k] > mine ? 1u : 0u;
        }
    }
    __syncthreads();
    for (int d = 16; d >= 1; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d);
    if (blockIdx.x == 0) {   // (every workgroup has the whole histogram; one publishes it)
        if (threadIdx.x < kNumLevels && h[threadIdx.x]) arrayHist[threadIdx.x] = h[threadIdx.x];
        if (threadIdx.x == 0) { uint32_t small = 0; for (int l = 0; l < (bits == 2 ? 3 : 4); ++l) small += h[l]; work->smallOmms = small; }
    }
    if (e < E && sub == 0) {
        const uint32_t item = (uint32_t)(mine & ((1ull << idxBits) - 1ull)), lvl = (uint32_t)(mine >> lvlShift), sz = omm_bytes(lvl, (uint32_t)bits);
        uint32_t above = 0; unsigned long long ofs = 0;
        for (uint32_t l = lvl + 1u; l < (uint32_t)kNumLevels; ++l) { above += h[l]; ofs += (unsigned long long)h[l] * omm_bytes(l, (uint32_t)bits); }
        ofs += (unsigned long long)(cnt - above) * sz;
        order[cnt] = item; dstOfs[cnt] = (uint32_t)ofs; sizes[cnt] = sz; itemValue[item] = (int32_t)cnt;
        if (cnt == E - 1u) work->arrayBytes = ofs + sz;
    }
}

// The same from the SORTED keys (ascending; descriptor j = key E - 1 - j): sizes, their exclusive scan in one pass -- a tile's total is published
// (flag 1), its exclusive prefix collected by looking back over the tiles before it until one with an inclusive prefix (flag 2) is met; tiles are
// handed out by a ticket, so a tile's predecessors are always running or done.  State word: flag << 62 | bytes.
__global__ __launch_bounds__(kPlaceTile) void tail_place(const unsigned long long* __restrict__ sorted, uint32_t idxBits, uint32_t bound, int bits, TailWork* __restrict__ work,
                                                         unsigned long long* __restrict__ tileState, uint32_t* __restrict__ order, uint32_t* __restrict__ dstOfs,
                                                         uint32_t* __restrict__ sizes, int32_t* __restrict__ itemValue, uint32_t* __restrict__ arrayHist)
{
    __shared__ uint32_t h[kNumLevels];
    __shared__ unsigned long long s_wave[kPlaceTile / 64u], s_excl;
    __shared__ uint32_t s_tile;
    if (threadIdx.x < kNumLevels) h[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_tile = atomicAdd(&work->ticket, 1u);
    __syncthreads();
    uint32_t E = __atomic_load_n(&work->numEmitted, __ATOMIC_RELAXED); if (E > bound) E = bound;
    const uint32_t tile = s_tile, j = tile * kPlaceTile + threadIdx.x;
    if (tile * kPlaceTile < E) {   // (uniform per workgroup)
        unsigned long long key = 0; uint32_t sz = 0, lvl = 0, item = 0;
        if (j < E) {
            key = sorted[E - 1u - j];
            item = (uint32_t)(key & ((1ull << idxBits) - 1ull)); lvl = (uint32_t)(key >> (idxBits + 26u)); sz = omm_bytes(lvl, (uint32_t)bits);
            atomicAdd(&h[lvl], 1u);
        }
        // inclusive scan of sz within the wave, wave totals through LDS
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        unsigned long long inc = sz;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d), hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d);
            if (lane >= (uint32_t)d) inc += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 63u) s_wave[wave] = inc;
        __syncthreads();
        unsigned long long waveBase = 0, total = 0;
        for (uint32_t w = 0; w < kPlaceTile / 64u; ++w) { const unsigned long long c = s_wave[w]; if (w < wave) waveBase += c; total += c; }
        if (wave == 0) { const unsigned long long excl = lookback_exclusive(tileState, tile, total); if (lane == 0) s_excl = excl; }
        __syncthreads();
        if (j < E) {
            const unsigned long long ofs = s_excl + waveBase + inc - sz;
            order[j] = item; dstOfs[j] = (uint32_t)ofs; sizes[j] = sz; itemValue[item] = (int32_t)j;
            if (j == E - 1u) work->arrayBytes = ofs + sz;
        }
    }
    __syncthreads();
    if (threadIdx.x < kNumLevels && h[threadIdx.x]) {
        atomicAdd(&arrayHist[threadIdx.x], h[threadIdx.x]);
        if (threadIdx.x < (bits == 2 ? 3u : 4u)) atomicAdd(&work->smallOmms, (unsigned long long)h[threadIdx.x]);
    }
}

// index buffer + index histogram (bake_cpu_impl.cpp:1697-1702,1856-1870)
__global__ __launch_bounds__(256) void tail_indices(TailInputs in, const uint32_t* __restrict__ rep, const int32_t* __restrict__ itemValue,
                                                    int32_t* __restrict__ out, uint32_t* __restrict__ indexHist, void* __restrict__ narrow, int narrowBytes)
{
    __shared__ uint32_t h[kNumLevels];
    if (threadIdx.x < kNumLevels) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < in.numTris) {
        const int32_t it = in.triToItem[t];
        int32_t v = in.unresolved;
        if (it >= 0) {
            const uint32_t r = rep[it];
            v = itemValue[r];
            if (v >= 0) atomicAdd(&h[in.level[r]], 1u);
        }
        out[t] = v;
        // index narrowing (bake_cpu_impl.cpp:1872-1902; was a launch of its own)
        if (narrow) { if (narrowBytes == 1) ((int8_t*)narrow)[t] = (int8_t)v; else if (narrowBytes == 2) ((int16_t*)narrow)[t] = (int16_t)v; else ((int32_t*)narrow)[t] = v; }
    }
    __syncthreads();
    if (threadIdx.x < kNumLevels && h[threadIdx.x]) atomicAdd(&indexHist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void tail_descs(const uint32_t* __restrict__ order, const uint32_t* __restrict__ dstOfs, const uint8_t* __restrict__ level,
                                                  int format, uint32_t numOmms, uint2* __restrict__ descs)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= numOmms) return;
    // ommCpuOpacityMicromapDesc { u32 offset; u16 subdivisionLevel; u16 format; }
    descs[j] = make_uint2(dstOfs[j], (uint32_t)level[order[j]] | ((uint32_t)format << 16));
}

void launch_write_descs(const uint32_t* order, const uint32_t* dstOfs, const uint8_t* level, int format, uint32_t numOmms, void* descArray, hipStream_t stream)
{
    if (numOmms == 0) return;
    hipLaunchKernelGGL(tail_descs, dim3((numOmms + 255u) / 256u), dim3(256), 0, stream, order, dstOfs, level, format, numOmms, (uint2*)descArray);
}

// ---- active-item compaction (after triage): per-level lists of the items that need per-micro-triangle work, and their
//      slots in the packed-state buffer ----
// One launch (round 5; it was flags + two rocPRIM scans + scatter = 6): a single-pass scan of (active flag, slot bytes) with decoupled look-back, like
// tail_place -- a tile publishes its totals (flag 1), collects its exclusive prefix from the tiles before it until one with an inclusive prefix (flag 2),
// tiles handed out by ticket.  The state words (prep_state_words(): ticket, then count and byte states per tile) sit at the start of the scratch block
// and are zeroed by triage_items, the launch in front of this one.
constexpr uint32_t kPrepTile = 1024u;   // (items per tile = threads per workgroup: a look-back per 1024 items)
static_assert(kPrepTile == 1024u, "prep_state_words() (bake_kernels.h) sizes the state block for tiles of 1024 items");
__global__ __launch_bounds__(kPrepTile) void prep_compact(const uint32_t* __restrict__ itemIds, const uint8_t* __restrict__ active, const uint8_t* __restrict__ level,
                                                          int bits, SetupCounters* __restrict__ counters, uint32_t* __restrict__ stateWords,
                                                          uint32_t* __restrict__ activeIds, uint64_t* __restrict__ stateOfs)
{
    __shared__ uint32_t s_cnt[kPrepTile / 64u], s_tile, s_exclCnt, s_start[kNumLevels + 1];
    __shared__ unsigned long long s_bytes[kPrepTile / 64u], s_exclBytes;
    if (threadIdx.x == 0) s_tile = atomicAdd(stateWords, 1u);
    if (threadIdx.x <= (uint32_t)kNumLevels) s_start[threadIdx.x] = counters->levelStart[threadIdx.x];
    __syncthreads();
    const uint32_t n = counters->numItems, tile = s_tile, p = tile * kPrepTile + threadIdx.x;
    if (n == 0u) {   // (no work items: the boundaries the host reads back are all zero)
        if (tile == 0u && threadIdx.x <= (uint32_t)kNumLevels) counters->activeStart[threadIdx.x] = 0u;
        if (tile == 0u && threadIdx.x == 0) counters->stateBytes = 0ull;
        return;
    }
    if (tile * kPrepTile >= n) return;   // (uniform per workgroup; nobody looks back at a tile past the end)
    unsigned long long* cntState = (unsigned long long*)(stateWords + 64);
    unsigned long long* bytState = cntState + (gridDim.x + 1u);
    uint32_t item = 0, f = 0; unsigned long long bytes = 0;
    if (p < n) {
        item = itemIds[p];
        f = active[item] ? 1u : 0u;
        if (f) { bytes = (((unsigned long long)1 << (2u * level[item])) * (unsigned long long)bits) >> 3; if (bytes < 16) bytes = 16; }   // 16-byte slots keep vector copies aligned
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(f != 0u);
    const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    unsigned long long inc = bytes;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d), hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d);
        if (lane >= (uint32_t)d) inc += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63u) { s_cnt[wave] = (uint32_t)__popcll(bal); s_bytes[wave] = inc; }
    __syncthreads();
    uint32_t waveCnt = 0, totalCnt = 0; unsigned long long waveBytes = 0, totalBytes = 0;
    for (uint32_t w = 0; w < kPrepTile / 64u; ++w) {
        const uint32_t c = s_cnt[w]; const unsigned long long bb = s_bytes[w];
        if (w < wave) { waveCnt += c; waveBytes += bb; }
        totalCnt += c; totalBytes += bb;
    }
    if (wave == 0) {
        const unsigned long long exC = lookback_exclusive(cntState, tile, totalCnt), exB = lookback_exclusive(bytState, tile, totalBytes);
        if (lane == 0) { s_exclCnt = (uint32_t)exC; s_exclBytes = exB; }
    }
    __syncthreads();
    if (p >= n) return;
    const uint32_t pos = s_exclCnt + waveCnt + before;
    const unsigned long long ofs = s_exclBytes + waveBytes + inc - bytes;
    if (f) { activeIds[pos] = item; stateOfs[item] = ofs; }
    // level boundaries of the compacted list + total bytes (levelStart[l] == n: the level and all above it are empty)
    for (int l = 0; l <= kNumLevels; ++l) if (s_start[l] == p) counters->activeStart[l] = pos;
    if (p == n - 1u) {
        for (int l = 0; l <= kNumLevels; ++l) if (s_start[l] >= n) counters->activeStart[l] = pos + f;
        counters->stateBytes = ofs + bytes;
    }
}

hipError_t run_prep(const uint32_t* itemIds, const uint8_t* active, const uint8_t* level, int bits, uint32_t maxItems, SetupCounters* counters,
                    uint32_t* activeIds, uint64_t* stateOfs, void* scratch, size_t scratchBytes, hipStream_t stream)
{
    const uint32_t n = maxItems;
    if (n == 0) return hipSuccess;
    if (scratchBytes < (size_t)prep_state_words(n) * 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prep_compact, dim3((n + kPrepTile - 1u) / kPrepTile), dim3(kPrepTile), 0, stream, itemIds, active, level, bits, counters, (uint32_t*)scratch, activeIds, stateOfs);
    return hipGetLastError();
}

// ---- multi-GPU load balance: interleave the active list of one level -------------------------------------------------------
// Ranks own CONTIGUOUS position ranges of the per-level active list.  In input order neighbouring triangles of a real mesh have
// similar classification cost, so a contiguous range can be much heavier than another.  Position j of the list handed to the
// ranks takes the item from natural position (j * stride) mod count, stride ~ count / golden ratio and coprime to count: a
// bijection under which every contiguous j-range samples the whole level evenly.  Same permutation on every rank (pure function
// of count), and nothing downstream depends on the order of the list.
__global__ __launch_bounds__(256) void shard_interleave(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t count, uint32_t stride)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) out[j] = in[(uint32_t)(((uint64_t)j * stride) % count)];
}
void launch_shard_interleave(const uint32_t* in, uint32_t* out, uint32_t count, uint32_t stride, hipStream_t stream)
{
    if (count == 0) return;
    hipLaunchKernelGGL(shard_interleave, dim3((count + 255u) / 256u), dim3(256), 0, stream, in, out, count, stride);
}

// ---- multi-GPU sharding: work items are partitioned over ranks, the tail is replicated ----
// rank ranges of the per-level active lists: rank r owns positions [bounds[l][r], bounds[l][r+1]) of the compacted list
__device__ __forceinline__ uint32_t owner_of_position(const ShardBounds& B, uint32_t p)
{
    uint32_t l = 0;
    while (l + 1 < (uint32_t)kNumLevels && p >= B.b[l + 1][0]) ++l;
    uint32_t r = 0;
    while (r + 1 < B.world && p >= B.b[l][r + 1]) ++r;
    return r;
}

// compact metadata of the active items for the cross-rank SUM all-reduce: [mask | known | digest lo | digest hi] x numActive,
// non-zero only in the positions this rank classified
__global__ __launch_bounds__(256) void shard_pack_meta(ShardBounds B, const uint32_t* __restrict__ activeIds, uint32_t numActive,
                                                       const uint32_t* __restrict__ mask, const uint32_t* __restrict__ known,
                                                       const uint64_t* __restrict__ digests, uint32_t* __restrict__ meta)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= numActive) return;
    const bool mine = owner_of_position(B, p) == B.rank;
    const uint32_t item = activeIds[p];
    const uint64_t dg = mine ? digests[item] : 0ull;
    meta[p] = mine ? mask[item] : 0u;
    meta[numActive + p] = mine ? known[item] : 0u;
    meta[2u * numActive + p] = (uint32_t)dg;
    meta[3u * numActive + p] = (uint32_t)(dg >> 32);
}

__global__ __launch_bounds__(256) void shard_unpack_meta(ShardBounds B, const uint32_t* __restrict__ activeIds, uint32_t numActive,
                                                         const uint32_t* __restrict__ meta, uint32_t* __restrict__ mask, uint32_t* __restrict__ known,
                                                         uint64_t* __restrict__ digests, uint8_t* __restrict__ owner)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= numActive) return;
    const uint32_t item = activeIds[p];
    mask[item] = meta[p]; known[item] = meta[numActive + p];
    digests[item] = (uint64_t)meta[2u * numActive + p] | ((uint64_t)meta[3u * numActive + p] << 32);
    owner[item] = (uint8_t)owner_of_position(B, p);
}

void launch_shard_pack_meta(const ShardBounds& B, const uint32_t* activeIds, uint32_t numActive, const uint32_t* mask, const uint32_t* known,
                            const uint64_t* digests, uint32_t* meta, hipStream_t stream)
{
    if (numActive) hipLaunchKernelGGL(shard_pack_meta, dim3((numActive + 255u) / 256u), dim3(256), 0, stream, B, activeIds, numActive, mask, known, digests, meta);
}
void launch_shard_unpack_meta(const ShardBounds& B, const uint32_t* activeIds, uint32_t numActive, const uint32_t* meta, uint32_t* mask, uint32_t* known,
                              uint64_t* digests, uint8_t* owner, hipStream_t stream)
{
    if (numActive) hipLaunchKernelGGL(shard_unpack_meta, dim3((numActive + 255u) / 256u), dim3(256), 0, stream, B, activeIds, numActive, meta, mask, known, digests, owner);
}

// per-rank layout of the surviving blocks: block j (final order) of owner r sits at cofs[j] inside rank r's contribution
__global__ __launch_bounds__(256) void shard_masked_sizes(const uint32_t* __restrict__ order, const uint32_t* __restrict__ sizes, const uint8_t* __restrict__ active,
                                                          const uint8_t* __restrict__ owner, uint32_t numOmms, uint32_t r, uint64_t* __restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= numOmms) return;
    const uint32_t item = order[j];
    out[j] = (active[item] && owner[item] == r) ? (uint64_t)sizes[j] : 0ull;
}
__global__ __launch_bounds__(256) void shard_take_offsets(const uint32_t* __restrict__ order, const uint8_t* __restrict__ active, const uint8_t* __restrict__ owner,
                                                          const uint64_t* __restrict__ masked, const uint64_t* __restrict__ scan, uint32_t numOmms, uint32_t r,
                                                          uint64_t* __restrict__ cofs, uint64_t* __restrict__ totals)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= numOmms) return;
    const uint32_t item = order[j];
    if (active[item] && owner[item] == r) cofs[j] = scan[j];
    if (j == numOmms - 1) totals[r] = scan[j] + masked[j];
}

hipError_t run_shard_layout(const uint32_t* order, const uint32_t* sizes, const uint8_t* active, const uint8_t* owner, uint32_t numOmms, uint32_t world,
                            uint64_t* cofs, uint64_t* totalsDev, uint64_t* totalsHost, void* scratch, size_t scratchBytes, hipStream_t stream)
{
    for (uint32_t r = 0; r < world; ++r) totalsHost[r] = 0;
    if (numOmms == 0) return hipSuccess;
    if (scratchBytes < tail_scratch_bytes(numOmms, 0)) return hipErrorInvalidValue;
    uint8_t* p = (uint8_t*)scratch;
    const size_t n64 = (((size_t)numOmms + 1) * 8 + 255) / 256 * 256;
    uint64_t* masked = (uint64_t*)p; p += n64; uint64_t* scan = (uint64_t*)p; p += n64;
    void* tmp = p; const size_t tmpBytes = scratchBytes - (size_t)(p - (uint8_t*)scratch);
    const dim3 grid((numOmms + 255u) / 256u), block(256);
    for (uint32_t r = 0; r < world; ++r) {
        hipLaunchKernelGGL(shard_masked_sizes, grid, block, 0, stream, order, sizes, active, owner, numOmms, r, masked);
        size_t tb = tmpBytes;
        TAIL_CHECK(rocprim::exclusive_scan(tmp, tb, masked, scan, (uint64_t)0, (size_t)numOmms, rocprim::plus<uint64_t>(), stream));
        hipLaunchKernelGGL(shard_take_offsets, grid, block, 0, stream, order, active, owner, masked, scan, numOmms, r, cofs, totalsDev);
    }
    TAIL_CHECK(hipMemcpyAsync(totalsHost, totalsDev, sizeof(uint64_t) * world, hipMemcpyDeviceToHost, stream));
    return hipStreamSynchronize(stream);
}

// rank-local gather: this rank's surviving blocks, densely, in final order
__global__ __launch_bounds__(256) void shard_gather_contribution(const uint8_t* __restrict__ states, const uint64_t* __restrict__ stateOfs,
                                                                 const uint8_t* __restrict__ active, const uint8_t* __restrict__ owner, uint32_t rank,
                                                                 const uint32_t* __restrict__ order, const uint64_t* __restrict__ cofs,
                                                                 const uint32_t* __restrict__ sizes, uint32_t numOmms, uint8_t* __restrict__ contrib)
{
    for (uint32_t j = blockIdx.x; j < numOmms; j += gridDim.x) {
        const uint32_t item = order[j];
        if (!active[item] || owner[item] != rank) continue;
        const uint8_t* src = states + stateOfs[item];
        uint8_t* dst = contrib + cofs[j];
        const uint32_t n = sizes[j];
        if (n >= 16u) { const uint4* s4 = (const uint4*)src; uint4* d4 = (uint4*)dst; for (uint32_t k = threadIdx.x; k < n / 16u; k += blockDim.x) d4[k] = s4[k]; }
        else if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
    }
}

// after the all-gather: place every rank's blocks at their final arrayData offsets (uniform items: constant pattern, computed locally).
// `gathered` holds the bytes [lo, hi) of every rank's contribution, rank r at gathered + r * rankPitch: the all-gather may arrive in
// chunks (omm_host.cpp), and each chunk is scattered while the next one is still on the wire; a block that straddles two chunks is
// written in two pieces.
__global__ __launch_bounds__(256) void shard_scatter_contributions(const uint8_t* __restrict__ gathered, uint64_t rankPitch, uint64_t lo, uint64_t hi,
                                                                   const uint8_t* __restrict__ active, const uint8_t* __restrict__ owner,
                                                                   const uint32_t* __restrict__ stateMask, const uint8_t* __restrict__ level, int bits,
                                                                   const uint32_t* __restrict__ order, const uint64_t* __restrict__ cofs,
                                                                   const uint32_t* __restrict__ dstOfs, const uint32_t* __restrict__ sizes, uint32_t numOmms,
                                                                   uint8_t* __restrict__ arrayData)
{
    for (uint32_t j = blockIdx.x; j < numOmms; j += gridDim.x) {
        const uint32_t item = order[j];
        uint8_t* dst = arrayData + dstOfs[j];
        const uint32_t n = sizes[j];
        if (!active[item]) {
            if (lo != 0) continue;   // (written with the first chunk)
            const uint32_t st = (uint32_t)(31 - __clz((int)stateMask[item]));
            uint32_t usedBits = (1u << (2u * level[item])) * (uint32_t)bits; if (usedBits > 8u) usedBits = 8u;
            uint32_t pat = 0;
            for (uint32_t b = 0; b < usedBits; b += (uint32_t)bits) pat |= st << b;
            for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) dst[k] = (uint8_t)pat;
            continue;
        }
        const uint64_t c0 = cofs[j], c1 = c0 + n;
        const uint64_t a = c0 > lo ? c0 : lo, b = c1 < hi ? c1 : hi;   // the part of this block that lives in [lo, hi)
        if (a >= b) continue;
        const uint8_t* src = gathered + (uint64_t)owner[item] * rankPitch + (a - lo);
        uint8_t* d = dst + (a - c0);
        const uint64_t len = b - a;
        if ((((uint64_t)src | (uint64_t)d | len) & 15ull) == 0) { const uint4* s4 = (const uint4*)src; uint4* d4 = (uint4*)d; for (uint64_t k = threadIdx.x; k < len / 16u; k += blockDim.x) d4[k] = s4[k]; }
        else for (uint64_t k = threadIdx.x; k < len; k += blockDim.x) d[k] = src[k];
    }
}

// ---- block exchange codec (multi-GPU: what crosses xGMI) ----
// The blocks of a bake are long runs of one state with a thin band of mixed micro-triangles along the alpha edge: of the 16-byte units of the metric
// workload's arrayData (64 micro-triangles in 4-state) 96.7 % hold one repeated state (configs[4]: 98.8 %, asset-sized cards: 58.6 %;
// profiles/scripts/r03_block_compressibility.py).  A rank's contribution therefore travels as
//     [0] bytes of the stream, [8] units | offsets: first raw unit of every 256-unit block (uint32, blocks + 1) | codes: one nibble per unit, 0..3 = the
//     unit is 16 bytes of 0x00 / 0x55 / 0xAA / 0xFF, 4 = raw | the raw units, 16 bytes each
// -- 6.4 % of the bytes at the metric configuration -- and is decoded on arrival, straight into the result (shard_scatter_streams).  Lossless for any data; a contribution that does not shrink
// below half its size is sent as it is (size word = kCodecIncompressible).
constexpr uint32_t kCodecBlock = 256;
struct CodecLayout { uint64_t units, blocks, offOfs, offCodes, offRaw; };
__host__ __device__ inline CodecLayout codec_layout(uint64_t contributionBytes)
{
    CodecLayout c; c.units = contributionBytes / 16u; c.blocks = (c.units + kCodecBlock - 1u) / kCodecBlock; c.offOfs = 16u;
    c.offCodes = (c.offOfs + 4u * (c.blocks + 1u) + 15u) & ~15ull; c.offRaw = (c.offCodes + (c.units + 1u) / 2u + 15u) & ~15ull;
    return c;
}
__device__ __forceinline__ uint32_t codec_code(const uint4& v)
{
    return codec_unit_code(v.x, v.y, v.z, v.w);
}
// raw units in front of this thread inside its 256-unit block, and the block's total (all threads call; s: 4 words of LDS)
__device__ __forceinline__ uint32_t codec_rank(bool raw, uint32_t* s, uint32_t& total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long v = __ballot(raw);
    if (lane == 0) s[wave] = (uint32_t)__popcll(v);
    __syncthreads();
    uint32_t before = 0; total = 0;
    for (uint32_t w = 0; w < 4u; ++w) { before += w < wave ? s[w] : 0u; total += s[w]; }
    return before + (uint32_t)__popcll(v & ((1ull << lane) - 1ull));
}
__global__ __launch_bounds__(256) void shard_codec_count(const uint4* __restrict__ contrib, CodecLayout c, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t s[4];
    const uint64_t u = (uint64_t)blockIdx.x * kCodecBlock + threadIdx.x;
    uint32_t total;
    (void)codec_rank(u < c.units && codec_code(contrib[u]) == 4u, s, total);
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void shard_codec_write(const uint4* __restrict__ contrib, CodecLayout c, uint8_t* __restrict__ comp, uint64_t capBytes)
{
    __shared__ uint32_t s[4];
    const uint32_t* ofs = (const uint32_t*)(comp + c.offOfs);
    if (c.offRaw + 16ull * ofs[c.blocks] > capBytes) return;   // does not shrink enough: nothing is written (shard_codec_finish says so)
    const uint64_t u = (uint64_t)blockIdx.x * kCodecBlock + threadIdx.x;
    const bool live = u < c.units;
    const uint4 v = live ? contrib[u] : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t code = live ? codec_code(v) : 0u;
    uint32_t total;
    const uint32_t rank = codec_rank(live && code == 4u, s, total);
    const uint32_t other = (uint32_t)__shfl_xor((int)code, 1);
    if (live && (threadIdx.x & 1u) == 0u) comp[c.offCodes + u / 2u] = (uint8_t)(code | (other << 4));   // (units is even: contributions are multiples of 256 bytes)
    if (live && code == 4u) ((uint4*)(comp + c.offRaw))[ofs[blockIdx.x] + rank] = v;
}
// the same stream from the gather's unit codes: only the raw units are read from the array
__global__ __launch_bounds__(256) void shard_codec_write_coded(const uint4* __restrict__ contrib, const uint8_t* __restrict__ unitCodes, CodecLayout c, uint8_t* __restrict__ comp, uint64_t capBytes)
{
    __shared__ uint32_t s[4];
    const uint32_t* ofs = (const uint32_t*)(comp + c.offOfs);
    if (c.offRaw + 16ull * ofs[c.blocks] > capBytes) return;
    const uint64_t u = (uint64_t)blockIdx.x * kCodecBlock + threadIdx.x;
    const bool live = u < c.units;
    const uint32_t code = live ? (uint32_t)unitCodes[u] : 0u;
    uint32_t total;
    const uint32_t rank = codec_rank(live && code == 4u, s, total);
    const uint32_t other = (uint32_t)__shfl_xor((int)code, 1);
    if (live && (threadIdx.x & 1u) == 0u) comp[c.offCodes + u / 2u] = (uint8_t)(code | (other << 4));
    if (live && code == 4u) ((uint4*)(comp + c.offRaw))[ofs[blockIdx.x] + rank] = contrib[u];
}
__global__ void shard_codec_finish(CodecLayout c, uint8_t* __restrict__ comp, uint64_t capBytes, uint32_t* __restrict__ sizeWord)
{
    const uint64_t bytes = c.offRaw + 16ull * ((const uint32_t*)(comp + c.offOfs))[c.blocks];
    ((uint64_t*)comp)[0] = bytes; ((uint64_t*)comp)[1] = c.units;
    *sizeWord = bytes <= capBytes ? (uint32_t)(bytes / 16u) : kCodecIncompressible;
}
// After the all-gather of the codec streams: every block goes from its owner's stream STRAIGHT to its final arrayData offset -- no expanded copy of the
// contributions in between (expansion + scatter moved 3 x the result through HBM; this writes it once).  One workgroup per OMM block, like
// shard_scatter_contributions; it walks the 256-unit codec blocks its bytes [cofs, cofs + size) overlap, thread t decoding unit t of each (the rank
// of a raw unit inside its codec block comes from the same ballots as in the expansion), and writes the part of the unit that belongs to the block:
// whole aligned units as one 16-byte store, the ragged ends of small or unaligned blocks byte by byte.  `streams` holds rank r's stream at
// streams + r * streamPitch; every stream describes a contribution of the same (padded) size, hence one layout `c`.
__global__ __launch_bounds__(256) void shard_scatter_streams(const uint8_t* __restrict__ streams, uint64_t streamPitch, CodecLayout c,
                                                             const uint8_t* __restrict__ active, const uint8_t* __restrict__ owner,
                                                             const uint32_t* __restrict__ stateMask, const uint8_t* __restrict__ level, int bits,
                                                             const uint32_t* __restrict__ order, const uint64_t* __restrict__ cofs,
                                                             const uint32_t* __restrict__ dstOfs, const uint32_t* __restrict__ sizes, uint32_t numOmms,
                                                             uint8_t* __restrict__ arrayData)
{
    __shared__ uint32_t s[4];
    for (uint32_t j = blockIdx.x; j < numOmms; j += gridDim.x) {
        const uint32_t item = order[j];
        uint8_t* dst = arrayData + dstOfs[j];
        const uint32_t n = sizes[j];
        if (!active[item]) {   // uniform item: constant pattern, computed locally
            const uint32_t st = (uint32_t)(31 - __clz((int)stateMask[item]));
            uint32_t usedBits = (1u << (2u * level[item])) * (uint32_t)bits; if (usedBits > 8u) usedBits = 8u;
            uint32_t pat = 0;
            for (uint32_t b = 0; b < usedBits; b += (uint32_t)bits) pat |= st << b;
            for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) dst[k] = (uint8_t)pat;
            continue;
        }
        const uint8_t* comp = streams + (uint64_t)owner[item] * streamPitch;
        const uint32_t* ofs = (const uint32_t*)(comp + c.offOfs);
        const uint4* raw = (const uint4*)(comp + c.offRaw);
        const uint64_t c0 = cofs[j], c1 = c0 + n;
        if (((c0 | (uint64_t)n) & (16u * kCodecBlock - 1u)) == 0u && (((uint64_t)dst) & 15ull) == 0ull) {
            // whole codec blocks (every block of level >= 7 in 4-state: 16 KB = 4 codec blocks), the common case at full size: one WAVE per codec block, no
            // barrier -- pass k of a wave decodes units 64 k + lane, so every store instruction of the wave writes 1 KB in a row; a raw unit's rank inside
            // its codec block = the raw units of the earlier passes + those of lower lanes in this pass, from the passes' ballots
            // (the barrier form below moved the result at 1.5 TB/s: four iterations of load - ballot - barrier per workgroup and block)
            const uint32_t lane = threadIdx.x & 63u;
            for (uint64_t B = c0 / (16u * kCodecBlock) + (threadIdx.x >> 6); B * (16u * kCodecBlock) < c1; B += blockDim.x >> 6) {
                const uint8_t* codes = comp + c.offCodes + B * (kCodecBlock / 2u);
                const uint4* rawB = raw + ofs[B];
                uint4* out = (uint4*)(dst + (B * (16u * kCodecBlock) - c0));
                uint32_t before = 0;
                #pragma unroll
                for (uint32_t k = 0; k < kCodecBlock / 64u; ++k) {
                    const uint32_t u = 64u * k + lane;
                    const uint32_t code = (codes[u >> 1] >> ((u & 1u) * 4u)) & 15u;
                    const unsigned long long rb = __ballot(code == 4u);
                    uint4 v;
                    if (code == 4u) v = rawB[before + (uint32_t)__popcll(rb & ((1ull << lane) - 1ull))];
                    else { const uint32_t p = code == 0u ? 0u : (code == 1u ? 0x55555555u : (code == 2u ? 0xAAAAAAAAu : 0xFFFFFFFFu)); v = make_uint4(p, p, p, p); }
                    out[u] = v;
                    before += (uint32_t)__popcll(rb);
                }
            }
            continue;   // (block-uniform branch: no thread of this workgroup reaches the barriers below for this block)
        }
        for (uint64_t B = c0 / (16u * kCodecBlock); B * (16u * kCodecBlock) < c1; ++B) {   // (block-uniform bounds: every thread takes part in the ballots)
            const uint64_t u = B * kCodecBlock + threadIdx.x;
            const bool live = u < c.units;
            const uint32_t code = live ? (comp[c.offCodes + u / 2u] >> ((u & 1u) * 4u)) & 15u : 0u;
            uint32_t total;
            const uint32_t rank = codec_rank(live && code == 4u, s, total);
            const uint64_t b0 = u * 16u, b1 = b0 + 16u;   // this unit's bytes of the contribution
            if (live && b1 > c0 && b0 < c1) {
                uint4 v;
                if (code == 4u) v = raw[ofs[B] + rank];
                else { const uint32_t p = code == 0u ? 0u : (code == 1u ? 0x55555555u : (code == 2u ? 0xAAAAAAAAu : 0xFFFFFFFFu)); v = make_uint4(p, p, p, p); }
                if (b0 >= c0 && b1 <= c1 && (((uint64_t)(dst + (b0 - c0))) & 15ull) == 0ull) *(uint4*)(dst + (b0 - c0)) = v;
                else {
                    const uint64_t lo = b0 > c0 ? b0 : c0, hi = b1 < c1 ? b1 : c1;
                    for (uint64_t b = lo; b < hi; ++b) {
                        const uint32_t k = (uint32_t)(b - b0);
                        const uint32_t w = k < 4u ? v.x : (k < 8u ? v.y : (k < 12u ? v.z : v.w));
                        dst[b - c0] = (uint8_t)(w >> (8u * (k & 3u)));
                    }
                }
            }
            __syncthreads();   // (s[] is rewritten by the next codec block)
        }
    }
}
void launch_shard_scatter_streams(const uint8_t* streams, uint64_t streamPitch, uint64_t contributionBytes, const uint8_t* active, const uint8_t* owner,
                                  const uint32_t* stateMask, const uint8_t* level, int bits, const uint32_t* order, const uint64_t* cofs, const uint32_t* dstOfs,
                                  const uint32_t* sizes, uint32_t numOmms, uint8_t* arrayData, hipStream_t stream)
{
    if (numOmms == 0) return;
    const uint32_t grid = numOmms < 262144u ? numOmms : 262144u;
    hipLaunchKernelGGL(shard_scatter_streams, dim3(grid), dim3(256), 0, stream, streams, streamPitch, codec_layout(contributionBytes), active, owner, stateMask, level, bits,
                       order, cofs, dstOfs, sizes, numOmms, arrayData);
}
size_t shard_codec_scratch_bytes(uint64_t contributionBytes)
{
    const CodecLayout c = codec_layout(contributionBytes);
    size_t tb = 0;
    (void)rocprim::exclusive_scan(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t)0, (size_t)(c.blocks + 1), rocprim::plus<uint32_t>());
    return ((size_t)(c.blocks + 1) * 4 + 255) / 256 * 256 + tb + 256;
}
// contribution (a multiple of 256 bytes) -> comp (capacity capBytes); *sizeWord <- length of the stream in 16-byte units, or kCodecIncompressible
hipError_t run_shard_compress(const uint8_t* contrib, uint64_t contributionBytes, uint8_t* comp, uint64_t capBytes, uint32_t* sizeWord, void* scratch, size_t scratchBytes, hipStream_t stream)
{
    const CodecLayout c = codec_layout(contributionBytes);
    if (scratchBytes < shard_codec_scratch_bytes(contributionBytes) || capBytes < c.offRaw + 16u || c.blocks >= 0x7FFFFFFFull) return hipErrorInvalidValue;
    uint32_t* counts = (uint32_t*)scratch;
    void* tmp = (uint8_t*)scratch + ((size_t)(c.blocks + 1) * 4 + 255) / 256 * 256;
    size_t tb = scratchBytes - ((size_t)(c.blocks + 1) * 4 + 255) / 256 * 256;
    TAIL_CHECK(hipMemsetAsync(counts + c.blocks, 0, 4, stream));
    hipLaunchKernelGGL(shard_codec_count, dim3((uint32_t)c.blocks), dim3(256), 0, stream, (const uint4*)contrib, c, counts);
    TAIL_CHECK(rocprim::exclusive_scan(tmp, tb, counts, (uint32_t*)(comp + c.offOfs), (uint32_t)0, (size_t)(c.blocks + 1), rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(shard_codec_write, dim3((uint32_t)c.blocks), dim3(256), 0, stream, (const uint4*)contrib, c, comp, capBytes);
    hipLaunchKernelGGL(shard_codec_finish, dim3(1), dim3(1), 0, stream, c, comp, capBytes, sizeWord);
    return hipGetLastError();
}
hipError_t run_shard_compress_coded(const uint8_t* contrib, uint64_t contributionBytes, const uint8_t* unitCodes, uint32_t* blockRawCounts, uint8_t* comp, uint64_t capBytes,
                                    uint32_t* sizeWord, void* scratch, size_t scratchBytes, hipStream_t stream)
{
    const CodecLayout c = codec_layout(contributionBytes);
    if (scratchBytes < shard_codec_scratch_bytes(contributionBytes) || capBytes < c.offRaw + 16u || c.blocks >= 0x7FFFFFFFull) return hipErrorInvalidValue;
    void* tmp = (uint8_t*)scratch + ((size_t)(c.blocks + 1) * 4 + 255) / 256 * 256;
    size_t tb = scratchBytes - ((size_t)(c.blocks + 1) * 4 + 255) / 256 * 256;
    TAIL_CHECK(rocprim::exclusive_scan(tmp, tb, blockRawCounts, (uint32_t*)(comp + c.offOfs), (uint32_t)0, (size_t)(c.blocks + 1), rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(shard_codec_write_coded, dim3((uint32_t)c.blocks), dim3(256), 0, stream, (const uint4*)contrib, unitCodes, c, comp, capBytes);
    hipLaunchKernelGGL(shard_codec_finish, dim3(1), dim3(1), 0, stream, c, comp, capBytes, sizeWord);
    return hipGetLastError();
}
void launch_shard_gather(const uint8_t* states, const uint64_t* stateOfs, const uint8_t* active, const uint8_t* owner, uint32_t rank, const uint32_t* order,
                         const uint64_t* cofs, const uint32_t* sizes, uint32_t numOmms, uint8_t* contrib, hipStream_t stream)
{
    if (numOmms == 0) return;
    const uint32_t grid = numOmms < 262144u ? numOmms : 262144u;
    hipLaunchKernelGGL(shard_gather_contribution, dim3(grid), dim3(256), 0, stream, states, stateOfs, active, owner, rank, order, cofs, sizes, numOmms, contrib);
}
void launch_shard_scatter(const uint8_t* gathered, uint64_t rankPitch, uint64_t lo, uint64_t hi, const uint8_t* active, const uint8_t* owner, const uint32_t* stateMask,
                          const uint8_t* level, int bits, const uint32_t* order, const uint64_t* cofs, const uint32_t* dstOfs, const uint32_t* sizes,
                          uint32_t numOmms, uint8_t* arrayData, hipStream_t stream)
{
    if (numOmms == 0) return;
    const uint32_t grid = numOmms < 262144u ? numOmms : 262144u;
    hipLaunchKernelGGL(shard_scatter_contributions, dim3(grid), dim3(256), 0, stream, gathered, rankPitch, lo, hi, active, owner, stateMask, level, bits, order, cofs,
                       dstOfs, sizes, numOmms, arrayData);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Streamed result (ommCpuBake): OMM blocks leave for the host WHILE the classification runs, straight to their final arrayData offsets.
//
// The final order of the blocks is known before anything is classified: descending (level, Morton key of the centroid, item index)
// over the items that end up EMITTED (bake_cpu_impl.cpp:1707-1754).  So the per-level lists of active items are sorted into that order
// first (run_stream_begin), the tile queue of the levels >= 6 is cut into ranges of it (classify_plan; ONE persistent launch drains them in
// order and counts each range's finished tiles: bake_kernels.hip), and on a second, high-priority stream, behind a one-lane kernel that waits
// for the range's count (stream_wait_sections), run_stream_segment() decides which of the range's items are emitted -- non-uniform, not
// rejected, first occurrence of their digest --, scans their sizes and packs their blocks behind the blocks of the earlier ranges: a
// contiguous piece of the final arrayData, which one SDMA copy moves to the host while the launch classifies the following ranges.
//
// "First occurrence of the digest" (DeduplicateExact, bake_cpu_impl.cpp:1031-1066: the LOWEST work-item index keeps the block) needs the
// digests of every item that could carry the same block.  The preview below finds those families before the classification; the members of a
// family are classified with the range of its FIRST member (never later than their own: early_range), so when a range is placed every digest
// that can decide about its blocks is in the table.  What the preview cannot see (items it does not cover, a family it split) makes the
// placement speculative in exactly one respect: a later range may bring a LOWER index for a digest that an earlier range has already
// emitted.  That event is detected (claimed[] below) and, belt and braces, the complete result layout of the ordinary tail is compared with
// the streamed one at the end (stream_verify); on any difference the bake falls back to the ordinary gather + copy.  The DATA of a placed
// block is complete by construction: all tiles of an item lie in sections of its own range or an earlier one, and a range is placed only
// when both of its sections have reported every tile (device-scope release / acquire on the section's count).
// ------------------------------------------------------------------------------------------------------------------------------
struct StreamScratch {
    HashTable table; uint8_t* claimed;        // digest -> lowest emitted-candidate index; claimed[slot] = 1 + range that emitted a block for it
    uint64_t *sizes64, *ofs64, *keysA, *keysB; void* tmp; size_t tmpBytes;
};
static size_t stream_prim_temp_bytes(uint32_t n)
{
    size_t a = 0, c = 0;
    (void)rocprim::radix_sort_keys(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0u, 62u);
    (void)rocprim::exclusive_scan(nullptr, c, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>());
    return ((a > c ? a : c) + 255) / 256 * 256 + 256;
}
static size_t align256(size_t v) { return (v + 255) / 256 * 256; }
size_t stream_scratch_bytes(uint32_t numActive)
{
    const uint32_t n = numActive ? numActive : 1, slots = hash_table_slots(n);
    return align256(hash_table_bytes(slots)) + align256((size_t)slots + 1) + 4 * align256((size_t)n * 8) + stream_prim_temp_bytes(n);
}
static StreamScratch stream_carve(void* base, uint32_t numActive)
{
    const uint32_t n = numActive ? numActive : 1, slots = hash_table_slots(n);
    StreamScratch s; uint8_t* p = (uint8_t*)base;
    s.table = hash_table_at(p, slots); p += align256(hash_table_bytes(slots));
    s.claimed = p; p += align256((size_t)slots + 1);
    s.sizes64 = (uint64_t*)p; p += align256((size_t)n * 8); s.ofs64 = (uint64_t*)p; p += align256((size_t)n * 8);
    s.keysA = (uint64_t*)p; p += align256((size_t)n * 8); s.keysB = (uint64_t*)p; p += align256((size_t)n * 8);
    s.tmp = p; s.tmpBytes = stream_prim_temp_bytes(n);
    return s;
}

// sort key of position p of the active list: level ascending (the list is already grouped that way), then Morton key DEscending, then item index DEscending
__global__ __launch_bounds__(256) void stream_sort_keys(const uint32_t* __restrict__ activeIds, uint32_t n, const float* __restrict__ uv, const uint8_t* __restrict__ level, uint64_t* __restrict__ keys)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t item = activeIds[p], lvl = level[item];
    const uint32_t morton = spatial_key30(uv + 6ull * item, lvl) & 0x3FFFFFFu;
    keys[p] = ((uint64_t)lvl << 58) | ((uint64_t)(0x3FFFFFFu - morton) << 32) | (uint64_t)(0xFFFFFFFFu - item);
}
__global__ __launch_bounds__(256) void stream_sort_unpack(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ activeIds)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) activeIds[p] = 0xFFFFFFFFu - (uint32_t)keys[p];
}

// sorts every level's sub-list of activeIds into the order of the final result and clears the digest table of the streamed placement
hipError_t run_stream_begin(uint32_t* activeIds, uint32_t numActive, const float* uv, const uint8_t* level, void* scratch, size_t scratchBytes, hipStream_t stream)
{
    if (numActive == 0) return hipSuccess;
    if (scratchBytes < stream_scratch_bytes(numActive)) return hipErrorInvalidValue;
    StreamScratch s = stream_carve(scratch, numActive);
    const dim3 grid((numActive + 255u) / 256u), block(256);
    hipLaunchKernelGGL(stream_sort_keys, grid, block, 0, stream, (const uint32_t*)activeIds, numActive, uv, level, s.keysA);
    size_t tb = s.tmpBytes;
    TAIL_CHECK(rocprim::radix_sort_keys(s.tmp, tb, s.keysA, s.keysB, (size_t)numActive, 0u, 62u, stream));
    hipLaunchKernelGGL(stream_sort_unpack, grid, block, 0, stream, (const uint64_t*)s.keysB, numActive, activeIds);
    const uint32_t slots = s.table.mask + 1u;
    TAIL_CHECK(hipMemsetAsync(s.table.keys, 0xFF, hash_table_bytes(slots), stream));
    TAIL_CHECK(hipMemsetAsync(s.claimed, 0, (size_t)slots + 1, stream));
    return hipGetLastError();
}

// ---- preview: which work items could end up as duplicates of each other? ----
// Identical blocks of DIFFERENT triangles are not rare: the patterns that a (locally straight) alpha edge cuts out of a triangle come from small
// families -- a corner or edge point that is just touched (a handful of unknown micro-triangles, one state everywhere else), k complete rows of
// micro-triangles parallel to an edge, ... -- measured on the bench workload: 959 of 77 627 blocks are shared by 2 .. 4 triangles.
// Two items with the same block have the same block at every coarser level too (a coarse micro-triangle is T / O exactly when all of its
// descendants are), so the preview classifies every active item of level >= 6 at level 5 (1024 micro-triangles: one 1024-tile each through the ordinary
// tile triage + persistent launch, with 4-state / ForceOpaque parameters and buffers of its own: < 2 % of the work of the bake), hashes the 256 bytes, and marks as `early` every item
// whose preview (a) is not one single state -- those become special indices, not blocks -- and (b) is shared with another item: a family.
// On the bench workload 41 425 of the 79 869 items with a mixed preview are early (8 392 families: the digital lines a 32 x 32 lattice can tell
// apart are few).  Classifying them all before the first range held the first copy back by 10 ms (their tiles are a third of the open tiles), so
// a family is classified with the range of its first member instead (stream_preview_leaders: lead[], early_range; the tile triage routes its
// tiles into that range's second queue section) and is digested and entered into the table with that range.  Items whose preview micro-triangles
// would be large (> 256 texels each: asset-sized triangles) are not previewed (a collapsed triangle stands in for them) and never early.  A wrong
// guess costs speed only: the placement is verified.
__global__ __launch_bounds__(256) void stream_preview_prepare(const uint32_t* __restrict__ ids, uint32_t n, const float* __restrict__ uv, float texW, float texH,
                                                              float* __restrict__ uv2, uint64_t* __restrict__ ofs2, uint8_t* __restrict__ early)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t item = ids[p];
    const float* t = uv + 6ull * item; float* o = uv2 + 6ull * item;
    const float lx = fminf(fminf(t[0], t[2]), t[4]), hx = fmaxf(fmaxf(t[0], t[2]), t[4]), ly = fminf(fminf(t[1], t[3]), t[5]), hy = fmaxf(fmaxf(t[1], t[3]), t[5]);
    const bool large = !((hx - lx) * texW * (hy - ly) * texH <= 262144.f);   // (NaN-safe: anything odd counts as large)
    for (int k = 0; k < 6; ++k) o[k] = large ? t[k & 1] : t[k];               // large: the three vertices collapse onto vertex 0
    ofs2[item] = (uint64_t)item * kPreviewSlotBytes;
    early[item] = large ? 2 : 0;                                              // 2 = not previewed
}
// signature of the preview: 64-bit hash of its 256 bytes (and the level); previews of one single state and items that were not previewed get none.
// The table keeps, per signature, the FIRST position (in the order of the final result) that carries it: the family's first member.
__global__ __launch_bounds__(256) void stream_preview_signature(const uint32_t* __restrict__ ids, uint32_t n, const uint8_t* __restrict__ states2, const uint8_t* __restrict__ level,
                                                                uint8_t* __restrict__ early, uint64_t* __restrict__ sig, HashTable table)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t item = ids[p];
    if (early[item] == 2) { early[item] = 0; sig[p] = kEmptyKey; return; }
    const uint4* w = (const uint4*)(states2 + (size_t)item * kPreviewSlotBytes);
    // (the level goes through a mixing round of its own: XORed straight into the seed it cancels against the first data word -- level 10 / first byte 03 and
    //  level 6 / first byte 0f gave one signature, a family across levels)
    uint64_t h = (0x9E3779B97F4A7C15ull ^ (uint64_t)level[item]) * 0xff51afd7ed558ccdull; h ^= h >> 32;
    bool uniform = true; const uint32_t w0 = w[0].x;
    for (uint32_t k = 0; k < kPreviewSlotBytes / 16u; ++k) {
        const uint4 v = w[k];
        uniform = uniform && v.x == w0 && v.y == w0 && v.z == w0 && v.w == w0;
        h = (h ^ (((uint64_t)v.y << 32) | v.x)) * 0xff51afd7ed558ccdull; h ^= h >> 32;
        h = (h ^ (((uint64_t)v.w << 32) | v.z)) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29;
    }
    uniform = uniform && (w0 == 0u || w0 == 0x55555555u || w0 == 0xAAAAAAAAu || w0 == 0xFFFFFFFFu);
    if (h == kEmptyKey) h = 0;
    sig[p] = uniform ? kEmptyKey : h;
    if (!uniform) hash_put_min(table, h, p);
}
// every member of a family of >= 2 items is early: the later members see another first position, and tell it
__global__ __launch_bounds__(256) void stream_preview_followers(const uint32_t* __restrict__ ids, uint32_t n, const uint64_t* __restrict__ sig, HashTable table,
                                                                uint8_t* __restrict__ followed, uint8_t* __restrict__ early)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || sig[p] == kEmptyKey) return;
    const uint32_t slot = hash_find_slot(table, sig[p]);
    if (slot != 0xFFFFFFFFu && table.vals[slot] != p) { early[ids[p]] = 1; followed[slot] = 1; }
}
// ... the first members join, every early item learns the position of its family's first member (lead[item], a position of the whole active list) and
// is counted for that member's range: ctl[kStreamCtlEarlyCount + range]
__global__ __launch_bounds__(256) void stream_preview_leaders(const uint32_t* __restrict__ ids, uint32_t n, uint32_t listOffset, const uint64_t* __restrict__ sig, HashTable table,
                                                              const uint8_t* __restrict__ followed, uint8_t* __restrict__ early, uint32_t* __restrict__ lead,
                                                              uint32_t* __restrict__ ctl, TileLevels L, TileSections S)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    bool e = false; uint32_t range = 0;
    if (p < n && sig[p] != kEmptyKey) {
        const uint32_t item = ids[p], slot = hash_find_slot(table, sig[p]);
        if (slot != 0xFFFFFFFFu) {
            const uint32_t first = table.vals[slot];
            if (first == p && followed[slot]) early[item] = 1;
            e = early[item] == 1;
            if (e) { lead[item] = listOffset + first; range = early_range(listOffset + first, listOffset + p, L, S); }
        }
    }
    const unsigned long long b = __ballot(e);
    unsigned long long todo = b;
    while (todo) {   // (wave-aggregated: neighbours in the final order mostly share their range)
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t r0 = (uint32_t)__shfl((int)range, leader);
        const unsigned long long same = __ballot(e && range == r0) & todo;
        todo &= ~same;
        if ((int)lane == leader) atomicAdd(ctl + kStreamCtlEarlyCount + r0, (uint32_t)__popcll(same));
    }
    if (lane == 0 && b) atomicAdd(ctl + 3, (uint32_t)__popcll(b));   // statistics: size of the early class
}
// the early class as one list ordered by the range its items are classified in: starts ...
__global__ void stream_early_starts(uint32_t* __restrict__ ctl, uint32_t ranges)
{
    uint32_t run = 0;
    for (uint32_t k = 0; k < ranges; ++k) { ctl[kStreamCtlEarlyStart + k] = run; run += ctl[kStreamCtlEarlyCount + k]; }
}
// ... and entries
__global__ __launch_bounds__(256) void stream_early_scatter(const uint32_t* __restrict__ ids, uint32_t n, uint32_t listOffset, const uint8_t* __restrict__ early, const uint32_t* __restrict__ lead,
                                                            uint32_t* __restrict__ ctl, uint32_t* __restrict__ earlyList, TileLevels L, TileSections S)
{
    __shared__ uint32_t s_cnt[kMaxStreamRanges], s_base[kMaxStreamRanges];   // (one global atomic per range and workgroup)
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < kMaxStreamRanges) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t item = p < n ? ids[p] : 0u;
    const bool e = p < n && early[item] == 1;
    uint32_t range = 0, local = 0;
    if (e) { range = early_range(lead[item], listOffset + p, L, S); local = atomicAdd(&s_cnt[range], 1u); }
    __syncthreads();
    if (threadIdx.x < kMaxStreamRanges && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(ctl + kStreamCtlEarlyFill + threadIdx.x, s_cnt[threadIdx.x]);
    __syncthreads();
    if (e) earlyList[ctl[kStreamCtlEarlyStart + range] + s_base[range] + local] = item;
}
void launch_stream_preview_prepare(const uint32_t* ids, uint32_t n, const float* uv, float texW, float texH, float* uv2, uint64_t* ofs2, uint8_t* early, hipStream_t stream)
{
    if (n) hipLaunchKernelGGL(stream_preview_prepare, dim3((n + 255u) / 256u), dim3(256), 0, stream, ids, n, uv, texW, texH, uv2, ofs2, early);
}
// after the preview classification.  Uses the (still empty) digest table of the streamed placement for the signatures and clears it again.
hipError_t run_stream_preview_flags(const uint32_t* ids, uint32_t n, uint32_t listOffset, uint32_t numActive, const uint8_t* states2, const uint8_t* level, uint8_t* early,
                                    uint32_t* ctl, void* scratch, size_t scratchBytes, uint32_t* earlyLead, uint32_t* earlyList, const ClassifyPlan& plan, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (scratchBytes < stream_scratch_bytes(numActive)) return hipErrorInvalidValue;
    StreamScratch s = stream_carve(scratch, numActive);
    const dim3 grid((n + 255u) / 256u), block(256);
    const uint32_t slots = s.table.mask + 1u;
    hipLaunchKernelGGL(stream_preview_signature, grid, block, 0, stream, ids, n, states2, level, early, s.sizes64, s.table);   // (sizes64: free until the first segment)
    hipLaunchKernelGGL(stream_preview_followers, grid, block, 0, stream, ids, n, (const uint64_t*)s.sizes64, s.table, s.claimed, early);
    hipLaunchKernelGGL(stream_preview_leaders, grid, block, 0, stream, ids, n, listOffset, (const uint64_t*)s.sizes64, s.table, (const uint8_t*)s.claimed, early, earlyLead, ctl,
                       plan.big, plan.ranges);
    hipLaunchKernelGGL(stream_early_starts, dim3(1), dim3(1), 0, stream, ctl, plan.ranges.n);
    hipLaunchKernelGGL(stream_early_scatter, grid, block, 0, stream, ids, n, listOffset, (const uint8_t*)early, (const uint32_t*)earlyLead, ctl, earlyList, plan.big, plan.ranges);
    TAIL_CHECK(hipMemsetAsync(s.table.keys, 0xFF, hash_table_bytes(slots), stream));
    TAIL_CHECK(hipMemsetAsync(s.claimed, 0, (size_t)slots + 1, stream));
    return hipGetLastError();
}

// can this item become a block?  non-uniform and not rejected (PromoteToSpecialIndices, bake_cpu_impl.cpp:1432-1472; special indices are enabled in streamed bakes)
__device__ __forceinline__ bool stream_candidate(const StreamSegment& g, uint32_t item)
{
    const uint32_t mask = g.stateMask[item];
    if ((mask & (mask - 1u)) == 0u) return false;
    if (g.rejectionThreshold > 0.f) {
        const uint32_t lvl = g.itemLevel ? (uint32_t)g.itemLevel[item] : g.level;
        const float frac = (float)g.knownCount[item] / (float)(1u << (2u * lvl));
        if (frac < g.rejectionThreshold) return false;
    }
    return true;
}
// (g.liveCount: the ids are a device-side slice, g.ids[*g.liveStart .. + *g.liveCount), of a list of capacity g.count)
__global__ __launch_bounds__(256) void stream_insert(StreamSegment g, HashTable table)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.count) return;
    if (g.liveCount) { if (p >= *g.liveCount) return; p += *g.liveStart; }
    const uint32_t item = g.ids[p];
    if (stream_candidate(g, item)) hash_put_min(table, g.digests[item], item);
}
// the early items that were classified with a range (a slice of the early list, mixed levels): their digests enter the table with the range's own
void launch_stream_insert_list(const StreamSegment& g, uint32_t numActive, void* scratch, size_t scratchBytes, hipStream_t stream)
{
    if (g.count == 0 || g.disableDedup || scratchBytes < stream_scratch_bytes(numActive)) return;
    StreamScratch s = stream_carve(scratch, numActive);
    hipLaunchKernelGGL(stream_insert, dim3((g.count + 255u) / 256u), dim3(256), 0, stream, g, s.table);
}
__global__ __launch_bounds__(256) void stream_flags(StreamSegment g, HashTable table, uint8_t* __restrict__ claimed, uint64_t* __restrict__ sizes64, uint32_t* __restrict__ ctl)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.count) return;
    const uint32_t item = g.ids[p];
    bool emit = stream_candidate(g, item);
    if (emit && !g.disableDedup) {
        const uint32_t slot = hash_find_slot(table, g.digests[item]);
        if (slot == 0xFFFFFFFFu) { atomicOr(ctl + 1, 1u); emit = false; }
        else if (table.vals[slot] != item) emit = false;                 // a lower index with this digest exists in this or an earlier range: that one keeps the block
        else {
            const uint8_t tag = (uint8_t)(g.range + 1u);
            if (claimed[slot] != 0 && claimed[slot] != tag) atomicOr(ctl + 1, 1u);   // an EARLIER range emitted a block for this digest, and this index is lower: misplaced
            claimed[slot] = tag;
        }
    }
    uint64_t bytes = 0;
    if (emit) { bytes = (((uint64_t)1 << (2u * g.level)) * (uint64_t)g.bits) >> 3; if (bytes < 1) bytes = 1; }   // Serialize: at least one byte per OMM (bake_cpu_impl.cpp:1768-1772)
    sizes64[p] = bytes;
    const unsigned long long b = __ballot(emit);
    if ((threadIdx.x & 63u) == 0 && b) atomicAdd(ctl, (uint32_t)__popcll(b));   // number of blocks placed so far
}
// exclusive scan of up to a few ten thousand sizes by one workgroup (wave scans + a carried total)
__global__ __launch_bounds__(256) void stream_scan_small(const uint64_t* __restrict__ sizes, uint64_t* __restrict__ ofs, uint32_t n)
{
    __shared__ uint64_t s_wave[4];
    uint64_t carry = 0;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i0 = 0; i0 < n; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint64_t v = i < n ? sizes[i] : 0ull;
        uint64_t x = v;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        if (lane == 63u) s_wave[wave] = x;
        __syncthreads();
        uint64_t before = 0, total = 0;
        for (uint32_t w = 0; w < 4u; ++w) { before += w < wave ? s_wave[w] : 0ull; total += s_wave[w]; }
        if (i < n) ofs[i] = carry + before + x - v;
        carry += total;
        __syncthreads();
    }
}
// one wave per item: the block goes behind everything placed so far
__global__ __launch_bounds__(256) void stream_copy(StreamSegment g, const uint64_t* __restrict__ sizes64, const uint64_t* __restrict__ ofs64,
                                                   const unsigned long long* __restrict__ cursor, uint8_t* __restrict__ stage, uint64_t* __restrict__ placed)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    const unsigned long long base = *cursor;
    for (uint32_t p = wave; p < g.count; p += waves) {
        const uint32_t item = g.ids[p];
        const uint64_t bytes = sizes64[p];
        if (bytes == 0) { if (lane == 0) placed[item] = ~0ull; continue; }
        const unsigned long long dstOfs = base + ofs64[p];
        const uint8_t* src = g.states + g.stateOfs[item]; uint8_t* dst = stage + dstOfs;
        if (((bytes | dstOfs) & 15ull) == 0) { const uint4* s4 = (const uint4*)src; uint4* d4 = (uint4*)dst; for (uint64_t k = lane; k < bytes / 16u; k += 64u) d4[k] = s4[k]; }
        else for (uint64_t k = lane; k < bytes; k += 64u) dst[k] = src[k];
        if (lane == 0) placed[item] = dstOfs;
    }
}
__global__ void stream_advance(unsigned long long* __restrict__ cursor, const uint64_t* __restrict__ sizes64, const uint64_t* __restrict__ ofs64, uint32_t count)
{
    *cursor += ofs64[count - 1u] + sizes64[count - 1u];
}
__global__ void stream_publish(const unsigned long long* __restrict__ cursor, volatile unsigned long long* __restrict__ hostSlot) { *hostSlot = *cursor; }

// one segment (items of ONE level, consecutive in the sorted active list) behind the classification launch that finished them.
// Digests of the segment's items must have been computed (launch_digest) unless dedup is disabled.
hipError_t run_stream_segment(const StreamSegment& g, uint32_t numActive, void* scratch, size_t scratchBytes, unsigned long long* cursor, uint8_t* stage,
                              uint64_t* placed, uint32_t* ctl, hipStream_t stream)
{
    if (g.count == 0) return hipSuccess;
    if (scratchBytes < stream_scratch_bytes(numActive)) return hipErrorInvalidValue;
    StreamScratch s = stream_carve(scratch, numActive);
    const dim3 grid((g.count + 255u) / 256u), block(256);
    if (!g.disableDedup) hipLaunchKernelGGL(stream_insert, grid, block, 0, stream, g, s.table);
    hipLaunchKernelGGL(stream_flags, grid, block, 0, stream, g, s.table, s.claimed, s.sizes64, ctl);
    // (a range has a few thousand items: one small workgroup scans them -- 16 VGPRs, so that it runs next to the persistent classification launch like the
    //  rest of the placement; the lower levels, behind that launch and possibly millions of items, keep rocPRIM)
    if (g.count <= 65536u) hipLaunchKernelGGL(stream_scan_small, dim3(1), dim3(256), 0, stream, (const uint64_t*)s.sizes64, s.ofs64, g.count);
    else {
        size_t tb = s.tmpBytes;
        TAIL_CHECK(rocprim::exclusive_scan(s.tmp, tb, s.sizes64, s.ofs64, (uint64_t)0, (size_t)g.count, rocprim::plus<uint64_t>(), stream));
    }
    const uint32_t blocks = (g.count + 3u) / 4u;
    hipLaunchKernelGGL(stream_copy, dim3(blocks < 8192u ? blocks : 8192u), dim3(256), 0, stream, g, (const uint64_t*)s.sizes64, (const uint64_t*)s.ofs64,
                       (const unsigned long long*)cursor, stage, placed);
    hipLaunchKernelGGL(stream_advance, dim3(1), dim3(1), 0, stream, cursor, (const uint64_t*)s.sizes64, (const uint64_t*)s.ofs64, g.count);
    return hipGetLastError();
}
// Holds the placement stream until sections [first, first + n) of the tile queue are complete: done == tail (bake_kernels.hip: classify_tiles).  One lane, asleep
// between two looks.  The tail is final (the stream was fenced behind the tile triage); a count that never arrives -- it cannot, short of a fault in
// the classification launch -- ends the wait after ~4 s of the 100 MHz wall clock with the violation word set, which discards the streamed result.
__global__ void stream_wait_sections(const uint32_t* __restrict__ queueCtl, uint32_t first, uint32_t n, uint32_t* __restrict__ ctl, unsigned long long timeoutTicks)
{
    const uint64_t t0 = wall_clock64();
    for (uint32_t sec = first; sec < first + n; ++sec) {
        const uint32_t want = __hip_atomic_load(queueCtl + kSecTails + sec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(queueCtl + kSecDone + sec, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) {
            __builtin_amdgcn_s_sleep(64);
            if (wall_clock64() - t0 > timeoutTicks) { atomicOr(ctl + 1, 1u); return; }   // (100 MHz ticks; the host scales it with the size of the bake)
        }
    }
}
void launch_stream_wait_sections(const uint32_t* queueCtl, uint32_t first, uint32_t n, uint32_t* ctl, double timeoutSeconds, hipStream_t stream)
{
    const double ticks = (timeoutSeconds < 4.0 ? 4.0 : (timeoutSeconds > 3600.0 ? 3600.0 : timeoutSeconds)) * 1e8;
    hipLaunchKernelGGL(stream_wait_sections, dim3(1), dim3(1), 0, stream, queueCtl, first, n, ctl, (unsigned long long)ticks);
}
void launch_stream_publish(const unsigned long long* cursor, unsigned long long* hostSlot, hipStream_t stream)
{
    hipLaunchKernelGGL(stream_publish, dim3(1), dim3(1), 0, stream, cursor, (volatile unsigned long long*)hostSlot);
}

// the ordinary tail has produced the exact layout: is the streamed placement the same?  ctl[0] = blocks placed, ctl[1] = violation seen, ctl[2] <- mismatch
__global__ __launch_bounds__(256) void stream_verify(const uint32_t* __restrict__ order, const uint32_t* __restrict__ dstOfs, uint32_t numOmms, const uint64_t* __restrict__ placed,
                                                     uint32_t* __restrict__ ctl)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && (ctl[0] != numOmms || ctl[1] != 0u)) atomicOr(ctl + 2, 1u);
    if (j < numOmms && placed[order[j]] != (uint64_t)dstOfs[j]) atomicOr(ctl + 2, 1u);
}
void launch_stream_verify(const uint32_t* order, const uint32_t* dstOfs, uint32_t numOmms, const uint64_t* placed, uint32_t* ctl, hipStream_t stream)
{
    hipLaunchKernelGGL(stream_verify, dim3((numOmms + 256u) / 256u), dim3(256), 0, stream, order, dstOfs, numOmms, placed, ctl);
}

// ---- scratch layout ----
struct Scratch {
    unsigned long long *keysA, *keysB;   // sort keys of the emitted items (spatial key << idxBits | item), unsorted / sorted
    void* hashBase;                      // digest table (hash_build.h) + kUniformBins words; directly in front of keysA: one fill sets both to all ones
    TailWork* work;                      // counters + read-back words, directly in front of tileState: one fill zeroes both
    unsigned long long* tileState;       // tail_place's look-back states
    void* tmp; size_t tmpBytes;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t prim_temp_bytes(uint32_t n)
{
    size_t a = 0;
    (void)rocprim::radix_sort_keys(nullptr, a, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)n, 0u, 62u);
    return align_up(a, 256) + 256;
}
static uint32_t place_tiles(uint32_t n) { return (n + kPlaceTile - 1u) / kPlaceTile; }
static size_t tail_hash_bytes(uint32_t n) { return align_up(hash_table_bytes(hash_table_slots(n), kUniformBins), 256); }

// (the block is laid out for numItems keys and distinct digests; a run uses the front of each part)
static Scratch carve(void* base, uint32_t n, uint32_t hashSlots)
{
    Scratch s; uint8_t* p = (uint8_t*)base;
    const size_t n64 = align_up((size_t)n * 8, 256);
    const size_t hb = align_up(hash_table_bytes(hashSlots, kUniformBins), 256);
    s.hashBase = p; p += hb; s.keysA = (unsigned long long*)p; p = (uint8_t*)base + tail_hash_bytes(n) + n64;
    s.keysB = (unsigned long long*)p; p += n64;
    s.work = (TailWork*)p; p += 256; s.tileState = (unsigned long long*)p; p += align_up((size_t)place_tiles(n) * 8, 256);
    s.tmp = p; s.tmpBytes = prim_temp_bytes(n);
    return s;
}

size_t tail_scratch_bytes(uint32_t numItems, uint32_t numTris)
{
    (void)numTris;
    const uint32_t n = numItems ? numItems : 1;
    const size_t tail = 2 * align_up((size_t)n * 8, 256) + tail_hash_bytes(n) + 256 + align_up((size_t)place_tiles(n) * 8, 256) + prim_temp_bytes(n);
    const size_t prep = (size_t)prep_state_words(n) * 4;   // (run_prep's tile states live in the same block)
    return tail > prep ? tail : prep;
}

hipError_t run_tail(const TailInputs& in, const TailOutputs& out, void* scratch, size_t scratchBytes, TailCounts* counts, hipStream_t stream, void* hostWork)
{
    const uint32_t n = in.numItems;
    counts->numOmms = 0; counts->arrayDataSize = 0; counts->smallOmms = 0;
    // (the bake takes the two histograms and the error word from its arena back to back: tail_summarize zeroes them with everything else)
    const bool adjacent = out.indexHist == out.arrayHist + 64 && in.errorFlag == out.arrayHist + 128;
    if (!adjacent || n == 0) {
        TAIL_CHECK(hipMemsetAsync(out.arrayHist, 0, sizeof(uint32_t) * kNumLevels, stream));
        TAIL_CHECK(hipMemsetAsync(out.indexHist, 0, sizeof(uint32_t) * kNumLevels, stream));
        TAIL_CHECK(hipMemsetAsync(in.errorFlag, 0, sizeof(uint32_t), stream));
    }
    TailWork* workDev = nullptr; uint32_t workBound = 0;
    if (n != 0) {
        if (scratchBytes < tail_scratch_bytes(n, in.numTris)) return hipErrorInvalidValue;
        // emitted items are first occurrences that are not special: with special indices only non-uniform items (all of them on the active lists), with
        // dedup at most one more per uniform (level, state) -- maxDistinctDigests covers both; with neither, every item
        const bool every = (in.disableDedup && in.disableSpecial) || in.maxDistinctDigests == 0u;
        const uint32_t bound = every || in.maxDistinctDigests > n ? n : in.maxDistinctDigests;
        const uint32_t distinct = in.maxDistinctDigests && in.maxDistinctDigests < n ? in.maxDistinctDigests : n;
        const uint32_t slots = in.disableDedup ? 0u : hash_table_slots(distinct);
        uint32_t idxBits = 1; while (idxBits < 32u && (1ull << idxBits) < (unsigned long long)n) ++idxBits;
        Scratch s = carve(scratch, n, slots ? slots : 1024u);
        const HashTable table = hash_table_at(s.hashBase, slots ? slots : 1024u);
        uint32_t* firstUniform = table.vals + (slots ? slots : 1024u) + 1u;   // keys, values and the uniform bins are adjacent
        const bool rank = bound <= kRankMax;
        const dim3 grid((n + 255u) / 256u), block(256);
        // fills: table (when there is one) + key list to all ones, in one piece; work words + tile states to zero; histograms + error word to zero
        uint32_t* ones = slots ? (uint32_t*)s.hashBase : (uint32_t*)s.keysA;
        const uint32_t onesWords = (uint32_t)(((uint8_t*)(s.keysA + bound) - (uint8_t*)ones) / 4);
        const uint32_t zeroBWords = 64u + (rank ? 0u : 2u * place_tiles(bound));
        hipLaunchKernelGGL(tail_summarize, grid, block, 0, stream, in, out.special, ones, onesWords, adjacent ? out.arrayHist : (uint32_t*)nullptr, adjacent ? 129u : 0u,
                           (uint32_t*)s.work, zeroBWords);
        if (!in.disableDedup)
        {
            hipLaunchKernelGGL(dedup_insert, grid, block, 0, stream, in.digests, in.stateMask, in.level, in.uniformDigest ? 1 : 0, n, table, firstUniform);
            if (in.uniformDigest) hipLaunchKernelGGL(dedup_insert_bins, dim3(1), dim3(64), 0, stream, in.uniformDigest, firstUniform, table);
        }
        hipLaunchKernelGGL(tail_emit, grid, block, 0, stream, in, out.special, table, idxBits, bound, out.rep, out.itemValue, s.keysA, s.work);
        if (rank)
            hipLaunchKernelGGL(tail_rank_place, dim3((bound + 31u) / 32u), dim3(1024), 0, stream, (const unsigned long long*)s.keysA, idxBits, bound, in.format, s.work,
                               out.order, out.dstOfs, out.sizes, out.itemValue, out.arrayHist);
        else {
            size_t tb = s.tmpBytes;
            TAIL_CHECK(rocprim::radix_sort_keys(s.tmp, tb, s.keysA, s.keysB, (size_t)bound, 0u, idxBits + 30u, stream));
            hipLaunchKernelGGL(tail_place, dim3(place_tiles(bound)), dim3(kPlaceTile), 0, stream, (const unsigned long long*)s.keysB, idxBits, bound, in.format, s.work, s.tileState,
                               out.order, out.dstOfs, out.sizes, out.itemValue, out.arrayHist);
        }
        workDev = s.work; workBound = bound;
    }
    // (the index buffer does not wait for the read-back: it runs while the host synchronises)
    if (in.numTris != 0)
        hipLaunchKernelGGL(tail_indices, dim3((in.numTris + 255u) / 256u), dim3(256), 0, stream, in, out.rep, out.itemValue, out.indexBuffer, out.indexHist, out.narrowIndex, out.narrowBytes);
    if (workDev) {
        TailWork local; memset(&local, 0, sizeof local);
        TailWork* dst = hostWork ? (TailWork*)hostWork : &local;
        TAIL_CHECK(hipMemcpyAsync(dst, workDev, sizeof local, hipMemcpyDeviceToHost, stream));
        TAIL_CHECK(hipStreamSynchronize(stream));
        const TailWork w = *dst;
        counts->numOmms = w.numEmitted < workBound ? w.numEmitted : workBound; counts->arrayDataSize = w.arrayBytes; counts->smallOmms = (uint32_t)w.smallOmms;
        if (w.err) return hipErrorAssert;
    }
    return hipGetLastError();
}

} // namespace ommx
