// Single-pass prefix sums with decoupled look-back (Merrill & Garland), as the three compactions of the bake use them (work items of the triangles, active
// items + their state slots, array offsets of the OMMs): a tile publishes its total (flag 1), collects its exclusive prefix from the tiles in front of it
// until it meets one whose inclusive prefix is known (flag 2), then publishes its own inclusive prefix.  Tiles are handed out by a ticket, so every tile in
// front of a running one is running or done and the wait cannot deadlock.  State word: flag << 62 | value; all zero = nothing published (the words are
// zeroed by a launch in front of the scan).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ommx {

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// All 64 lanes of ONE wave of the tile's workgroup call this with the tile's total; every lane gets the exclusive prefix.  Lane l inspects the l-th tile in
// front, 64 at a time: the look-back of a launch with thousands of tiles in flight is a few steps, not a walk over every tile that has only its total out.
__device__ __forceinline__ unsigned long long lookback_exclusive(unsigned long long* __restrict__ state, uint32_t tile, unsigned long long total)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long kMask = (1ull << 62) - 1ull;
    if (tile == 0u) { if (lane == 0) __atomic_store_n(state, (2ull << 62) | (total & kMask), __ATOMIC_RELAXED); return 0ull; }
    if (lane == 0) __atomic_store_n(state + tile, (1ull << 62) | (total & kMask), __ATOMIC_RELAXED);
    unsigned long long excl = 0;
    for (long long base = (long long)tile - 1; ; base -= 64) {
        const long long t = base - (long long)lane;
        unsigned long long s = 2ull << 62;   // (in front of tile 0: prefix 0, known)
        // (relaxed: the word carries everything that is read -- an acquire load would invalidate the caches of every wave on the CU at each turn of the wait)
        if (t >= 0) while (((s = __atomic_load_n(state + t, __ATOMIC_RELAXED)) >> 62) == 0ull) __builtin_amdgcn_s_sleep(1);
        const unsigned long long known = __ballot((s >> 62) == 2ull);
        const uint32_t first = known ? (uint32_t)__ffsll((long long)known) - 1u : 64u;
        excl += wave_sum_u64(lane <= first ? (s & kMask) : 0ull);
        if (known) break;
    }
    if (lane == 0) __atomic_store_n(state + tile, (2ull << 62) | ((excl + total) & kMask), __ATOMIC_RELAXED);
    return excl;
}

} // namespace ommx
