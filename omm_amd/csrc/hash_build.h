// Wave-cooperative hash build shared by the two "first occurrence wins" steps of the bake: UV-triangle dedup (SetupWorkItems,
// bake_cpu_impl.cpp:589-660) and exact OMM-block dedup by XXH64 digest (DeduplicateExact, bake_cpu_impl.cpp:1031-1066).
//
// Open addressing over 64-bit keys, linear probing, slot value = smallest index seen (atomicMin); a later look-up of the same key
// returns that index.  Empty = all ones, which is also the initial value of the values: one memset fills both arrays.  A key that IS
// all ones lives in a dedicated slot past the table (values[slots]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ommx {

constexpr uint64_t kEmptyKey = ~0ull;

struct HashTable {
    unsigned long long* keys; uint32_t* vals;   // slots keys, slots + 1 values
    uint32_t mask, shift;                       // slots - 1, 64 - log2(slots)
};
// smallest power of two >= 2 * n (load factor <= 0.5), at least 1024
inline uint32_t hash_table_slots(uint64_t n) { uint32_t s = 1024; while ((uint64_t)s < 2 * n && s < 0x80000000u) s <<= 1; return s; }
inline HashTable hash_table_at(void* base, uint32_t slots)   // layout: keys | values (slots + 1); see hash_table_bytes()
{
    HashTable t; t.keys = (unsigned long long*)base; t.vals = (uint32_t*)((uint8_t*)base + (size_t)slots * 8); t.mask = slots - 1u;
    uint32_t lg = 0; while ((1u << lg) < slots) ++lg;
    t.shift = 64u - lg;
    return t;
}
inline size_t hash_table_bytes(uint32_t slots, uint32_t extraWords = 0) { return (size_t)slots * 8 + ((size_t)slots + 1 + extraWords) * 4; }

// Fibonacci hashing: the keys of the UV dedup can be small integers (triangles that must not merge), the digests are already mixed
__device__ __forceinline__ uint32_t hash_slot_of(const HashTable& t, uint64_t k) { return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> t.shift); }

// probe for `k` (claiming an empty slot if it is not in the table yet) and lower the slot's value to `i`.  Plain (L2-coherent) loads
// first: a slot's value only ever decreases, so a value <= i read here -- however stale -- proves that i is not the first occurrence and
// no atomic is needed (same-address atomics cost ~9 ns each on this chip).
__device__ __forceinline__ void hash_put_min(const HashTable& t, uint64_t k, uint32_t i)
{
    uint32_t slot = k == kEmptyKey ? t.mask + 1u : hash_slot_of(t, k);
    bool found = k == kEmptyKey;
    if (k != kEmptyKey)
        for (uint32_t probes = 0; probes <= t.mask; ++probes) {   // (the table is at most half full: the bound only guards against a hang)
            unsigned long long cur = __atomic_load_n(t.keys + slot, __ATOMIC_RELAXED);
            if (cur == kEmptyKey) cur = atomicCAS(t.keys + slot, (unsigned long long)kEmptyKey, (unsigned long long)k);
            if (cur == kEmptyKey || cur == k) { found = true; break; }
            slot = (slot + 1u) & t.mask;
        }
    // a full table (only reachable on the bake's error path) leaves the key out: hash_get() then answers `self`, never another key's slot
    if (found && __atomic_load_n(t.vals + slot, __ATOMIC_RELAXED) > i) atomicMin(t.vals + slot, i);
}

// One call per workgroup (every thread, uniformly): all lanes put in parallel, except lanes that can see a lower lane of their own wave
// with the same key.  They find it through 64 LDS buckets per wave -- each bucket keeps the lowest lane that hashed to it -- so a wave
// full of copies of one key costs the table one operation instead of 64 same-address atomics, while a wave of 64 different keys (the
// common case) stays fully parallel.  A copy whose bucket was taken by a lower lane with another key just puts as well: redundant, never
// wrong.  Indices must ascend with the lane.  `buckets`: 64 words of LDS per wave of the workgroup.
__device__ __forceinline__ void hash_put_min_block(const HashTable& t, bool live, uint64_t k, uint32_t i, uint32_t* buckets)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* mine = buckets + (threadIdx.x >> 6) * 64u;
    const uint32_t b = (uint32_t)((k * 0xD6E8FEB86659FD93ull) >> 58);
    mine[lane] = 0xFFFFFFFFu;
    __syncthreads();
    if (live) atomicMin(&mine[b], lane);
    __syncthreads();
    const uint32_t w = live ? mine[b] : lane;
    const uint64_t kw = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(k >> 32), (int)w) << 32) | (uint32_t)__shfl((int)(uint32_t)k, (int)w);
    if (live && !(w < lane && kw == k)) hash_put_min(t, k, i);
}

// index of the slot that holds `k`, or 0xFFFFFFFF if the key was never put
__device__ __forceinline__ uint32_t hash_find_slot(const HashTable& t, uint64_t k)
{
    if (k == kEmptyKey) return t.mask + 1u;
    uint32_t slot = hash_slot_of(t, k), probes = 0;
    while (t.keys[slot] != k) { if (t.keys[slot] == kEmptyKey || ++probes > t.mask) return 0xFFFFFFFFu; slot = (slot + 1u) & t.mask; }
    return slot;
}

// value of the slot that holds `k`; `self` if the key was never put (cannot happen in the two users: the bound only guards against a hang)
__device__ __forceinline__ uint32_t hash_get(const HashTable& t, uint64_t k, uint32_t self)
{
    uint32_t slot = t.mask + 1u;
    if (k != kEmptyKey) {
        slot = hash_slot_of(t, k);
        uint32_t probes = 0;
        while (t.keys[slot] != k) { if (t.keys[slot] == kEmptyKey || ++probes > t.mask) return self; slot = (slot + 1u) & t.mask; }
    }
    const uint32_t v = t.vals[slot];
    return v == 0xFFFFFFFFu ? self : v;
}

} // namespace ommx
