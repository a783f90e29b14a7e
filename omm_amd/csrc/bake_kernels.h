// bake_kernels.h -- host-callable launch wrappers of the HIP kernels (bake_kernels.hip, tail_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bake_types.h"

namespace ommx {

struct SetupCounters;

// classification of the active items of ALL levels: level l = activeIds[first[l] .. first[l] + count[l]).  Items of level >= 5 are cut into
// tiles; `queue` holds classify_queue_records(count) tile records of kTileRecordBytes, `queueCtl` kClassifyCtlWords words (zeroed here).  numCUs sizes the persistent grid.
constexpr size_t kTileRecordBytes = 112;   // 48 bytes of tile data + 64 verdict bytes (one per 64-group: triage_groups)
uint64_t classify_queue_records(const uint32_t count[kNumLevels], bool sections = false);   // sections: a streamed bake (chunks.count > 1), whose queue holds a second copy of the levels >= 6
// `queueCtl` (zeroed here): per section k of the 4096-tile queue the words [kSecTails + k] records appended, [kSecHeads + k] records handed out,
// [kSecBases + k] first record, [kSecDone + k] records whose tiles are finished AND visible device-wide (== tail: the section is complete); the same four
// words of the 1024-tile queue (one section) follow at kCtl1024.
// `chunks` (optional, streamed bakes): the work items of the levels >= 6 are cut into `count` ranges of about equal tile counts -- in the order of the final
// result: highest level first, then the position in that level's active list -- and every range gets a PAIR of queue sections: 2k for its own items, 2k + 1 for
// the early items of later ranges whose family starts in it; ONE persistent launch drains the sections in order.  after(user, k, segments, n, false) is called
// once per range (after that launch is enqueued) with the range as segments of the active lists: its work must wait for sections 2k and 2k + 1
// (launch_stream_wait_sections).  after(user, count, ..., true) follows with the lower levels (which come last in the
// result).  mark(user) is called right before the persistent launch of the levels >= 6.
constexpr uint32_t kMaxClassifyChunks = 64;   // sections of the 4096-tile queue; a streamed bake uses two per range (kMaxStreamRanges ranges at most)
constexpr uint32_t kMaxStreamRanges = kMaxClassifyChunks / 2;
constexpr uint32_t kSecTails = 0, kSecHeads = kMaxClassifyChunks, kSecBases = 2 * kMaxClassifyChunks, kSecDone = 3 * kMaxClassifyChunks, kCtl1024 = 4 * kMaxClassifyChunks;
// (free words of the 1024-tile block: records in the staging list of the early items' tiles, per-range fill counts of their scatter)
constexpr uint32_t kCtlEarlyStaged = kCtl1024 + 1, kCtlEarlyFill = kCtl1024 + 2;
constexpr uint32_t kClassifyCtlWords = 8 * kMaxClassifyChunks;
struct ClassifySegment { uint32_t level, first, count; };   // activeIds[first .. first + count), all of one level
// the sliced levels of one tile size, highest level first: items activeIds[first[k] ..), tiles [tileStart[k], tileStart[k + 1]) of the tile enumeration
struct TileLevels { uint32_t n; uint32_t level[kNumLevels], first[kNumLevels], tileStart[kNumLevels + 1]; };
// ranges of the 4096-tile enumeration (= of the final order of the result), cut at work-item boundaries: range k = tiles [cut[k], cut[k + 1])
struct TileSections { uint32_t n; uint32_t cut[kMaxClassifyChunks + 1]; };
struct ClassifyPlan { TileLevels big, small; uint64_t totalBig, totalSmall; TileSections ranges; };
// range of the work item at position `pos` of the active list
__device__ __forceinline__ uint32_t section_of_position(uint32_t pos, const TileLevels& L, const TileSections& S)
{
    uint32_t t = 0;
    for (uint32_t g = 0; g < L.n; ++g) {
        const uint32_t shift = 2u * (L.level[g] - 6u), cnt = (L.tileStart[g + 1] - L.tileStart[g]) >> shift;
        if (pos >= L.first[g] && pos - L.first[g] < cnt) t = L.tileStart[g] + ((pos - L.first[g]) << shift);
    }
    uint32_t sec = 0;
    while (sec + 1u < S.n && t >= S.cut[sec + 1u]) ++sec;
    return sec;
}

// the plan launch_classify follows for these lists (pure host arithmetic; a streamed bake needs the ranges before the launch: section_of_position)
void classify_plan(const uint32_t first[kNumLevels], const uint32_t count[kNumLevels], uint32_t ranges, ClassifyPlan* plan);
// The range an early item is classified with: that of the first member of its family -- but never later than its own.  (The first member of a real family
// comes first by construction; the clamp makes "every block of a range is complete when the range's sections are" hold whatever the preview's
// signatures did -- a hash collision then costs a fallback through the duplicate check, never an unfinished block.)
__device__ __forceinline__ uint32_t early_range(uint32_t leadPos, uint32_t ownPos, const TileLevels& L, const TileSections& S)
{
    const uint32_t a = section_of_position(leadPos, L, S), b = section_of_position(ownPos, L, S);
    return a < b ? a : b;
}
// Deferred generic pass (asset-sized triangles: micro-triangles of several texels).  The persistent launch queues the micro-triangles that need the
// generic texel loops instead of walking them itself -- entry = {item | degenerate << 30, level << 24 | micro-triangle index}, their packed state left 0 --
// and classify_generic() classifies them afterwards -- its lanes take the entries one after the other as their walks end -- and ORs the states in.  count[1] is that launch's cursor, count[2] the number of micro-triangles it classified (null entries not counted).  *count only grows and may exceed capacity: a tile
// whose reservation does not fit walks its micro-triangles itself and fills the part of the reservation that lies inside the queue with null entries (x == ~0).
struct GenericQueue { uint2* entries; unsigned long long* count; uint32_t capacity; };
struct ClassifyChunks {
    uint32_t count;
    void (*after)(void* user, uint32_t chunk, const ClassifySegment* segs, uint32_t numSegs, bool last);
    void (*mark)(void* user);
    void (*markGeneric)(void* user);   // or null: called right before the deferred generic pass is enqueued
    void* user;
    // Streamed bakes, or null: early[item] == 1 marks a possible duplicate (tail_kernels.hip "preview"); it is classified in the range of the FIRST member of
    // its family, earlyLead[item] = that member's position in activeIds -- so a range's section pair holds everything the placement of the range depends on.
    const uint8_t* early; const uint32_t* earlyLead;
    void* earlyStage;       // >= kTileRecordBytes x (open tiles of early items) bytes of scratch, free until the persistent launch starts
    GenericQueue generic;   // entries != null: deferred generic pass (not together with `after`: a streamed range must be complete when its sections are)
};
hipError_t launch_classify(const ClassifyParams& P, const ItemArrays& A, const uint32_t* activeIds, const uint32_t first[kNumLevels], const uint32_t count[kNumLevels],
                           void* queue, uint32_t* queueCtl, uint32_t numCUs, hipStream_t stream, const ClassifyChunks* chunks = nullptr);
// level-0 hierarchical query per work item (summed-area table, then the curve-free-region test): uniform items get stateMask = 1 << state and active = 0
void launch_triage(const ClassifyParams& P, const float* uv, const uint8_t* level, const uint8_t* degenerate, const SetupCounters* counters, uint32_t maxItems,
                   uint32_t* stateMask, uint8_t* active, void* prepScratch /* run_prep's scratch block: its first prep_state_words() words are zeroed */, hipStream_t stream);
// XXH64(seed 42) of the 3-state byte stream of each listed item -> digests[item]
// (only != null: the listed items with (only[item] != 0) == (want != 0))
void launch_digest(const uint8_t* states, const uint64_t* stateOfs, const uint32_t* itemIds, uint32_t numItems, uint32_t level, uint32_t bits,
                   uint64_t* digests, hipStream_t stream, const uint8_t* only = nullptr, int want = 0);
// ... of the active items of all levels at once (level l: activeIds[first[l] .. first[l] + count[l])): the few-long-items levels share one launch
void launch_digest_levels(const uint8_t* states, const uint64_t* stateOfs, const uint32_t* activeIds, const uint32_t first[kNumLevels], const uint32_t count[kNumLevels],
                          uint32_t bits, uint64_t* digests, hipStream_t stream);
// streamed bakes, levels >= 6: two lists in one launch of small workgroups (runs next to the persistent classification launch)
struct DigestLists {
    // list A: ids[0 .. count) of ONE level, optionally only the items with (only[item] != 0) == (want != 0)
    const uint32_t* ids; uint32_t count, level; const uint8_t* only; int want;
    // list B: listB[*liveStart .. + *liveCount) (device words), at most capacityB entries, every item with its own level itemLevel[item] >= 6
    const uint32_t* listB; uint32_t capacityB; const uint32_t* liveStart; const uint32_t* liveCount; const uint8_t* itemLevel;
};
void launch_digest_lists(const uint8_t* states, const uint64_t* stateOfs, const DigestLists& lists, uint32_t bits, uint64_t* digests, hipStream_t stream);
// summed-area table of (alpha > cutoff)
// (scratch: sat_scratch_bytes(w, h) bytes of device memory, free again once the stream has passed the build)
size_t sat_scratch_bytes(int w, int h);
void launch_sat_build(const void* texels, int fp32, uint32_t* sat, uint32_t* scratch, int w, int h, float cutoff, hipStream_t stream);
// active[item] == 0: the item has no stored states (settled by triage); its block is the constant pattern of its single state
// storeBits: packing of `states` (== bits, except a 2-state bake without fine pass, whose states are kept in 2 bits: the gather then packs them to 1 bit by
// the reference's rule, byte[i >> 3] |= state << (i & 7) truncated to the byte, bake_cpu_impl.cpp:1811)
void launch_gather_omms(const uint8_t* states, const uint64_t* stateOfs, const uint8_t* active, const uint32_t* stateMask, const uint8_t* level, int bits, int storeBits,
                        const uint32_t* order, const uint32_t* dstOfs, const uint32_t* sizes, uint32_t numOmms, uint8_t* arrayData, hipStream_t stream,
                        uint8_t* unitCodes = nullptr, uint32_t* blockRawCounts = nullptr,
                        void* descs = nullptr /* ommCpuOpacityMicromapDesc[numOmms]: written by the gather when given */);
void launch_write_indices(const int32_t* triToItem, const uint32_t* rep, const int32_t* itemValue, uint32_t numTris, int32_t unresolved,
                          int32_t* out, hipStream_t stream);

// ---- device work-item setup (setup_kernels.hip) ----
struct SetupParams {
    const void* texCoords; uint32_t stride; int uvFormat;     // ommTexCoordFormat, stride in bytes (already defaulted)
    const void* indices; int indexFormat; uint32_t numTris;   // ommIndexFormat
    const uint8_t* perTriLevels;                               // or null
    int globalLevel; float dynScale; int edgeHeuristic;
    int texW, texH; int disableDedup; int wantWorkload;
    int degenerateInvalid;   // internal flag DisableLevelLineIntersection: degenerate triangles count as invalid (bake_cpu_impl.cpp:569-572)
    int format;              // ommFormat of the bake: part of the work-item id (vm_id.h)
};
struct SetupCounters {                                         // one device-resident block, read back in a single copy
    uint32_t numItems, numDisabled, numPending, reserved_;
    uint64_t workload;
    uint32_t levelCount[kNumLevels];
    uint32_t levelStart[kNumLevels + 1];
    uint32_t activeStart[kNumLevels + 1];                      // filled by run_prep
    uint32_t pad_;
    uint64_t stateBytes;                                       // filled by run_prep
};
size_t setup_scratch_bytes(uint32_t numTris);
// (also zeroes two word ranges of the caller -- the bake's per-item known counts and its statistic counters -- and fills the UV-dedup table run_setup_items uses)
hipError_t run_setup_fetch(const SetupParams& S, void* scratch, size_t scratchBytes, SetupCounters* counters, uint32_t* zero, uint32_t zeroWords, uint32_t* zero2, uint32_t zero2Words,
                           float* triArea /* per triangle UV area, or null */, hipStream_t stream);
hipError_t copy_pending_to_host(void* scratch, size_t scratchBytes, uint32_t numTris, uint32_t numPending, uint32_t* pendingTris, float* pendingUv, void* tmp, hipStream_t stream);
hipError_t run_setup_fix_pending(const SetupParams& S, void* scratch, size_t scratchBytes, const uint32_t* pendingTris, const uint8_t* levels, uint32_t numPending, void* tmp, hipStream_t stream);
hipError_t run_setup_items(const SetupParams& S, void* scratch, size_t scratchBytes, SetupCounters* counters, float* itemUv, uint8_t* itemLevel,
                           uint8_t* itemDegenerate, int32_t* triToItem, uint32_t* itemIds, hipStream_t stream);

// ---- multi-GPU sharding helpers (tail_kernels.hip) ----
constexpr int kMaxRanks = 16;
struct ShardBounds { uint32_t rank, world; uint32_t b[kNumLevels][kMaxRanks + 1]; }; // b[l][r]: first active-list position of rank r at level l
void launch_shard_interleave(const uint32_t* in, uint32_t* out, uint32_t count, uint32_t stride, hipStream_t stream);
void launch_shard_pack_meta(const ShardBounds& B, const uint32_t* activeIds, uint32_t numActive, const uint32_t* mask, const uint32_t* known,
                            const uint64_t* digests, uint32_t* meta, hipStream_t stream);
void launch_shard_unpack_meta(const ShardBounds& B, const uint32_t* activeIds, uint32_t numActive, const uint32_t* meta, uint32_t* mask, uint32_t* known,
                              uint64_t* digests, uint8_t* owner, hipStream_t stream);
hipError_t run_shard_layout(const uint32_t* order, const uint32_t* sizes, const uint8_t* active, const uint8_t* owner, uint32_t numOmms, uint32_t world,
                            uint64_t* cofs, uint64_t* totalsDev, uint64_t* totalsHost, void* scratch, size_t scratchBytes, hipStream_t stream);
void launch_shard_gather(const uint8_t* states, const uint64_t* stateOfs, const uint8_t* active, const uint8_t* owner, uint32_t rank, const uint32_t* order,
                         const uint64_t* cofs, const uint32_t* sizes, uint32_t numOmms, uint8_t* contrib, hipStream_t stream);
// gathered = bytes [lo, hi) of every rank's contribution, rank r at gathered + r * rankPitch (one call per all-gather chunk)
// block exchange codec (tail_kernels.hip): a rank's contribution (a multiple of 256 bytes) as a stream of unit codes + raw units
constexpr uint32_t kCodecIncompressible = 0xFFFFFFFEu;
size_t shard_codec_scratch_bytes(uint64_t contributionBytes);
// The same stream from codes and per-block raw counts that the gather has produced next to the array (launch_gather_omms with unitCodes / blockRawCounts: one
// byte per 16-byte unit, one zeroed word per 256-unit block + 1): the array is read once more for its raw units only, not twice in full.
hipError_t run_shard_compress_coded(const uint8_t* contrib, uint64_t contributionBytes, const uint8_t* unitCodes, uint32_t* blockRawCounts, uint8_t* comp, uint64_t capBytes,
                                    uint32_t* sizeWord, void* scratch, size_t scratchBytes, hipStream_t stream);
// the code of one 16-byte unit of arrayData: 0..3 = sixteen bytes of 0x00 / 0x55 / 0xAA / 0xFF, 4 = raw
__host__ __device__ inline uint32_t codec_unit_code(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    const bool same = x == y && x == z && x == w;
    return !same ? 4u : (x == 0u ? 0u : (x == 0x55555555u ? 1u : (x == 0xAAAAAAAAu ? 2u : (x == 0xFFFFFFFFu ? 3u : 4u))));
}
constexpr uint32_t kCodecBlockUnits = 256;
hipError_t run_shard_compress(const uint8_t* contrib, uint64_t contributionBytes, uint8_t* comp, uint64_t capBytes, uint32_t* sizeWord, void* scratch, size_t scratchBytes, hipStream_t stream);
void launch_shard_scatter_streams(const uint8_t* streams, uint64_t streamPitch, uint64_t contributionBytes, const uint8_t* active, const uint8_t* owner,
                                  const uint32_t* stateMask, const uint8_t* level, int bits, const uint32_t* order, const uint64_t* cofs, const uint32_t* dstOfs,
                                  const uint32_t* sizes, uint32_t numOmms, uint8_t* arrayData, hipStream_t stream);
void launch_shard_scatter(const uint8_t* gathered, uint64_t rankPitch, uint64_t lo, uint64_t hi, const uint8_t* active, const uint8_t* owner, const uint32_t* stateMask,
                          const uint8_t* level, int bits, const uint32_t* order, const uint64_t* cofs, const uint32_t* dstOfs, const uint32_t* sizes,
                          uint32_t numOmms, uint8_t* arrayData, hipStream_t stream);

// ---- streamed result of ommCpuBake (tail_kernels.hip: "Streamed result") ----
struct StreamSegment {
    const uint32_t* ids; uint32_t count, level, range;        // items activeIds[..] of ONE level, consecutive in the sorted list; range = index of the classification launch
    const uint32_t* stateMask; const uint32_t* knownCount; const uint64_t* digests; const uint8_t* states; const uint64_t* stateOfs;
    float rejectionThreshold; int bits, disableDedup;
    const uint8_t* early;   // or null: per item, 1 = classified (and its digest entered into the table) before the first range
    const uint32_t* liveCount; const uint32_t* liveStart;   // or null: the ids in use are the device-side slice ids[*liveStart .. + *liveCount) (count = capacity)
    const uint8_t* itemLevel;    // or null: per-item level (a slice of the early list mixes levels; `level` is ignored then)
};
// a slice of the early list (g.liveStart / g.liveCount / g.itemLevel): digests into the table (run after launch_digest_list on the same slice)
void launch_stream_insert_list(const StreamSegment& g, uint32_t numActive, void* scratch, size_t scratchBytes, hipStream_t stream);
size_t stream_scratch_bytes(uint32_t numActive);
hipError_t run_stream_begin(uint32_t* activeIds, uint32_t numActive, const float* uv, const uint8_t* level, void* scratch, size_t scratchBytes, hipStream_t stream);
// placed[item] <- final arrayData offset of the item's block (~0: no block); *cursor advances by the bytes placed; ctl: 3 words {blocks placed, violation, mismatch}
hipError_t run_stream_segment(const StreamSegment& g, uint32_t numActive, void* scratch, size_t scratchBytes, unsigned long long* cursor, uint8_t* stage,
                              uint64_t* placed, uint32_t* ctl, hipStream_t stream);
void launch_stream_publish(const unsigned long long* cursor, unsigned long long* hostSlot, hipStream_t stream);
// `stream` does not pass until sections [first, first + n) of the 4096-tile queue are complete (every block in them classified and visible); the stream
// must already be ordered behind the tile triage.  ctl: the streamed result's control words (a wait that gives up sets the violation word)
// timeoutSeconds: give up after that long (at least 4 s; the caller scales it with the estimated duration of the classification: a contended GPU or a profiler
// must not turn a long first range into a discarded stream)
void launch_stream_wait_sections(const uint32_t* queueCtl, uint32_t first, uint32_t n, uint32_t* ctl, double timeoutSeconds, hipStream_t stream);
// preview of the items of level >= 6 (tail_kernels.hip "preview"): prepare -> launch_classify() of the items as ONE level-5 class into the preview buffers -> flags
constexpr uint32_t kPreviewLevel = 5, kPreviewSlotBytes = 256;   // 1024 micro-triangles x 2 bits
void launch_stream_preview_prepare(const uint32_t* ids, uint32_t n, const float* uv, float texW, float texH, float* uv2, uint64_t* ofs2, uint8_t* early, hipStream_t stream);
// ctl: kStreamCtlWords words {blocks placed, violation, mismatch, early items; per range: early items classified with it, start of their slice of the early list, fill}
constexpr uint32_t kStreamCtlEarlyCount = 4, kStreamCtlEarlyStart = 4 + kMaxStreamRanges, kStreamCtlEarlyFill = 4 + 2 * kMaxStreamRanges, kStreamCtlWords = 4 + 3 * kMaxStreamRanges;
// ids = activeIds + listOffset (the levels >= 6).  early[item] <- 1 for the early class, earlyLead[item] <- position (in activeIds) of the first member of its
// family; the class also as a list ordered by the range of that member: earlyList[ctl[kStreamCtlEarlyStart + k] .. + ctl[kStreamCtlEarlyCount + k])
hipError_t run_stream_preview_flags(const uint32_t* ids, uint32_t n, uint32_t listOffset, uint32_t numActive, const uint8_t* states2, const uint8_t* level, uint8_t* early,
                                    uint32_t* ctl, void* scratch, size_t scratchBytes, uint32_t* earlyLead, uint32_t* earlyList, const ClassifyPlan& plan, hipStream_t stream);
void launch_stream_verify(const uint32_t* order, const uint32_t* dstOfs, uint32_t numOmms, const uint64_t* placed, uint32_t* ctl, hipStream_t stream);

// ---- device tail (tail_kernels.hip) ----
struct TailInputs {
    uint32_t numItems, numTris;
    uint32_t maxDistinctDigests; // upper bound of the number of different digests (0 = unknown: numItems); sizes the dedup hash table
    const float*    uv;         // 6 per item
    const uint8_t*  level;      // per item
    const uint32_t* stateMask;  // per item
    const uint32_t* knownCount; // per item or null
    uint64_t*       digests;    // per item: filled for non-uniform items by launch_digest; uniform ones are filled here
    const uint64_t* uniformDigest; // [13][4] table: XXH64 of 4^level bytes of value s (s = 0,1,3)
    const int32_t*  triToItem;  // per triangle, -1 = unresolved
    int      format;            // global format (bits per micro-triangle)
    int      disableSpecial, disableDedup;
    float    rejectionThreshold;
    int32_t  unresolved;
    uint32_t* errorFlag;        // device word, set when an item was left unclassified (internal consistency check)
};
struct TailOutputs {            // device buffers owned by the caller
    int32_t*  special;          // per item: 0 = none, else special index
    uint32_t* rep;              // per item: first item with the same digest
    uint32_t* order;            // [numOmms] item of descriptor j
    uint32_t* dstOfs;           // [numOmms] byte offset in arrayData
    uint32_t* sizes;            // [numOmms]
    int32_t*  itemValue;        // per item: special index or descriptor slot
    int32_t*  indexBuffer;      // per triangle
    uint32_t* arrayHist;        // [13]
    uint32_t* indexHist;        // [13]
    void*     narrowIndex;      // per triangle, narrowBytes (1 / 2 / 4) each: the index buffer in the result's format, or null
    int       narrowBytes;
};
struct TailCounts { uint32_t numOmms; uint64_t arrayDataSize; uint32_t smallOmms; };   // smallOmms: emitted OMMs of less than 16 bytes (levels 0 - 2; 0 - 3 in 2-state)

// compaction of the active (non-uniform) items into per-level lists + their packed-state slots; synchronises the stream once
// (item count and level boundaries are read from / written to the device-resident counters block; no synchronisation)
// words at the start of the scratch block that launch_triage zeroes for run_prep (ticket + count / byte states of its tiles of 1024 items)
inline uint32_t prep_state_words(uint32_t maxItems) { return 64u + 4u * ((maxItems + 1023u) / 1024u + 1u); }
hipError_t run_prep(const uint32_t* itemIds, const uint8_t* active, const uint8_t* level, int bits, uint32_t maxItems, SetupCounters* counters,
                    uint32_t* activeIds, uint64_t* stateOfs, void* scratch, size_t scratchBytes, hipStream_t stream);
void launch_narrow_indices(const int32_t* in, uint32_t n, int bytesPerIndex, void* out, hipStream_t stream);
// scratch handling: call with scratch == nullptr to get the size
size_t tail_scratch_bytes(uint32_t numItems, uint32_t numTris);
// runs the device tail up to (and including) offsets; returns counts (synchronises the stream once)
// (hostWork: 64 bytes of pinned host memory for the read-back, or null)
hipError_t run_tail(const TailInputs& in, const TailOutputs& out, void* scratch, size_t scratchBytes, TailCounts* counts, hipStream_t stream, void* hostWork = nullptr);
void launch_write_descs(const uint32_t* order, const uint32_t* dstOfs, const uint8_t* level, int format, uint32_t numOmms, void* descArray, hipStream_t stream);

} // namespace ommx
