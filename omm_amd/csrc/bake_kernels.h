// bake_kernels.h -- host-callable launch wrappers of the HIP kernels (bake_kernels.hip, tail_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bake_types.h"

namespace ommx {

struct SetupCounters;

// classification of the active items of ALL levels: level l = activeIds[first[l] .. first[l] + count[l]).  Items of level >= 5 are cut into
// tiles; `queue` holds classify_queue_records(count) tile records of kTileRecordBytes, `queueCtl` kClassifyCtlWords words (zeroed here).  numCUs sizes the persistent grid.
constexpr size_t kTileRecordBytes = 48;
uint64_t classify_queue_records(const uint32_t count[kNumLevels], bool sections = false);   // sections: a streamed bake (chunks.count > 1), whose queue holds a second copy of the levels >= 6
// `queueCtl` (zeroed here): per section k of the 4096-tile queue the words [kSecTails + k] records appended, [kSecHeads + k] records handed out,
// [kSecBases + k] first record, [kSecDone + k] records whose tiles are finished AND visible device-wide (== tail: the section is complete); the same four
// words of the 1024-tile queue (one section) follow at kCtl1024.
// `chunks` (optional, streamed bakes): the work items of the levels >= 6 are cut into `count` ranges of about equal tile counts -- in the order of the final
// result: highest level first, then the position in that level's active list -- and every range gets a section of the queue; ONE persistent launch drains the
// sections in order.  after(user, k, segments, n, false) is called once per range (after that launch is enqueued) with the range as segments of the active
// lists: its work must wait for section k (launch_stream_wait_section).  after(user, count, ..., true) follows with the lower levels (which come last in the
// result).  mark(user) is called right before the persistent launch of the levels >= 6.
constexpr uint32_t kMaxClassifyChunks = 64;
constexpr uint32_t kSecTails = 0, kSecHeads = kMaxClassifyChunks, kSecBases = 2 * kMaxClassifyChunks, kSecDone = 3 * kMaxClassifyChunks, kCtl1024 = 4 * kMaxClassifyChunks;
constexpr uint32_t kClassifyCtlWords = 8 * kMaxClassifyChunks;
struct ClassifySegment { uint32_t level, first, count; };   // activeIds[first .. first + count), all of one level
struct ClassifyChunks {
    uint32_t count;
    void (*after)(void* user, uint32_t chunk, const ClassifySegment* segs, uint32_t numSegs, bool last);
    void (*mark)(void* user);
    void* user;
    const uint8_t* early;   // or null.  Per item: 1 = classify it in a launch of its own class BEFORE the first range (streamed bakes: possible duplicates)
    void (*afterEarly)(void* user, const ClassifySegment* segs, uint32_t numSegs);   // behind that launch: one segment per level >= 6 (all its items)
};
// the whole-item kernel at `level` over a plain item list (used for the level-2 preview of a streamed bake)
hipError_t launch_classify_items(const ClassifyParams& P, const ItemArrays& A, const uint32_t* ids, uint32_t count, uint32_t level, hipStream_t stream);
hipError_t launch_classify(const ClassifyParams& P, const ItemArrays& A, const uint32_t* activeIds, const uint32_t first[kNumLevels], const uint32_t count[kNumLevels],
                           void* queue, uint32_t* queueCtl, uint32_t numCUs, hipStream_t stream, const ClassifyChunks* chunks = nullptr);
// level-0 hierarchical query per work item: uniform items get stateMask = 1 << state and active = 0
void launch_triage(const ClassifyParams& P, const float* uv, const SetupCounters* counters, uint32_t maxItems, uint32_t* stateMask, uint8_t* active, hipStream_t stream);
// XXH64(seed 42) of the 3-state byte stream of each listed item -> digests[item]
// (only != null: the listed items with (only[item] != 0) == (want != 0); liveCount != null: a device word with the number of listed items, <= numItems)
void launch_digest(const uint8_t* states, const uint64_t* stateOfs, const uint32_t* itemIds, uint32_t numItems, uint32_t level, uint32_t bits,
                   uint64_t* digests, hipStream_t stream, const uint8_t* only = nullptr, int want = 0, const uint32_t* liveCount = nullptr);
// summed-area table of (alpha > cutoff)
// (scratch: sat_scratch_bytes(w, h) bytes of device memory, free again once the stream has passed the build)
size_t sat_scratch_bytes(int w, int h);
void launch_sat_build(const void* texels, int fp32, uint32_t* sat, uint32_t* scratch, int w, int h, float cutoff, hipStream_t stream);
// active[item] == 0: the item has no stored states (settled by triage); its block is the constant pattern of its single state
void launch_gather_omms(const uint8_t* states, const uint64_t* stateOfs, const uint8_t* active, const uint32_t* stateMask, const uint8_t* level, int bits,
                        const uint32_t* order, const uint32_t* dstOfs, const uint32_t* sizes, uint32_t numOmms, uint8_t* arrayData, hipStream_t stream);
void launch_write_indices(const int32_t* triToItem, const uint32_t* rep, const int32_t* itemValue, uint32_t numTris, int32_t unresolved,
                          int32_t* out, hipStream_t stream);

// ---- device work-item setup (setup_kernels.hip) ----
struct SetupParams {
    const void* texCoords; uint32_t stride; int uvFormat;     // ommTexCoordFormat, stride in bytes (already defaulted)
    const void* indices; int indexFormat; uint32_t numTris;   // ommIndexFormat
    const uint8_t* perTriLevels;                               // or null
    int globalLevel; float dynScale; int edgeHeuristic;
    int texW, texH; int disableDedup; int wantWorkload;
    uint64_t keyMask;   // all ones; tests narrow it (ommxBakerKnob_SetupKeyBits) to force key collisions and exercise the exact host redo
};
struct SetupCounters {                                         // one device-resident block, read back in a single copy
    uint32_t numItems, numDisabled, numPending, collision;
    uint64_t workload;
    uint32_t levelCount[kNumLevels];
    uint32_t levelStart[kNumLevels + 1];
    uint32_t activeStart[kNumLevels + 1];                      // filled by run_prep
    uint32_t pad_;
    uint64_t stateBytes;                                       // filled by run_prep
};
size_t setup_scratch_bytes(uint32_t numTris);
hipError_t run_setup_fetch(const SetupParams& S, void* scratch, size_t scratchBytes, SetupCounters* counters, hipStream_t stream);
hipError_t copy_pending_to_host(void* scratch, size_t scratchBytes, uint32_t numTris, uint32_t numPending, uint32_t* pendingTris, float* pendingUv, hipStream_t stream);
hipError_t run_setup_fix_pending(const SetupParams& S, void* scratch, size_t scratchBytes, const uint32_t* pendingTris, const uint8_t* levels, uint32_t numPending, hipStream_t stream);
hipError_t run_setup_items(const SetupParams& S, void* scratch, size_t scratchBytes, SetupCounters* counters, float* itemUv, uint8_t* itemLevel,
                           uint8_t* itemDegenerate, int32_t* triToItem, uint32_t* itemIds, float* triArea /* per triangle UV area, or null */, hipStream_t stream);

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
void launch_shard_scatter(const uint8_t* gathered, uint64_t rankPitch, uint64_t lo, uint64_t hi, const uint8_t* active, const uint8_t* owner, const uint32_t* stateMask,
                          const uint8_t* level, int bits, const uint32_t* order, const uint64_t* cofs, const uint32_t* dstOfs, const uint32_t* sizes,
                          uint32_t numOmms, uint8_t* arrayData, hipStream_t stream);

// ---- streamed result of ommCpuBake (tail_kernels.hip: "Streamed result") ----
struct StreamSegment {
    const uint32_t* ids; uint32_t count, level, range;        // items activeIds[..] of ONE level, consecutive in the sorted list; range = index of the classification launch
    const uint32_t* stateMask; const uint32_t* knownCount; const uint64_t* digests; const uint8_t* states; const uint64_t* stateOfs;
    float rejectionThreshold; int bits, disableDedup;
    const uint8_t* early;   // or null: per item, 1 = classified (and its digest entered into the table) before the first range
    const uint32_t* liveCount;   // or null: device word with the number of ids in use (<= count; the early lists are filled on the device)
};
// the early items of a segment: their digests enter the table before anything is placed (run after their classification + launch_digest(.., early, 1))
void launch_stream_insert_early(const StreamSegment& g, uint32_t numActive, void* scratch, size_t scratchBytes, hipStream_t stream);
size_t stream_scratch_bytes(uint32_t numActive);
hipError_t run_stream_begin(uint32_t* activeIds, uint32_t numActive, const float* uv, const uint8_t* level, void* scratch, size_t scratchBytes, hipStream_t stream);
// placed[item] <- final arrayData offset of the item's block (~0: no block); *cursor advances by the bytes placed; ctl: 3 words {blocks placed, violation, mismatch}
hipError_t run_stream_segment(const StreamSegment& g, uint32_t numActive, void* scratch, size_t scratchBytes, unsigned long long* cursor, uint8_t* stage,
                              uint64_t* placed, uint32_t* ctl, hipStream_t stream);
void launch_stream_publish(const unsigned long long* cursor, unsigned long long* hostSlot, hipStream_t stream);
// `stream` does not pass until section `section` of the 4096-tile queue is complete (every block of that range classified and visible); the stream must
// already be ordered behind the tile triage.  ctl: the streamed result's control words (a wait that gives up sets the violation word)
void launch_stream_wait_section(const uint32_t* queueCtl, uint32_t section, uint32_t* ctl, hipStream_t stream);
// preview of the items of level >= 6 (tail_kernels.hip "preview"): prepare -> launch_classify_items(kPreviewLevel, preview buffers) -> flags
constexpr uint32_t kPreviewLevel = 5, kPreviewSlotBytes = 256;   // 1024 micro-triangles x 2 bits
void launch_stream_preview_prepare(const uint32_t* ids, uint32_t n, const float* uv, float texW, float texH, float* uv2, uint64_t* ofs2, uint8_t* early, hipStream_t stream);
// ctl: kStreamCtlWords words {blocks placed, violation, mismatch, early items, early items per level [kStreamCtlEarly + level]}
constexpr uint32_t kStreamCtlEarly = 4, kStreamCtlWords = 4 + 16;
// early[item] <- 1 for the early class; the class also as per-level lists earlyList[levelStart[level] + k], k < ctl[kStreamCtlEarly + level]
hipError_t run_stream_preview_flags(const uint32_t* ids, uint32_t n, uint32_t numActive, const uint8_t* states2, const uint8_t* level, uint8_t* early, uint32_t* ctl,
                                    void* scratch, size_t scratchBytes, uint32_t* earlyList, const uint32_t levelStart[kNumLevels], hipStream_t stream);
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
};
struct TailCounts { uint32_t numOmms; uint64_t arrayDataSize; };

// compaction of the active (non-uniform) items into per-level lists + their packed-state slots; synchronises the stream once
// (item count and level boundaries are read from / written to the device-resident counters block; no synchronisation)
hipError_t run_prep(const uint32_t* itemIds, const uint8_t* active, const uint8_t* level, int bits, uint32_t maxItems, SetupCounters* counters,
                    uint32_t* activeIds, uint64_t* stateOfs, void* scratch, size_t scratchBytes, hipStream_t stream);
void launch_narrow_indices(const int32_t* in, uint32_t n, int bytesPerIndex, void* out, hipStream_t stream);
// scratch handling: call with scratch == nullptr to get the size
size_t tail_scratch_bytes(uint32_t numItems, uint32_t numTris);
// runs the device tail up to (and including) offsets; returns counts (synchronises the stream once)
hipError_t run_tail(const TailInputs& in, const TailOutputs& out, void* scratch, size_t scratchBytes, TailCounts* counts, hipStream_t stream);
void launch_write_descs(const uint32_t* order, const uint32_t* dstOfs, const uint8_t* level, int format, uint32_t numOmms, void* descArray, hipStream_t stream);

} // namespace ommx
