/*
 * omm_mi355x_ext.h -- MI355X-specific additions to the C ABI (not part of the reference SDK).
 *
 * The reference exposes no timing or device-resident interface: its baker is a host library
 * (libraries/omm-lib/include/omm.h:568-594).  These entry points exist so a harness can (a) read HIP-event
 * timings of the kernels a bake launched on the library's own stream and (b) run a bake whose inputs and
 * outputs stay in HBM.  They use only plain C types.
 */
#ifndef OMM_MI355X_EXT_H
#define OMM_MI355X_EXT_H
#include "omm_mi355x.h"

/* HIP-event timings of the most recent successful bake on this baker (milliseconds). */
typedef struct ommxBakeTimings {
    float    hostSetupMs;      /* work-item setup on the host (0 when the device setup path ran) */
    float    uploadMs;         /* host -> device copies of the work-item tables */
    float    classifyMs;       /* all classify_tiles launches (hierarchical + per-micro-triangle SAT pass, fine level-line pass) */
    float    digestMs;         /* XXH64 digests */
    float    tailMs;           /* promote / dedup / sort / offsets / index buffer */
    float    gatherMs;         /* gather of surviving OMM blocks + descriptors */
    float    downloadMs;       /* device -> host copy of the result arrays */
    float    totalMs;          /* wall clock of the whole call */
    uint64_t microTriangles;   /* sum of 4^level over unique work items */
    uint32_t uniqueItems;
    uint32_t classifyLaunches;
    uint64_t stateBytes;       /* packed state bytes written by classification */
    float    triageMs;         /* level-0 hierarchical query per item + compaction of the active items */
    uint32_t activeItems;      /* items that needed per-micro-triangle classification */
    uint64_t fineMicroTriangles; /* micro-triangles that needed the level-line (fine) pass */
    float    setupMs;          /* device work-item setup: UV fetch, level selection, first-occurrence dedup, level grouping */
    /* ommCpuBake only: the finished OMM blocks are copied to their final place in the host array WHILE the classification runs (ommxBakerKnob_StreamChunks) */
    uint32_t streamChunks;     /* ranges of the classification whose blocks were streamed (0 = the result was copied after the bake) */
    uint64_t streamedBytes;    /* bytes that crossed PCIe that way */
    float    streamTailMs;     /* wall clock from the end of the classification to the last streamed byte on the host (the exposed part of the copy) */
    uint32_t openTiles;        /* tiles (4096 micro-triangles; 1024 for level-5 items) that the tile triage left open, i.e. the units the persistent
                                  classify_tiles launches really process */
    uint64_t openTileMicroTriangles; /* micro-triangles in those tiles */
    uint32_t streamEarlyItems; /* streamed bakes: work items classified with an EARLIER range than their own because another item shares their level-5
                                  preview (possible duplicates: the first member of such a family pulls the others into its range) */
    float    persistentMs;     /* HIP events around the persistent classify_tiles launch of the levels >= 6 alone (classifyMs also holds the tile triage, the
                                  launches of the lower levels and, for streamed bakes, the preview); 0 for streamed bakes */
    float    genericMs;        /* HIP events around the deferred generic pass (classify_generic: micro-triangles of several texels, ommxBakerKnob_GenericPass); 0 when
                                  that work ran inside the persistent launch */
    uint64_t genericMicroTriangles; /* micro-triangles that pass classified */
    uint64_t exchangeBytes;    /* ommxShardedBakeRccl: bytes every rank put on the wire in the block all-gather (the contributions travel as unit codes + raw
                                  units; a contribution that does not shrink below half its size travels as it is) */
    uint64_t contributionBytes; /* ... and the size of a rank's (padded) contribution before that */
    /* (round 4; read them through ommxGetLastBakeTimingsSized) streamed ommCpuBake, wall clock in ms since the bake's device work was first enqueued: */
    float    streamPreviewMs;      /* HIP events around the level-5 preview of the items of level >= 6 (the possible-duplicate search in front of the classification) */
    float    streamFirstCopyMs;    /* the first range's blocks are placed and their copy is issued */
    float    streamLastCopyMs;     /* the last range's copy is issued */
    float    streamRangeReadyMs[32]; /* range k placed (its event seen by the host thread, which then issues its copy) */
    /* (round 5) how arrayData reached the host (ommxResultTransfer_*; ommCpuBake only) and, for the compressed form: */
    uint32_t resultTransfer;
    uint32_t expandThreads;        /* host threads that expanded the codec stream (the caller's included) */
    uint64_t compressedBytes;      /* bytes of the codec stream that crossed PCIe instead of arrayDataSize */
    float    compressMs;           /* wall clock from the end of the bake proper to the stream's size on the host (codec kernels + the small read-backs) */
    float    expandMs;             /* wall clock of the stream's copy and its expansion into arrayData (overlapped slice by slice) */
    uint32_t devices;              /* devices a multi-device ommCpuBake ran on (ommxBakerKnob_Devices); 0 / 1: one */
    /* (round 6) compressed transfer: the helper threads zero an idle result block of the baker WHILE the device bakes; codec blocks that repeat state 0 are then not written again */
    uint64_t prefilledBytes;       /* bytes of the result array that were zeroed ahead (0: no idle block, first bake, another transfer) */
    uint64_t expandSkippedBytes;   /* bytes of arrayData the expansion did not have to write because of that */
} ommxBakeTimings;

/* ommxBakeTimings only ever grows at its END.  ommxGetLastBakeTimingsSized copies min(outBytes, the library's size) bytes and zeros the rest of `out`, so a
 * caller and a library built from different versions of this header stay compatible; *libraryBytes (optional) <- the library's sizeof(ommxBakeTimings).
 * ommxGetLastBakeTimings is the round-3 symbol: it fills the fields up to and including contributionBytes (the struct as it was then) and never writes
 * beyond them, whatever the caller's header says -- the fields from streamPreviewMs on are only available through the sized call. */
OMM_MI355X_API ommResult ommxGetLastBakeTimingsSized(ommBaker baker, void* out, size_t outBytes, size_t* libraryBytes);
/* DEPRECATED (kept for binaries built against the round-3 header): fills only the round-3 prefix of the struct and leaves the rest of `out` untouched --
 * zero the struct first, or better call ommxGetLastBakeTimingsSized(baker, &t, sizeof t, NULL). */
OMM_MI355X_API ommResult ommxGetLastBakeTimings(ommBaker baker, ommxBakeTimings* out)
#if defined(__GNUC__)
    __attribute__((deprecated("use ommxGetLastBakeTimingsSized")))
#endif
    ;

/* ---- per-baker knobs ----
 * Tuning and test switches are state of ONE baker, set explicitly through this call; the library reads no environment variables.
 * A value of 0 restores the default.  Unknown knobs -> INVALID_ARGUMENT. */
typedef enum ommxBakerKnob {
    ommxBakerKnob_Reserved0        = 0, /* (was ommxBakerKnob_SetupKeyBits, a test switch of rounds 1 - 5: the work-item ids are the reference's own 64-bit ids
                                           since round 6, see omm_amd/csrc/vm_id.h; setting it has no effect) */
    ommxBakerKnob_ShardChunkBytes  = 1, /* sharded bake: bytes per rank and chunk of the block all-gather (>= 256; default 64 MiB, at most 8 chunks) */
    ommxBakerKnob_StreamChunks     = 2, /* ommCpuBake: number of ranges (of work items, in the order of the result) whose finished OMM blocks are copied to
                                           their place in the host array while the following ranges are being classified; a value (1..32) forces streaming
                                           whatever the size of the bake (1 = classify everything, then one copy).  Default: bakes with >= 64 MiB of packed
                                           states stream, one range per 32 MiB, at most 32 */
    ommxBakerKnob_GenericPass      = 3, /* where micro-triangles that span several texels (asset-sized triangles) are classified: 1 = inside the persistent
                                           classification launch, one lane each; 2 = queued and classified by a second launch whose lanes pull them as their walks end (not for streamed
                                           bakes); 0 = automatic: 2 when the texels under the triangles outweigh the micro-triangles */
    ommxBakerKnob_RetainMemory     = 4, /* what a baker keeps between bakes.  0 / default: the working set of a bake (device arenas, streams), up to six idle device result
                                           blocks and up to two idle PINNED host blocks for arrayData (a fresh 1.3 GB host block costs more in page faults and munmap than the
                                           PCIe copy of its contents) stay with the baker until it is destroyed; 1: nothing is retained -- every block goes back to the system
                                           when the result that uses it is destroyed.  For pipelines that create many bakers, or bake rarely.  See also ommxTrimBaker. */
    ommxBakerKnob_ResultTransfer   = 5, /* ommCpuBake: how a large arrayData reaches the caller's memory, one of ommxResultTransfer_*.  Auto (0 / default): Compressed when the
                                           bake carries ommCpuBakeFlags_EnableInternalThreads and the process has >= 6 CPUs (affinity / cgroup quota), else Streamed */
    ommxBakerKnob_ExpandThreads    = 6, /* threads (the caller's included, <= 64) that expand a compressed result; 0 / default: three quarters of the CPUs the process may use, at most 12 */
    ommxBakerKnob_Devices          = 7, /* ommCpuBake over N >= 2 devices of this process (at most 16): one host thread per device, rank r on HIP device (baker's + r) mod the
                                           device count; the texture is copied to the other devices by the first bake that needs it; every device classifies its share of
                                           the work items and sends its own blocks to the host as a codec stream over its own PCIe link.  Not for bakes with near-duplicate
                                           merging / maxArrayDataSize or per-triangle formats (those keep to one device).  0 / 1: one device.
                                           HELPER THREADS: a baker starts threads of its own only with the caller's permission.  That permission is
                                           ommCpuBakeFlags_EnableInternalThreads on the bake (the automatic choice of the compressed transfer looks at it), OR setting one of the two
                                           knobs that cannot work without threads: ommxBakerKnob_ResultTransfer = ommxResultTransfer_Compressed (helper threads expand the
                                           result) and ommxBakerKnob_Devices >= 2 (a host thread per device + the expansion).  Setting such a knob IS the permission, whatever
                                           the bake's flags say; without either, ommCpuBake runs on the calling thread only, like the reference without the flag */
    ommxBakerKnob_HelperAffinity   = 8, /* where the baker's helper threads run while they fill a large arrayData.  0 / default: each helper thread is bound to its own slice of
                                           the physical cores of the NUMA node that holds the array (pthread_setaffinity_np on the baker's OWN threads only -- never on the
                                           caller's; measured on the two-socket host of the GPU box: 16.6 - 16.8 ms per bake bound, 17 - 26 ms unbound).  1: the threads are
                                           left where the scheduler puts them (for hosts whose thread placement is managed from outside) */
    ommxBakerKnob_ZeroAhead        = 9, /* compressed transfer: 0 / default: while the device bakes, the helper threads zero the idle result block the previous bake of this
                                           baker left (default allocator only), and the expansion does not write codec blocks of zeros again (ommxBakeTimings::prefilledBytes,
                                           expandSkippedBytes); 1: off -- the helper threads only run during the expansion */
    ommxBakerKnob_MAX_NUM          = 10
} ommxBakerKnob;
typedef enum ommxResultTransfer {
    ommxResultTransfer_Auto        = 0,
    ommxResultTransfer_Plain       = 1, /* one device-to-host copy of the finished array after the bake (what small results always get) */
    ommxResultTransfer_Streamed    = 2, /* rounds 3 - 4: finished blocks are placed and copied to their final offsets by the DMA engine WHILE the classification runs
                                           (>= 64 MiB of packed states; ommxBakerKnob_StreamChunks); bound by the PCIe link */
    ommxResultTransfer_Compressed  = 3  /* round 5: the finished array crosses PCIe as a codec stream (one nibble per 16-byte unit: the state it repeats, or raw: 6 % of the
                                           bytes at the metric configuration) and is expanded into the caller's array by helper threads of the baker (ommxBakerKnob_ExpandThreads) */
} ommxResultTransfer;
OMM_MI355X_API ommResult ommxSetBakerKnob(ommBaker baker, ommxBakerKnob knob, uint64_t value);
/* Gives every idle pooled block of the baker back to the system now: pinned host blocks, device result blocks, device working sets of finished bakes.  Results
 * that are still alive keep their memory.  Safe to call at any time from any thread; the next bake allocates again. */
OMM_MI355X_API ommResult ommxTrimBaker(ommBaker baker);

/* ---- device-resident bake ----
 * Same contract as ommCpuBake (include/omm_mi355x.h; reference omm.h:574) except for where the bulk data lives:
 *   in : desc->texCoords, desc->indexBuffer and desc->subdivisionLevels are DEVICE pointers (HBM of the current HIP device);
 *        desc->formats must be NULL.  All other fields, validation, result codes and log messages are those of ommCpuBake.
 *   out: an ommCpuBakeResultDesc whose arrayData, descArray and indexBuffer are DEVICE pointers; the two histograms are
 *        host arrays.  Buffers stay valid until ommxDestroyDeviceBakeResult.
 * Stream ordering: the library works on its own non-blocking HIP stream, which does NOT order against the caller's streams or the
 * null stream.  Device buffers handed in (inputs here, `words` / `gathered` of the sharded bake below) must be complete -- the
 * producing stream synchronised -- before the call; everything handed back is complete when the call returns.
 * The call returns when the result is complete (the library synchronises its own stream). */
typedef struct _ommxDeviceBakeResult* ommxDeviceBakeResult;
OMM_MI355X_API ommResult ommxBakeDevice(ommBaker baker, const ommCpuBakeInputDesc* deviceDesc, ommxDeviceBakeResult* outResult);
OMM_MI355X_API ommResult ommxGetDeviceBakeResultDesc(ommxDeviceBakeResult result, const ommCpuBakeResultDesc** desc);
OMM_MI355X_API ommResult ommxDestroyDeviceBakeResult(ommxDeviceBakeResult result);


/* ---- multi-GPU sharded bake (one process per GPU) ----
 * Every rank calls the same four functions with the SAME desc (device-resident inputs as for ommxBakeDevice); the caller
 * performs the two collectives in between with whatever transport it has (MPI, torch.distributed, ...).  ommxShardedBakeRccl further
 * down is the same bake as one call with the collectives done by the library.  The handle owns its device working set from Begin to
 * Destroy and holds no lock in between: other bakes on the same baker run concurrently, Destroy may come from any thread.
 *
 *   ommxShardedBegin   work-item setup + triage (replicated, cheap), then classification and digests of THIS rank's share of
 *                      the active work items (contiguous ranges of the per-level lists)
 *   ommxShardedGetMeta -> device array of `numWords` uint32 (4 per active item: state mask, known count, digest lo/hi), zero
 *                      for items of other ranks.           CALLER: all-reduce(SUM) over all ranks, in place.
 *   ommxShardedTail    replicated deterministic tail (promote, first-occurrence dedup, spatial sort, offsets) on the merged
 *                      metadata; packs this rank's surviving OMM blocks -> `contribution` (device), its size, and the common
 *                      padded size `strideBytes`.          CALLER: all-gather of strideBytes per rank into gathered[world][strideBytes].
 *   ommxShardedFinish  places every rank's blocks at their final arrayData offsets -> the same ommxDeviceBakeResult on every
 *                      rank, bit-identical to a single-GPU ommxBakeDevice of the same desc. */
typedef struct _ommxShardedBake* ommxShardedBake;
OMM_MI355X_API ommResult ommxShardedBegin(ommBaker baker, const ommCpuBakeInputDesc* deviceDesc, uint32_t rank, uint32_t worldSize, ommxShardedBake* out);
OMM_MI355X_API ommResult ommxShardedGetMeta(ommxShardedBake bake, void** deviceWords, uint64_t* numWords);
OMM_MI355X_API ommResult ommxShardedTail(ommxShardedBake bake, void** contribution, uint64_t* contributionBytes, uint64_t* strideBytes);
OMM_MI355X_API ommResult ommxShardedFinish(ommxShardedBake bake, const void* gathered, ommxDeviceBakeResult* outResult);
OMM_MI355X_API ommResult ommxShardedDestroy(ommxShardedBake bake);

/* ---- the same sharded bake as ONE call, collectives included (RCCL over xGMI, issued by the library from C++) ----
 * Every rank calls ommxShardedBakeRccl with the SAME desc; the library runs Begin, the SUM all-reduce of the metadata words
 * (ncclAllReduce, in place, on the bake's own stream), Tail, the all-gather of the contributions and Finish.  The contributions cross the links
 * as codec streams -- one nibble per 16-byte unit (which of the four states it repeats, or "raw") plus the raw units: a few per cent of the bytes
 * for ordinary bakes -- and are decoded straight to their final arrayData offsets on arrival; if a rank's contribution does not shrink
 * below half its size, all ranks send the padded contributions themselves (ncclAllGather in chunks of <= 64 MiB per rank on a second stream,
 * each chunk scattered while the next one is on the wire).  A rank-local failure stops all ranks (status agreement before each exchange).  Every rank returns the same merged ommxDeviceBakeResult, bit-identical to a single-GPU ommxBakeDevice of the desc.
 * librccl.so.1 is bound with dlopen at the first call: callers that never shard need no RCCL.
 *
 * Communicator: either wrap an ncclComm_t the application already owns (ommxRcclCommWrap; not destroyed by the library), or let the
 * library create one: rank 0 calls ommxRcclGetUniqueId and distributes the 128 bytes by any means (MPI, a file, torch.distributed),
 * then every rank calls ommxRcclCommInitRank on its own HIP device (collective: it returns once all ranks have joined). */
typedef struct _ommxRcclComm* ommxRcclComm;
#define OMMX_RCCL_UNIQUE_ID_BYTES 128
OMM_MI355X_API ommResult ommxRcclGetUniqueId(void* outId, size_t idBytes);
OMM_MI355X_API ommResult ommxRcclCommInitRank(const void* id, size_t idBytes, uint32_t rank, uint32_t worldSize, ommxRcclComm* outComm);
OMM_MI355X_API ommResult ommxRcclCommWrap(void* ncclComm, ommxRcclComm* outComm);
OMM_MI355X_API ommResult ommxRcclCommDestroy(ommxRcclComm comm);
/* rank and size as the communicator itself reports them (a wrapped or library-made RCCL communicator: ncclCommUserRank / ncclCommCount asked again at the
 * time of the call; a communicator of caller collectives: the values it was made with) */
OMM_MI355X_API ommResult ommxRcclCommInfo(ommxRcclComm comm, uint32_t* outRank, uint32_t* outWorldSize);
OMM_MI355X_API ommResult ommxShardedBakeRccl(ommBaker baker, const ommCpuBakeInputDesc* deviceDesc, ommxRcclComm comm, ommxDeviceBakeResult* outResult);

/* ---- the one-call sharded bake over a transport of the caller (MPI with device pointers, torch.distributed, a test harness) ----
 * ommxCommFromCollectives makes a communicator whose two collectives are the caller's functions instead of RCCL's; ommxShardedBakeRccl takes it like
 * any other and runs the identical sequence (status agreement, metadata all-reduce, codec streams or raw chunks, scatter).  Both functions
 * work on DEVICE pointers and are collective: every rank calls them in the same order with the same counts.  They are stream-ordered like their RCCL
 * counterparts: `send` is complete once the work queued on `hipStream` so far has run, and work queued on `hipStream` after the call must see `recv`
 * (a blocking implementation synchronises the stream, exchanges, and returns).  `send` may equal `recv` in allReduceU32.  Return 0 for success.
 * The structure is copied; `user` must stay valid until ommxRcclCommDestroy. */
typedef enum ommxReduceOp { ommxReduceOp_Sum = 0, ommxReduceOp_Max = 1, ommxReduceOp_Min = 2 } ommxReduceOp;
typedef struct ommxCollectives
{
    int (*allReduceU32)(void* user, const void* send, void* recv, size_t count, ommxReduceOp op, void* hipStream);       /* `count` uint32 elements */
    int (*allGatherBytes)(void* user, const void* send, void* recv, size_t bytesPerRank, void* hipStream);               /* recv[rank r] = recv + r * bytesPerRank */
    void* user;
} ommxCollectives;
OMM_MI355X_API ommResult ommxCommFromCollectives(const ommxCollectives* collectives, uint32_t rank, uint32_t worldSize, ommxRcclComm* outComm);

#endif
