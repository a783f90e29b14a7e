/*
 * omm_mi355x.h -- C ABI of the MI355X-native opacity-micromap baker.
 *
 * This is the drop-in boundary: the shared library built from omm_amd/csrc exports
 * exactly these `omm*` symbols with exactly these struct layouts, so a host program
 * compiled against the OMM SDK's own C header links and runs against it unchanged.
 * Every declaration cites the reference interface it replaces
 * (paths relative to /root/reference/libraries/omm-lib/).
 *
 * All 25 `omm*` exports of the SDK library are present: the CPU-baker
 * surface is implemented; the GPU-baker / debug-image entry points are link-compatible stubs
 * that return ommResult_NOT_IMPLEMENTED (out of scope for this build).
 *
 * Plain C: no torch types, no C++ types, pointers + sizes only.
 */
#ifndef OMM_MI355X_H
#define OMM_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define OMM_MI355X_API extern "C" __attribute__((visibility("default")))
#else
#define OMM_MI355X_API
#endif

/* include/omm.h:17-19 -- the SDK version this ABI is layout-compatible with. */
#define OMM_VERSION_MAJOR 1
#define OMM_VERSION_MINOR 9
#define OMM_VERSION_BUILD 0

/* ---- opaque handles (include/omm.h:55-70). Baker / texture handles carry a 3-bit type
 *      tag in their low bits (src/omm_handle.h:17-54); the bake result is untagged. ---- */
typedef uint8_t ommBool;
typedef struct _ommBaker*                 ommBaker;
typedef struct _ommCpuBakeResult*         ommCpuBakeResult;
typedef struct _ommCpuTexture*            ommCpuTexture;
typedef struct _ommCpuSerializedResult*   ommCpuSerializedResult;
typedef struct _ommCpuDeserializedResult* ommCpuDeserializedResult;

/* include/omm.h:72-76 */
typedef void* (*ommAllocate)(void* userArg, size_t size, size_t alignment);
typedef void* (*ommReallocate)(void* userArg, void* memory, size_t size, size_t alignment);
typedef void  (*ommFree)(void* userArg, void* memory);

/* include/omm.h:78-87 */
typedef enum ommResult {
    ommResult_SUCCESS, ommResult_FAILURE, ommResult_INVALID_ARGUMENT,
    ommResult_INSUFFICIENT_SCRATCH_MEMORY, ommResult_NOT_IMPLEMENTED,
    ommResult_WORKLOAD_TOO_BIG, ommResult_MAX_NUM
} ommResult;

/* include/omm.h:89-96 */
typedef enum ommMessageSeverity {
    ommMessageSeverity_Info, ommMessageSeverity_PerfWarning, ommMessageSeverity_Error,
    ommMessageSeverity_Fatal, ommMessageSeverity_MAX_NUM
} ommMessageSeverity;

/* include/omm.h:98-104 -- the 2-bit micro-triangle states. */
typedef enum ommOpacityState {
    ommOpacityState_Transparent, ommOpacityState_Opaque,
    ommOpacityState_UnknownTransparent, ommOpacityState_UnknownOpaque
} ommOpacityState;

/* include/omm.h:106-112 -- negative index-buffer entries. */
typedef enum ommSpecialIndex {
    ommSpecialIndex_FullyTransparent = -1, ommSpecialIndex_FullyOpaque = -2,
    ommSpecialIndex_FullyUnknownTransparent = -3, ommSpecialIndex_FullyUnknownOpaque = -4
} ommSpecialIndex;

/* include/omm.h:114-122 -- value == bits per micro-triangle. */
typedef enum ommFormat {
    ommFormat_INVALID, ommFormat_OC1_2_State = 1, ommFormat_OC1_4_State = 2, ommFormat_MAX_NUM = 3
} ommFormat;

/* include/omm.h:124-134 */
typedef enum ommUnknownStatePromotion {
    ommUnknownStatePromotion_Nearest, ommUnknownStatePromotion_ForceOpaque,
    ommUnknownStatePromotion_ForceTransparent, ommUnknownStatePromotion_MAX_NUM
} ommUnknownStatePromotion;

/* include/omm.h:136-141 */
typedef enum ommBakerType { ommBakerType_GPU, ommBakerType_CPU, ommBakerType_MAX_NUM } ommBakerType;

/* include/omm.h:143-149 */
typedef enum ommTexCoordFormat {
    ommTexCoordFormat_UV16_UNORM, ommTexCoordFormat_UV16_FLOAT, ommTexCoordFormat_UV32_FLOAT,
    ommTexCoordFormat_MAX_NUM
} ommTexCoordFormat;

/* include/omm.h:151-159 (note the non-monotonic numbering) */
typedef enum ommIndexFormat {
    ommIndexFormat_UINT_16, ommIndexFormat_UINT_32, ommIndexFormat_UINT_8, ommIndexFormat_MAX_NUM
} ommIndexFormat;

/* include/omm.h:161-169 */
typedef enum ommTextureAddressMode {
    ommTextureAddressMode_Wrap, ommTextureAddressMode_Mirror, ommTextureAddressMode_Clamp,
    ommTextureAddressMode_Border, ommTextureAddressMode_MirrorOnce, ommTextureAddressMode_MAX_NUM
} ommTextureAddressMode;

/* include/omm.h:171-176 */
typedef enum ommTextureFilterMode {
    ommTextureFilterMode_Nearest, ommTextureFilterMode_Linear, ommTextureFilterMode_MAX_NUM
} ommTextureFilterMode;

/* include/omm.h:178-183 */
typedef enum ommAlphaMode { ommAlphaMode_Test, ommAlphaMode_Blend, ommAlphaMode_MAX_NUM } ommAlphaMode;

/* include/omm.h:185-189 */
typedef enum ommCpuSerializeFlags { ommCpuSerializeFlags_None, ommCpuSerializeFlags_Compress } ommCpuSerializeFlags;

/* include/omm.h:191-196 */
typedef struct ommLibraryDesc { uint8_t versionMajor, versionMinor, versionBuild; } ommLibraryDesc;

/* include/omm.h:198-212 */
typedef struct ommSamplerDesc {
    ommTextureAddressMode addressingMode;
    ommTextureFilterMode  filter;
    float                 borderAlpha;
} ommSamplerDesc;

static inline ommSamplerDesc ommSamplerDescDefault(void) {
    ommSamplerDesc v = { ommTextureAddressMode_MAX_NUM, ommTextureFilterMode_MAX_NUM, 0.f };
    return v;
}

/* include/omm.h:214-242. All host allocations made on behalf of a baker go through
 * these callbacks; NULL allocate selects an internal aligned malloc. */
typedef struct ommMemoryAllocatorInterface {
    ommAllocate   allocate;
    ommReallocate reallocate;
    ommFree       free;
    void*         userArg;
} ommMemoryAllocatorInterface;

static inline ommMemoryAllocatorInterface ommMemoryAllocatorInterfaceDefault(void) {
    ommMemoryAllocatorInterface v = { NULL, NULL, NULL, NULL };
    return v;
}

/* include/omm.h:244-258 */
typedef void (*ommMessageCallback)(ommMessageSeverity severity, const char* message, void* userArg);
typedef struct ommMessageInterface { ommMessageCallback messageCallback; void* userArg; } ommMessageInterface;

static inline ommMessageInterface ommMessageInterfaceDefault(void) {
    ommMessageInterface v = { NULL, NULL };
    return v;
}

/* include/omm.h:260-274 */
typedef struct ommBakerCreationDesc {
    ommBakerType                type;
    ommMemoryAllocatorInterface memoryAllocatorInterface;
    ommMessageInterface         messageInterface;
} ommBakerCreationDesc;

static inline ommBakerCreationDesc ommBakerCreationDescDefault(void) {
    ommBakerCreationDesc v;
    v.type = ommBakerType_MAX_NUM;
    v.memoryAllocatorInterface = ommMemoryAllocatorInterfaceDefault();
    v.messageInterface = ommMessageInterfaceDefault();
    return v;
}

/* include/omm.h:282-295 */
typedef enum ommCpuTextureFormat { ommCpuTextureFormat_UNORM8, ommCpuTextureFormat_FP32, ommCpuTextureFormat_MAX_NUM } ommCpuTextureFormat;
typedef enum ommCpuTextureFlags { ommCpuTextureFlags_None, ommCpuTextureFlags_DisableZOrder = 1u << 0 } ommCpuTextureFlags;

/* include/omm.h:298-334 */
typedef enum ommCpuBakeFlags {
    ommCpuBakeFlags_None,
    ommCpuBakeFlags_EnableInternalThreads        = 1u << 0,
    ommCpuBakeFlags_DisableSpecialIndices        = 1u << 1,
    ommCpuBakeFlags_Force32BitIndices            = 1u << 2,
    ommCpuBakeFlags_DisableDuplicateDetection    = 1u << 3,
    ommCpuBakeFlags_EnableNearDuplicateDetection = 1u << 4,
    ommCpuBakeFlags_EnableValidation             = 1u << 5,
    ommCpuBakeFlags_Allow8BitIndices             = 1u << 6,
    ommCpuBakeFlags_EnableWorkloadValidation     = 1u << 5
} ommCpuBakeFlags;

/* include/omm.h:340-356. rowPitch quirk mirrored from src/texture_impl.cpp:139-184:
 * bytes when the texture is created with DisableZOrder, texels otherwise; 0 = tight. */
typedef struct ommCpuTextureMipDesc {
    uint32_t    width;
    uint32_t    height;
    uint32_t    rowPitch;
    const void* textureData;
} ommCpuTextureMipDesc;

static inline ommCpuTextureMipDesc ommCpuTextureMipDescDefault(void) {
    ommCpuTextureMipDesc v = { 0, 0, 0, NULL };
    return v;
}

/* include/omm.h:358-378. alphaCutoff >= 0 makes the texture carry a summed-area table
 * of (alpha > cutoff), which enables the coarse classification pass. */
typedef struct ommCpuTextureDesc {
    ommCpuTextureFormat         format;
    ommCpuTextureFlags          flags;
    const ommCpuTextureMipDesc* mips;
    uint32_t                    mipCount;
    float                       alphaCutoff;
} ommCpuTextureDesc;

static inline ommCpuTextureDesc ommCpuTextureDescDefault(void) {
    ommCpuTextureDesc v;
    v.format = ommCpuTextureFormat_MAX_NUM; v.flags = ommCpuTextureFlags_None;
    v.mips = NULL; v.mipCount = 0; v.alphaCutoff = -1.f;
    return v;
}

/* include/omm.h:380-460 -- the bake input contract; sizeof == 136 on LP64
 * (src/serialize_impl.cpp:86 asserts the same). Field names keep the SDK spelling; the
 * deprecated union aliases of the SDK header are not reproduced (same offsets). */
typedef struct ommCpuBakeInputDesc {
    ommCpuBakeFlags          bakeFlags;
    ommCpuTexture            texture;
    ommSamplerDesc           runtimeSamplerDesc;
    ommAlphaMode             alphaMode;
    ommTexCoordFormat        texCoordFormat;
    const void*              texCoords;
    uint32_t                 texCoordStrideInBytes;   /* 0 => packed */
    ommIndexFormat           indexFormat;
    const void*              indexBuffer;
    uint32_t                 indexCount;              /* 3 * triangle count */
    float                    dynamicSubdivisionScale; /* <= 0 disables the per-triangle level heuristic */
    float                    rejectionThreshold;
    float                    alphaCutoff;
    float                    nearDuplicateDeduplicationFactor;
    ommOpacityState          alphaCutoffLessEqual;
    ommOpacityState          alphaCutoffGreater;
    ommFormat                format;
    const ommFormat*         formats;
    ommUnknownStatePromotion unknownStatePromotion;
    ommSpecialIndex          unresolvedTriState;
    uint8_t                  maxSubdivisionLevel;     /* [0,12] */
    uint32_t                 maxArrayDataSize;
    const uint8_t*           subdivisionLevels;       /* per triangle, > 12 => global/dynamic */
    uint64_t                 maxWorkloadSize;
} ommCpuBakeInputDesc;

/* include/omm.h:462-490 */
static inline ommCpuBakeInputDesc ommCpuBakeInputDescDefault(void) {
    ommCpuBakeInputDesc v;
    v.bakeFlags = ommCpuBakeFlags_None;
    v.texture = 0;
    v.runtimeSamplerDesc = ommSamplerDescDefault();
    v.alphaMode = ommAlphaMode_MAX_NUM;
    v.texCoordFormat = ommTexCoordFormat_MAX_NUM;
    v.texCoords = NULL;
    v.texCoordStrideInBytes = 0;
    v.indexFormat = ommIndexFormat_MAX_NUM;
    v.indexBuffer = NULL;
    v.indexCount = 0;
    v.dynamicSubdivisionScale = 2;
    v.rejectionThreshold = 0;
    v.alphaCutoff = 0.5f;
    v.nearDuplicateDeduplicationFactor = 0.15f;
    v.alphaCutoffLessEqual = ommOpacityState_Transparent;
    v.alphaCutoffGreater = ommOpacityState_Opaque;
    v.format = ommFormat_OC1_4_State;
    v.formats = NULL;
    v.unknownStatePromotion = ommUnknownStatePromotion_ForceOpaque;
    v.unresolvedTriState = ommSpecialIndex_FullyUnknownOpaque;
    v.maxSubdivisionLevel = 8;
    v.maxArrayDataSize = 0xFFFFFFFF;
    v.subdivisionLevels = NULL;
    v.maxWorkloadSize = 0xFFFFFFFFFFFFFFFFull;
    return v;
}

/* include/omm.h:492-500 */
typedef struct ommCpuOpacityMicromapDesc {
    uint32_t offset;           /* byte offset into arrayData */
    uint16_t subdivisionLevel;
    uint16_t format;
} ommCpuOpacityMicromapDesc;

/* include/omm.h:502-510 */
typedef struct ommCpuOpacityMicromapUsageCount {
    uint32_t count;
    uint16_t subdivisionLevel;
    uint16_t format;
} ommCpuOpacityMicromapUsageCount;

/* include/omm.h:512-530 -- the bake output contract. All pointers are owned by the
 * ommCpuBakeResult and stay valid until ommCpuDestroyBakeResult. */
typedef struct ommCpuBakeResultDesc {
    const void*                            arrayData;
    uint32_t                               arrayDataSize;
    const ommCpuOpacityMicromapDesc*       descArray;
    uint32_t                               descArrayCount;
    const ommCpuOpacityMicromapUsageCount* descArrayHistogram;
    uint32_t                               descArrayHistogramCount;
    const void*                            indexBuffer;   /* signed entries: >= 0 desc index, < 0 special */
    uint32_t                               indexCount;
    ommIndexFormat                         indexFormat;
    const ommCpuOpacityMicromapUsageCount* indexHistogram;
    uint32_t                               indexHistogramCount;
} ommCpuBakeResultDesc;

/* include/omm.h:532-566 -- blob (de)serialisation descriptors (next-tier row f1). */
typedef struct ommCpuBlobDesc { void* data; uint64_t size; } ommCpuBlobDesc;
typedef struct ommCpuDeserializedDesc {
    ommCpuSerializeFlags        flags;
    int                         numInputDescs;
    const ommCpuBakeInputDesc*  inputDescs;
    int                         numResultDescs;
    const ommCpuBakeResultDesc* resultDescs;
} ommCpuDeserializedDesc;

/* include/omm.h:1170-1181 -- aggregate statistics every reference known-answer test is
 * phrased in (src/debug_impl.cpp:512-641). */
typedef struct ommDebugStats {
    uint64_t totalOpaque;
    uint64_t totalTransparent;
    uint64_t totalUnknownTransparent;
    uint64_t totalUnknownOpaque;
    uint32_t totalFullyOpaque;
    uint32_t totalFullyTransparent;
    uint32_t totalFullyUnknownOpaque;
    uint32_t totalFullyUnknownTransparent;
    float    knownAreaMetric;
} ommDebugStats;

/* ---- entry points ---- */

/* include/omm.h:276, src/bake.cpp:36-42 */
OMM_MI355X_API ommLibraryDesc ommGetLibraryDesc(void);
/* include/omm.h:278, src/bake.cpp:410-457 */
OMM_MI355X_API ommResult ommCreateBaker(const ommBakerCreationDesc* bakeCreationDesc, ommBaker* outBaker);
/* include/omm.h:280, src/bake.cpp:459-479 */
OMM_MI355X_API ommResult ommDestroyBaker(ommBaker baker);
/* include/omm.h:568, src/bake.cpp:44-69, src/texture_impl.cpp:77-224 */
OMM_MI355X_API ommResult ommCpuCreateTexture(ommBaker baker, const ommCpuTextureDesc* desc, ommCpuTexture* outTexture);
/* include/omm.h:570, src/bake.cpp:71-82, src/texture_impl.cpp:280-325 */
OMM_MI355X_API ommResult ommCpuGetTextureDesc(ommCpuTexture texture, ommCpuTextureDesc* outDesc);
/* include/omm.h:572, src/bake.cpp:84-101 */
OMM_MI355X_API ommResult ommCpuDestroyTexture(ommBaker baker, ommCpuTexture texture);
/* include/omm.h:574, src/bake.cpp:103-116, src/bake_cpu_impl.cpp:105-119,1923-1985 -- THE hot path. */
OMM_MI355X_API ommResult ommCpuBake(ommBaker baker, const ommCpuBakeInputDesc* bakeInputDesc, ommCpuBakeResult* outBakeResult);
/* include/omm.h:576, src/bake.cpp:118-127 */
OMM_MI355X_API ommResult ommCpuDestroyBakeResult(ommCpuBakeResult bakeResult);
/* include/omm.h:578, src/bake.cpp:129-135 */
OMM_MI355X_API ommResult ommCpuGetBakeResultDesc(ommCpuBakeResult bakeResult, const ommCpuBakeResultDesc** desc);
/* ---- blob (de)serialisation: include/omm.h:580-594, src/bake.cpp:136-260, src/serialize_impl.cpp (format v5, reads v1..v5, LZ4 optional).
 * The SDK header declares the desc arguments of ommCpuSerialize / ommCpuDeserialize as C++ references inside extern "C"; a
 * reference is passed as a pointer, so these pointer declarations are binary compatible with it. ---- */
OMM_MI355X_API ommResult ommCpuSerialize(ommBaker baker, const ommCpuDeserializedDesc* desc, ommCpuSerializedResult* outResult);
OMM_MI355X_API ommResult ommCpuGetSerializedResultDesc(ommCpuSerializedResult result, const ommCpuBlobDesc** desc);
OMM_MI355X_API ommResult ommCpuDestroySerializedResult(ommCpuSerializedResult result);
OMM_MI355X_API ommResult ommCpuDeserialize(ommBaker baker, const ommCpuBlobDesc* desc, ommCpuDeserializedResult* outResult);
OMM_MI355X_API ommResult ommCpuGetDeserializedDesc(ommCpuDeserializedResult result, const ommCpuDeserializedDesc** desc);
OMM_MI355X_API ommResult ommCpuDestroyDeserializedResult(ommCpuDeserializedResult result);
/* include/omm.h:1201, src/debug_impl.cpp:643-652 -- statistics of a result desc; knownAreaMetric stays 0 (no per-triangle areas). */
OMM_MI355X_API ommResult ommDebugGetStats(ommBaker baker, const ommCpuBakeResultDesc* res, ommDebugStats* out);
/* include/omm.h:1202, src/bake.cpp:359-386, src/debug_impl.cpp:512-641 -- the same on a result OBJECT of ommCpuBake, which keeps the
 * UV-space area of every input triangle (src/bake_cpu_impl.cpp:1904-1915): knownAreaMetric = area-weighted known fraction. */
OMM_MI355X_API ommResult ommDebugGetStats2(ommBaker baker, ommCpuBakeResult res, ommDebugStats* out);
/* include/omm.h:1204, src/bake.cpp:388-408, src/debug_impl.cpp:654-670 -- writes blob bytes to `path` (the SDK declares `data` as a
 * C++ reference inside extern "C": a pointer at the ABI level, like ommCpuSerialize above). */
OMM_MI355X_API ommResult ommDebugSaveBinaryToDisk(ommBaker baker, const ommCpuBlobDesc* data, const char* path);

/* ---- link compatibility with the rest of the SDK's export list --------------------------------------------
 * The SDK's GPU baker (include/omm.h:596-1141: a dispatch-chain generator for the client's D3D12 / Vulkan RHI, HLSL shaders) and its
 * PNG debug dump are out of scope for this build.  Their seven entry points are exported so that any binary linked against the SDK's
 * libomm-lib.so -- which may reference them without calling them -- also loads against this library; called, they validate the
 * handles like src/bake.cpp:262-336 and return ommResult_NOT_IMPLEMENTED with a log line.  The desc types are opaque here: only
 * pointers cross the ABI. */
typedef struct _ommGpuPipeline*             ommGpuPipeline;             /* include/omm.h:596-597 */
typedef struct ommGpuPipelineConfigDesc     ommGpuPipelineConfigDesc;   /* include/omm.h:944-948 (opaque) */
typedef struct ommGpuPipelineInfoDesc       ommGpuPipelineInfoDesc;     /* include/omm.h:1085-1095 (opaque) */
typedef struct ommGpuDispatchConfigDesc     ommGpuDispatchConfigDesc;   /* include/omm.h:997-1055 (opaque) */
typedef struct ommGpuPreDispatchInfo        ommGpuPreDispatchInfo;      /* include/omm.h:958-980 (opaque) */
typedef struct ommGpuDispatchChain          ommGpuDispatchChain;        /* include/omm.h:1116-1122 (opaque) */
typedef struct ommDebugSaveImagesDesc       ommDebugSaveImagesDesc;     /* include/omm.h:1143-1156 (opaque) */
typedef enum ommGpuResourceType {                                       /* include/omm.h:608-636 (passed by value) */
    ommGpuResourceType_IN_ALPHA_TEXTURE, ommGpuResourceType_IN_TEXCOORD_BUFFER, ommGpuResourceType_IN_INDEX_BUFFER,
    ommGpuResourceType_IN_SUBDIVISION_LEVEL_BUFFER, ommGpuResourceType_OUT_OMM_ARRAY_DATA, ommGpuResourceType_OUT_OMM_DESC_ARRAY,
    ommGpuResourceType_OUT_OMM_DESC_ARRAY_HISTOGRAM, ommGpuResourceType_OUT_OMM_INDEX_BUFFER, ommGpuResourceType_OUT_OMM_INDEX_HISTOGRAM,
    ommGpuResourceType_OUT_POST_DISPATCH_INFO, ommGpuResourceType_TRANSIENT_POOL_BUFFER, ommGpuResourceType_STATIC_VERTEX_BUFFER,
    ommGpuResourceType_STATIC_INDEX_BUFFER, ommGpuResourceType_MAX_NUM
} ommGpuResourceType;
/* include/omm.h:1127-1141, src/bake.cpp:262-312 */
OMM_MI355X_API ommResult ommGpuGetStaticResourceData(ommGpuResourceType resource, uint8_t* data, size_t* outByteSize);
OMM_MI355X_API ommResult ommGpuCreatePipeline(ommBaker baker, const ommGpuPipelineConfigDesc* pipelineCfg, ommGpuPipeline* outPipeline);
OMM_MI355X_API ommResult ommGpuDestroyPipeline(ommBaker baker, ommGpuPipeline pipeline);
OMM_MI355X_API ommResult ommGpuGetPipelineDesc(ommGpuPipeline pipeline, const ommGpuPipelineInfoDesc** outPipelineDesc);
OMM_MI355X_API ommResult ommGpuGetPreDispatchInfo(ommGpuPipeline pipeline, const ommGpuDispatchConfigDesc* config, ommGpuPreDispatchInfo* outPreDispatchInfo);
OMM_MI355X_API ommResult ommGpuDispatch(ommGpuPipeline pipeline, const ommGpuDispatchConfigDesc* config, const ommGpuDispatchChain** outDispatchDesc);
/* include/omm.h:1199, src/bake.cpp:314-336 */
OMM_MI355X_API ommResult ommDebugSaveAsImages(ommBaker baker, const ommCpuBakeInputDesc* bakeInputDesc, const ommCpuBakeResultDesc* res, const ommDebugSaveImagesDesc* desc);

#endif /* OMM_MI355X_H */
