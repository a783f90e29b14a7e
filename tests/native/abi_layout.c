/* Prints the ABI facts a drop-in depends on: size of every struct of the CPU-baker interface, offset of every field, value of every
 * enumerator used on the path.  Compiled twice by tests/test_abi_exports.py -- once against include/omm_mi355x.h and, where the
 * reference checkout exists, once against the SDK's own omm.h -- and the two outputs must be identical. */
#include <stdio.h>
#include <stddef.h>
#include OMM_HEADER

#define S(T) printf("sizeof " #T " %zu\n", sizeof(T))
#define F(T, f) printf("offsetof " #T "." #f " %zu\n", offsetof(T, f))
#define E(v) printf("enum " #v " %lld\n", (long long)(v))

int main(void)
{
    S(ommLibraryDesc); F(ommLibraryDesc, versionMajor); F(ommLibraryDesc, versionMinor); F(ommLibraryDesc, versionBuild);
    S(ommSamplerDesc); F(ommSamplerDesc, addressingMode); F(ommSamplerDesc, filter); F(ommSamplerDesc, borderAlpha);
    S(ommMemoryAllocatorInterface); F(ommMemoryAllocatorInterface, userArg);
    S(ommMessageInterface); F(ommMessageInterface, messageCallback); F(ommMessageInterface, userArg);
    S(ommBakerCreationDesc); F(ommBakerCreationDesc, type); F(ommBakerCreationDesc, memoryAllocatorInterface); F(ommBakerCreationDesc, messageInterface);
    S(ommCpuTextureMipDesc); F(ommCpuTextureMipDesc, width); F(ommCpuTextureMipDesc, height); F(ommCpuTextureMipDesc, rowPitch); F(ommCpuTextureMipDesc, textureData);
    S(ommCpuTextureDesc); F(ommCpuTextureDesc, format); F(ommCpuTextureDesc, flags); F(ommCpuTextureDesc, mips); F(ommCpuTextureDesc, mipCount); F(ommCpuTextureDesc, alphaCutoff);
    S(ommCpuBakeInputDesc);
    F(ommCpuBakeInputDesc, bakeFlags); F(ommCpuBakeInputDesc, texture); F(ommCpuBakeInputDesc, runtimeSamplerDesc); F(ommCpuBakeInputDesc, alphaMode);
    F(ommCpuBakeInputDesc, texCoordFormat); F(ommCpuBakeInputDesc, texCoords); F(ommCpuBakeInputDesc, texCoordStrideInBytes); F(ommCpuBakeInputDesc, indexFormat);
    F(ommCpuBakeInputDesc, indexBuffer); F(ommCpuBakeInputDesc, indexCount); F(ommCpuBakeInputDesc, dynamicSubdivisionScale); F(ommCpuBakeInputDesc, rejectionThreshold);
    F(ommCpuBakeInputDesc, alphaCutoff); F(ommCpuBakeInputDesc, nearDuplicateDeduplicationFactor); F(ommCpuBakeInputDesc, alphaCutoffLessEqual);
    F(ommCpuBakeInputDesc, alphaCutoffGreater); F(ommCpuBakeInputDesc, format); F(ommCpuBakeInputDesc, formats); F(ommCpuBakeInputDesc, unknownStatePromotion);
    F(ommCpuBakeInputDesc, unresolvedTriState); F(ommCpuBakeInputDesc, maxSubdivisionLevel); F(ommCpuBakeInputDesc, maxArrayDataSize);
    F(ommCpuBakeInputDesc, subdivisionLevels); F(ommCpuBakeInputDesc, maxWorkloadSize);
    S(ommCpuOpacityMicromapDesc); F(ommCpuOpacityMicromapDesc, offset); F(ommCpuOpacityMicromapDesc, subdivisionLevel); F(ommCpuOpacityMicromapDesc, format);
    S(ommCpuOpacityMicromapUsageCount); F(ommCpuOpacityMicromapUsageCount, count); F(ommCpuOpacityMicromapUsageCount, subdivisionLevel); F(ommCpuOpacityMicromapUsageCount, format);
    S(ommCpuBakeResultDesc);
    F(ommCpuBakeResultDesc, arrayData); F(ommCpuBakeResultDesc, arrayDataSize); F(ommCpuBakeResultDesc, descArray); F(ommCpuBakeResultDesc, descArrayCount);
    F(ommCpuBakeResultDesc, descArrayHistogram); F(ommCpuBakeResultDesc, descArrayHistogramCount); F(ommCpuBakeResultDesc, indexBuffer); F(ommCpuBakeResultDesc, indexCount);
    F(ommCpuBakeResultDesc, indexFormat); F(ommCpuBakeResultDesc, indexHistogram); F(ommCpuBakeResultDesc, indexHistogramCount);
    S(ommCpuBlobDesc); F(ommCpuBlobDesc, data); F(ommCpuBlobDesc, size);
    S(ommCpuDeserializedDesc); F(ommCpuDeserializedDesc, flags); F(ommCpuDeserializedDesc, numInputDescs); F(ommCpuDeserializedDesc, inputDescs);
    F(ommCpuDeserializedDesc, numResultDescs); F(ommCpuDeserializedDesc, resultDescs);
    S(ommDebugStats);
    F(ommDebugStats, totalOpaque); F(ommDebugStats, totalTransparent); F(ommDebugStats, totalUnknownTransparent); F(ommDebugStats, totalUnknownOpaque);
    F(ommDebugStats, totalFullyOpaque); F(ommDebugStats, totalFullyTransparent); F(ommDebugStats, totalFullyUnknownOpaque); F(ommDebugStats, totalFullyUnknownTransparent);
    F(ommDebugStats, knownAreaMetric);
    E(ommResult_SUCCESS); E(ommResult_FAILURE); E(ommResult_INVALID_ARGUMENT); E(ommResult_INSUFFICIENT_SCRATCH_MEMORY); E(ommResult_NOT_IMPLEMENTED); E(ommResult_WORKLOAD_TOO_BIG);
    E(ommMessageSeverity_Info); E(ommMessageSeverity_PerfWarning); E(ommMessageSeverity_Error); E(ommMessageSeverity_Fatal);
    E(ommOpacityState_Transparent); E(ommOpacityState_Opaque); E(ommOpacityState_UnknownTransparent); E(ommOpacityState_UnknownOpaque);
    E(ommSpecialIndex_FullyTransparent); E(ommSpecialIndex_FullyOpaque); E(ommSpecialIndex_FullyUnknownTransparent); E(ommSpecialIndex_FullyUnknownOpaque);
    E(ommFormat_OC1_2_State); E(ommFormat_OC1_4_State);
    E(ommUnknownStatePromotion_Nearest); E(ommUnknownStatePromotion_ForceOpaque); E(ommUnknownStatePromotion_ForceTransparent);
    E(ommBakerType_GPU); E(ommBakerType_CPU);
    E(ommTexCoordFormat_UV16_UNORM); E(ommTexCoordFormat_UV16_FLOAT); E(ommTexCoordFormat_UV32_FLOAT);
    E(ommIndexFormat_UINT_16); E(ommIndexFormat_UINT_32); E(ommIndexFormat_UINT_8);
    E(ommTextureAddressMode_Wrap); E(ommTextureAddressMode_Mirror); E(ommTextureAddressMode_Clamp); E(ommTextureAddressMode_Border); E(ommTextureAddressMode_MirrorOnce);
    E(ommTextureFilterMode_Nearest); E(ommTextureFilterMode_Linear);
    E(ommAlphaMode_Test); E(ommAlphaMode_Blend);
    E(ommCpuTextureFormat_UNORM8); E(ommCpuTextureFormat_FP32);
    E(ommCpuTextureFlags_None); E(ommCpuTextureFlags_DisableZOrder);
    E(ommCpuBakeFlags_None); E(ommCpuBakeFlags_EnableInternalThreads); E(ommCpuBakeFlags_DisableSpecialIndices); E(ommCpuBakeFlags_Force32BitIndices);
    E(ommCpuBakeFlags_DisableDuplicateDetection); E(ommCpuBakeFlags_EnableNearDuplicateDetection); E(ommCpuBakeFlags_EnableValidation); E(ommCpuBakeFlags_Allow8BitIndices);
    E(ommCpuSerializeFlags_None); E(ommCpuSerializeFlags_Compress);
    return 0;
}
