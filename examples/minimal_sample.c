/* Minimal C user of the drop-in library: the donut / triangle-fan sample of the SDK's support/tests/test_minimal_sample.cpp,
 * written against include/omm_mi355x.h only (an SDK user keeps including the SDK's own omm.h -- same names, same layouts).
 *   gcc -std=c99 -Iinclude examples/minimal_sample.c -o /tmp/minimal_sample -Lomm_amd/lib -lomm-lib -lm -Wl,-rpath,$PWD/omm_amd/lib
 * Prints one line per micro-map: level, byte offset, and the number of opaque micro-triangles. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "omm_mi355x.h"

static void log_cb(ommMessageSeverity severity, const char* message, void* user) { (void)user; fprintf(stderr, "[omm %d] %s\n", (int)severity, message); }

int main(void)
{
    enum { W = 256, H = 256 };
    float* alpha = (float*)malloc(sizeof(float) * W * H);
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            const float dx = (float)i / (float)W - 0.5f, dy = (float)j / (float)W - 0.5f;
            const float r = sqrtf(dx * dx + dy * dy);
            alpha[i + j * W] = (r > 0.2f && r < 0.3f) ? 1.f : 0.f;
        }
    const float texCoords[5][2] = { { 0.05f, 0.50f }, { 0.50f, 0.05f }, { 0.50f, 0.50f }, { 0.95f, 0.50f }, { 0.50f, 0.95f } };
    const uint32_t indices[12] = { 0, 1, 2, 1, 3, 2, 3, 4, 2, 2, 4, 0 };
    const uint8_t levels[4] = { 2, 3, 4, 5 };

    ommBakerCreationDesc bd; memset(&bd, 0, sizeof bd);
    bd.type = ommBakerType_CPU;                 /* the "CPU baker" entry points; the work runs on the MI355X */
    bd.messageInterface.messageCallback = log_cb;
    ommBaker baker = 0;
    if (ommCreateBaker(&bd, &baker) != ommResult_SUCCESS) return 1;

    ommCpuTextureMipDesc mip; memset(&mip, 0, sizeof mip);
    mip.width = W; mip.height = H; mip.textureData = alpha;
    ommCpuTextureDesc td; memset(&td, 0, sizeof td);
    td.format = ommCpuTextureFormat_FP32; td.mips = &mip; td.mipCount = 1; td.alphaCutoff = -1.f;
    ommCpuTexture tex = 0;
    if (ommCpuCreateTexture(baker, &td, &tex) != ommResult_SUCCESS) return 2;

    ommCpuBakeInputDesc in; memset(&in, 0, sizeof in);
    in.bakeFlags = ommCpuBakeFlags_EnableValidation;
    in.texture = tex;
    in.runtimeSamplerDesc.addressingMode = ommTextureAddressMode_Clamp;
    in.runtimeSamplerDesc.filter = ommTextureFilterMode_Linear;
    in.alphaMode = ommAlphaMode_Test;
    in.texCoordFormat = ommTexCoordFormat_UV32_FLOAT; in.texCoords = texCoords; in.texCoordStrideInBytes = 8;
    in.indexFormat = ommIndexFormat_UINT_32; in.indexBuffer = indices; in.indexCount = 12;
    in.subdivisionLevels = levels;
    in.alphaCutoff = 0.5f;
    in.alphaCutoffLessEqual = ommOpacityState_Transparent; in.alphaCutoffGreater = ommOpacityState_Opaque;
    in.format = ommFormat_OC1_2_State;
    in.unknownStatePromotion = ommUnknownStatePromotion_ForceOpaque;
    in.unresolvedTriState = ommSpecialIndex_FullyUnknownOpaque;
    in.maxSubdivisionLevel = 8;
    in.maxArrayDataSize = 0xFFFFFFFFu;
    in.maxWorkloadSize = 0xFFFFFFFFFFFFFFFFull;
    in.rejectionThreshold = 0.f; in.dynamicSubdivisionScale = 2.f; in.nearDuplicateDeduplicationFactor = 0.15f;

    ommCpuBakeResult result = 0;
    const ommResult r = ommCpuBake(baker, &in, &result);
    if (r != ommResult_SUCCESS) { fprintf(stderr, "ommCpuBake failed: %d\n", (int)r); return 3; }
    const ommCpuBakeResultDesc* out = 0;
    if (ommCpuGetBakeResultDesc(result, &out) != ommResult_SUCCESS) return 4;
    printf("%u micro-maps, %u bytes, %u triangle indices\n", out->descArrayCount, out->arrayDataSize, out->indexCount);
    for (uint32_t k = 0; k < out->descArrayCount; ++k) {
        const ommCpuOpacityMicromapDesc* d = &out->descArray[k];
        const uint32_t n = 1u << (2u * d->subdivisionLevel);
        uint32_t opaque = 0;
        for (uint32_t i = 0; i < n; ++i) opaque += (((const uint8_t*)out->arrayData)[d->offset + (i >> 3)] >> (i & 7u)) & 1u;
        printf("omm %u: level %u offset %u opaque %u / %u\n", k, (unsigned)d->subdivisionLevel, d->offset, opaque, n);
    }
    ommCpuDestroyBakeResult(result);
    ommCpuDestroyTexture(baker, tex);
    ommDestroyBaker(baker);
    free(alpha);
    return 0;
}
