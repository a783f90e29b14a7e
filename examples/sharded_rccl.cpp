// C++ user of the multi-GPU entry point: one process per GPU, the library issues the RCCL collectives itself (ommxShardedBakeRccl).
//
//   hipcc -std=c++17 -Iinclude examples/sharded_rccl.cpp -o /tmp/sharded_rccl -Lomm_amd/lib -lomm-lib -Wl,-rpath,$PWD/omm_amd/lib
//   /tmp/sharded_rccl                      one rank: creates its own unique id (what the GPU test of this repository runs)
//   /tmp/sharded_rccl <rank> <world> <id-file>   several ranks on one node: rank 0 writes the 128-byte RCCL unique id to <id-file>,
//                                                the others wait for it; rank r uses HIP device r
//
// Every rank bakes the same mesh; the active work items are split over the ranks, item metadata travel through ncclAllReduce and the
// surviving OMM blocks through a chunked ncclAllGather, and every rank ends with the merged result in its HBM.  The program checks
// that result byte for byte against a single-GPU ommxBakeDevice of the same desc on the same device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <vector>
#include "omm_mi355x.h"
#include "omm_mi355x_ext.h"

#define HIP(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP failure: %s\n", #x); return 10; } } while (0)
#define OMM(x) do { const ommResult r_ = (x); if (r_ != ommResult_SUCCESS) { fprintf(stderr, "%s -> %d\n", #x, (int)r_); return 11; } } while (0)

static void log_cb(ommMessageSeverity severity, const char* message, void*) { fprintf(stderr, "[omm %d] %s\n", (int)severity, message); }
static uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
static float unit(uint32_t x) { return (float)(hash32(x) >> 8) * (1.f / 16777216.f); }

static int fetch(const void* dev, size_t bytes, std::vector<uint8_t>& out) { out.resize(bytes); return bytes ? (int)hipMemcpy(out.data(), dev, bytes, hipMemcpyDeviceToHost) : 0; }

int main(int argc, char** argv)
{
    const uint32_t rank = argc > 2 ? (uint32_t)atoi(argv[1]) : 0, world = argc > 2 ? (uint32_t)atoi(argv[2]) : 1;
    const char* idFile = argc > 3 ? argv[3] : nullptr;
    HIP(hipSetDevice((int)rank));

    // alpha texture: soft discs on a 512^2 grid
    const int W = 512, H = 512;
    std::vector<uint8_t> alpha((size_t)W * H);
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            const float fx = fmodf((float)i, 64.f) - 32.f, fy = fmodf((float)j, 64.f) - 32.f;
            const float d = sqrtf(fx * fx + fy * fy);
            alpha[(size_t)i + (size_t)j * W] = (uint8_t)(d < 18.f ? 255 : (d < 24.f ? (int)(255.f * (24.f - d) / 6.f) : 0));
        }
    // triangles: random small triangles, mixed levels 3..7, a few duplicates
    const uint32_t T = 6000;
    std::vector<float> uv((size_t)T * 6); std::vector<uint32_t> idx((size_t)T * 3); std::vector<uint8_t> lvl(T);
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t s = (t % 17 == 5) ? t - 5 : t;   // every 17th triangle repeats an earlier one
        const float cx = unit(s * 8 + 1), cy = unit(s * 8 + 2);
        for (int k = 0; k < 3; ++k) { uv[(size_t)t * 6 + 2 * k] = cx + 0.03f * (unit(s * 8 + 3 + k) - 0.5f); uv[(size_t)t * 6 + 2 * k + 1] = cy + 0.03f * (unit(s * 8 + 6 + k) - 0.5f); idx[(size_t)t * 3 + k] = t * 3 + k; }
        lvl[t] = (uint8_t)(3 + hash32(s + 99) % 5);
    }

    ommBakerCreationDesc bd; memset(&bd, 0, sizeof bd);
    bd.type = ommBakerType_CPU; bd.messageInterface.messageCallback = log_cb;
    ommBaker baker = 0; OMM(ommCreateBaker(&bd, &baker));
    ommCpuTextureMipDesc mip; memset(&mip, 0, sizeof mip); mip.width = W; mip.height = H; mip.textureData = alpha.data();
    ommCpuTextureDesc td; memset(&td, 0, sizeof td); td.format = ommCpuTextureFormat_UNORM8; td.mips = &mip; td.mipCount = 1; td.alphaCutoff = 0.5f;
    ommCpuTexture tex = 0; OMM(ommCpuCreateTexture(baker, &td, &tex));

    void *dUv = nullptr, *dIdx = nullptr, *dLvl = nullptr;
    HIP(hipMalloc(&dUv, uv.size() * 4)); HIP(hipMalloc(&dIdx, idx.size() * 4)); HIP(hipMalloc(&dLvl, lvl.size()));
    HIP(hipMemcpy(dUv, uv.data(), uv.size() * 4, hipMemcpyHostToDevice)); HIP(hipMemcpy(dIdx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(dLvl, lvl.data(), lvl.size(), hipMemcpyHostToDevice));

    ommCpuBakeInputDesc in; memset(&in, 0, sizeof in);
    in.bakeFlags = ommCpuBakeFlags_EnableInternalThreads; in.texture = tex;
    in.runtimeSamplerDesc.addressingMode = ommTextureAddressMode_Wrap; in.runtimeSamplerDesc.filter = ommTextureFilterMode_Linear;
    in.alphaMode = ommAlphaMode_Test;
    in.texCoordFormat = ommTexCoordFormat_UV32_FLOAT; in.texCoords = dUv; in.texCoordStrideInBytes = 8;     // DEVICE pointers (ommxBakeDevice contract)
    in.indexFormat = ommIndexFormat_UINT_32; in.indexBuffer = dIdx; in.indexCount = T * 3; in.subdivisionLevels = (const uint8_t*)dLvl;
    in.alphaCutoff = 0.5f; in.alphaCutoffLessEqual = ommOpacityState_Transparent; in.alphaCutoffGreater = ommOpacityState_Opaque;
    in.format = ommFormat_OC1_4_State; in.unknownStatePromotion = ommUnknownStatePromotion_ForceOpaque; in.unresolvedTriState = ommSpecialIndex_FullyUnknownOpaque;
    in.maxSubdivisionLevel = 7; in.maxArrayDataSize = 0xFFFFFFFFu; in.maxWorkloadSize = 0xFFFFFFFFFFFFFFFFull; in.nearDuplicateDeduplicationFactor = 0.15f;

    // ---- communicator: the 128-byte unique id comes from rank 0's library ----
    uint8_t id[OMMX_RCCL_UNIQUE_ID_BYTES];
    if (rank == 0) {
        OMM(ommxRcclGetUniqueId(id, sizeof id));
        if (idFile) { FILE* f = fopen(idFile, "wb"); if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 12; fclose(f); }
    } else {
        if (!idFile) return 13;
        for (int tries = 0;; ++tries) { FILE* f = fopen(idFile, "rb"); if (f) { const size_t n = fread(id, 1, sizeof id, f); fclose(f); if (n == sizeof id) break; } if (tries > 600) return 14; usleep(100000); }
    }
    ommxRcclComm comm = 0; OMM(ommxRcclCommInitRank(id, sizeof id, rank, world, &comm));

    ommxDeviceBakeResult sharded = 0, single = 0;
    OMM(ommxShardedBakeRccl(baker, &in, comm, &sharded));
    OMM(ommxDestroyDeviceBakeResult(sharded));
    OMM(ommxShardedBakeRccl(baker, &in, comm, &sharded));      // (the second bake re-uses the baker's pooled working set: no hipMalloc)
    OMM(ommxBakeDevice(baker, &in, &single));
    const ommCpuBakeResultDesc *a = nullptr, *b = nullptr;
    OMM(ommxGetDeviceBakeResultDesc(sharded, &a)); OMM(ommxGetDeviceBakeResultDesc(single, &b));
    int bad = 0;
    bad |= a->arrayDataSize != b->arrayDataSize || a->descArrayCount != b->descArrayCount || a->indexCount != b->indexCount || a->indexFormat != b->indexFormat;
    bad |= a->descArrayHistogramCount != b->descArrayHistogramCount || a->indexHistogramCount != b->indexHistogramCount;
    if (!bad) {
        std::vector<uint8_t> x, y;
        const size_t isz = a->indexFormat == ommIndexFormat_UINT_8 ? 1 : (a->indexFormat == ommIndexFormat_UINT_16 ? 2 : 4);
        HIP((hipError_t)fetch(a->arrayData, a->arrayDataSize, x)); HIP((hipError_t)fetch(b->arrayData, b->arrayDataSize, y)); bad |= x != y;
        HIP((hipError_t)fetch(a->descArray, 8ull * a->descArrayCount, x)); HIP((hipError_t)fetch(b->descArray, 8ull * b->descArrayCount, y)); bad |= x != y;
        HIP((hipError_t)fetch(a->indexBuffer, isz * a->indexCount, x)); HIP((hipError_t)fetch(b->indexBuffer, isz * b->indexCount, y)); bad |= x != y;
        bad |= memcmp(a->descArrayHistogram, b->descArrayHistogram, 8ull * a->descArrayHistogramCount) != 0;
        bad |= memcmp(a->indexHistogram, b->indexHistogram, 8ull * a->indexHistogramCount) != 0;
    }
    printf("rank %u/%u: %u OMMs, %u bytes of arrayData, sharded %s single-GPU\n", rank, world, a->descArrayCount, a->arrayDataSize, bad ? "DIFFERS FROM" : "==");
    OMM(ommxDestroyDeviceBakeResult(sharded)); OMM(ommxDestroyDeviceBakeResult(single));
    OMM(ommxRcclCommDestroy(comm));
    OMM(ommCpuDestroyTexture(baker, tex)); OMM(ommDestroyBaker(baker));
    (void)hipFree(dUv); (void)hipFree(dIdx); (void)hipFree(dLvl);
    return bad ? 1 : 0;
}
