// The one-call sharded bake over a transport of the CALLER (ommxCommFromCollectives): here the "ranks" are threads of one process on one GPU and the two
// collectives are a barrier plus device-to-device copies -- the smallest complete implementation of the ommxCollectives contract.  An MPI or socket
// transport has the same shape: synchronise the stream you are given, exchange, return 0.
//
//   hipcc -std=c++17 -Iinclude examples/sharded_threads.cpp -o /tmp/sharded_threads -Lomm_amd/lib -lomm-lib -Wl,-rpath,$PWD/omm_amd/lib -lpthread
//   /tmp/sharded_threads [ranks = 4]
//
// Every rank bakes the same mesh with its own baker; the library splits the active work items over the ranks, merges their metadata through the
// all-reduce and their blocks (as codec streams) through the all-gather, and every rank ends with the complete result, which the program compares
// byte for byte with a single-GPU ommxBakeDevice of the same desc.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "omm_mi355x.h"
#include "omm_mi355x_ext.h"

#define HIP(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP failure: %s\n", #x); return 10; } } while (0)
#define OMM(x) do { const ommResult r_ = (x); if (r_ != ommResult_SUCCESS) { fprintf(stderr, "%s -> %d\n", #x, (int)r_); return 11; } } while (0)

static void log_cb(ommMessageSeverity severity, const char* message, void*) { fprintf(stderr, "[omm %d] %s\n", (int)severity, message); }
static uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
static float unit(uint32_t x) { return (float)(hash32(x) >> 8) * (1.f / 16777216.f); }

// ---- the transport: a reusable barrier and one slot per rank for its send pointer ----
struct World {
    int size = 1; std::mutex mu; std::condition_variable cv; int waiting = 0; unsigned generation = 0;
    std::vector<const void*> send;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned gen = generation;
        if (++waiting == size) { waiting = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
struct Rank { World* world; int rank; };

static int all_reduce_u32(void* user, const void* send, void* recv, size_t count, ommxReduceOp op, void* hipStream)
{
    Rank* me = (Rank*)user; World& w = *me->world; hipStream_t stream = (hipStream_t)hipStream;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;              // `send` is complete
    w.send[me->rank] = send;
    w.barrier();
    std::vector<uint32_t> acc(count), part(count);
    for (int r = 0; r < w.size; ++r) {
        if (hipMemcpy(part.data(), w.send[r], count * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        for (size_t i = 0; i < count; ++i)
            acc[i] = r == 0 ? part[i] : (op == ommxReduceOp_Sum ? acc[i] + part[i] : (op == ommxReduceOp_Max ? (acc[i] > part[i] ? acc[i] : part[i]) : (acc[i] < part[i] ? acc[i] : part[i])));
    }
    w.barrier();                                                            // everybody has read every `send` (recv may be the same buffer)
    if (hipMemcpyAsync(recv, acc.data(), count * 4, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return 3;
    return 0;
}
static int all_gather_bytes(void* user, const void* send, void* recv, size_t bytesPerRank, void* hipStream)
{
    Rank* me = (Rank*)user; World& w = *me->world; hipStream_t stream = (hipStream_t)hipStream;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    w.send[me->rank] = send;
    w.barrier();
    for (int r = 0; r < w.size; ++r)
        if (hipMemcpyAsync((uint8_t*)recv + (size_t)r * bytesPerRank, w.send[r], bytesPerRank, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 2;
    if (hipStreamSynchronize(stream) != hipSuccess) return 3;
    w.barrier();                                                            // nobody reuses its `send` before all copies are done
    return 0;
}

static int fetch(const void* dev, size_t bytes, std::vector<uint8_t>& out) { out.resize(bytes); return bytes ? (int)hipMemcpy(out.data(), dev, bytes, hipMemcpyDeviceToHost) : 0; }

static int run_rank(World* world, int rank, const std::vector<uint8_t>* alpha, int W, int H, const std::vector<float>* uv, const std::vector<uint32_t>* idx,
                    const std::vector<uint8_t>* lvl)
{
    HIP(hipSetDevice(0));
    const uint32_t T = (uint32_t)lvl->size();
    ommBakerCreationDesc bd; memset(&bd, 0, sizeof bd);
    bd.type = ommBakerType_CPU; bd.messageInterface.messageCallback = log_cb;
    ommBaker baker = 0; OMM(ommCreateBaker(&bd, &baker));
    ommCpuTextureMipDesc mip; memset(&mip, 0, sizeof mip); mip.width = (uint32_t)W; mip.height = (uint32_t)H; mip.textureData = alpha->data();
    ommCpuTextureDesc td; memset(&td, 0, sizeof td); td.format = ommCpuTextureFormat_UNORM8; td.mips = &mip; td.mipCount = 1; td.alphaCutoff = 0.5f;
    ommCpuTexture tex = 0; OMM(ommCpuCreateTexture(baker, &td, &tex));
    void *dUv = nullptr, *dIdx = nullptr, *dLvl = nullptr;
    HIP(hipMalloc(&dUv, uv->size() * 4)); HIP(hipMalloc(&dIdx, idx->size() * 4)); HIP(hipMalloc(&dLvl, lvl->size()));
    HIP(hipMemcpy(dUv, uv->data(), uv->size() * 4, hipMemcpyHostToDevice)); HIP(hipMemcpy(dIdx, idx->data(), idx->size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(dLvl, lvl->data(), lvl->size(), hipMemcpyHostToDevice));

    ommCpuBakeInputDesc in; memset(&in, 0, sizeof in);
    in.bakeFlags = ommCpuBakeFlags_EnableInternalThreads; in.texture = tex;
    in.runtimeSamplerDesc.addressingMode = ommTextureAddressMode_Wrap; in.runtimeSamplerDesc.filter = ommTextureFilterMode_Linear;
    in.alphaMode = ommAlphaMode_Test;
    in.texCoordFormat = ommTexCoordFormat_UV32_FLOAT; in.texCoords = dUv; in.texCoordStrideInBytes = 8;     // DEVICE pointers (ommxBakeDevice contract)
    in.indexFormat = ommIndexFormat_UINT_32; in.indexBuffer = dIdx; in.indexCount = T * 3; in.subdivisionLevels = (const uint8_t*)dLvl;
    in.alphaCutoff = 0.5f; in.alphaCutoffLessEqual = ommOpacityState_Transparent; in.alphaCutoffGreater = ommOpacityState_Opaque;
    in.format = ommFormat_OC1_4_State; in.unknownStatePromotion = ommUnknownStatePromotion_ForceOpaque; in.unresolvedTriState = ommSpecialIndex_FullyUnknownOpaque;
    in.maxSubdivisionLevel = 7; in.maxArrayDataSize = 0xFFFFFFFFu; in.maxWorkloadSize = 0xFFFFFFFFFFFFFFFFull; in.nearDuplicateDeduplicationFactor = 0.15f;

    Rank me{ world, rank };
    ommxCollectives table; table.allReduceU32 = all_reduce_u32; table.allGatherBytes = all_gather_bytes; table.user = &me;
    ommxRcclComm comm = 0; OMM(ommxCommFromCollectives(&table, (uint32_t)rank, (uint32_t)world->size, &comm));

    ommxDeviceBakeResult sharded = 0, single = 0;
    OMM(ommxShardedBakeRccl(baker, &in, comm, &sharded));
    OMM(ommxDestroyDeviceBakeResult(sharded));
    OMM(ommxShardedBakeRccl(baker, &in, comm, &sharded));      // (the second bake re-uses the baker's pooled working set)
    world->barrier();                                           // (the single-GPU bakes below are not collective: keep them out of the others' exchanges)
    OMM(ommxBakeDevice(baker, &in, &single));
    const ommCpuBakeResultDesc *a = nullptr, *b = nullptr;
    OMM(ommxGetDeviceBakeResultDesc(sharded, &a)); OMM(ommxGetDeviceBakeResultDesc(single, &b));
    int bad = 0;
    bad |= a->arrayDataSize != b->arrayDataSize || a->descArrayCount != b->descArrayCount || a->indexCount != b->indexCount || a->indexFormat != b->indexFormat;
    if (!bad) {
        std::vector<uint8_t> x, y;
        const size_t isz = a->indexFormat == ommIndexFormat_UINT_8 ? 1 : (a->indexFormat == ommIndexFormat_UINT_16 ? 2 : 4);
        HIP((hipError_t)fetch(a->arrayData, a->arrayDataSize, x)); HIP((hipError_t)fetch(b->arrayData, b->arrayDataSize, y)); bad |= x != y;
        HIP((hipError_t)fetch(a->descArray, 8ull * a->descArrayCount, x)); HIP((hipError_t)fetch(b->descArray, 8ull * b->descArrayCount, y)); bad |= x != y;
        HIP((hipError_t)fetch(a->indexBuffer, isz * a->indexCount, x)); HIP((hipError_t)fetch(b->indexBuffer, isz * b->indexCount, y)); bad |= x != y;
    }
    printf("rank %d/%d: %u OMMs, %u bytes of arrayData, sharded %s single-GPU\n", rank, world->size, a->descArrayCount, a->arrayDataSize, bad ? "DIFFERS FROM" : "==");
    OMM(ommxDestroyDeviceBakeResult(sharded)); OMM(ommxDestroyDeviceBakeResult(single));
    OMM(ommxRcclCommDestroy(comm));
    OMM(ommCpuDestroyTexture(baker, tex)); OMM(ommDestroyBaker(baker));
    (void)hipFree(dUv); (void)hipFree(dIdx); (void)hipFree(dLvl);
    return bad ? 1 : 0;
}

int main(int argc, char** argv)
{
    const int ranks = argc > 1 ? atoi(argv[1]) : 4;
    if (ranks < 1 || ranks > 64) return 2;
    // alpha texture: soft discs on a 512^2 grid; triangles: random small ones, levels 3..7, a few duplicates
    const int W = 512, H = 512;
    std::vector<uint8_t> alpha((size_t)W * H);
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            const float fx = fmodf((float)i, 64.f) - 32.f, fy = fmodf((float)j, 64.f) - 32.f;
            const float d = sqrtf(fx * fx + fy * fy);
            alpha[(size_t)i + (size_t)j * W] = (uint8_t)(d < 18.f ? 255 : (d < 24.f ? (int)(255.f * (24.f - d) / 6.f) : 0));
        }
    const uint32_t T = 6000;
    std::vector<float> uv((size_t)T * 6); std::vector<uint32_t> idx((size_t)T * 3); std::vector<uint8_t> lvl(T);
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t s = (t % 17 == 5) ? t - 5 : t;
        const float cx = unit(s * 8 + 1), cy = unit(s * 8 + 2);
        for (int k = 0; k < 3; ++k) { uv[(size_t)t * 6 + 2 * k] = cx + 0.03f * (unit(s * 8 + 3 + k) - 0.5f); uv[(size_t)t * 6 + 2 * k + 1] = cy + 0.03f * (unit(s * 8 + 6 + k) - 0.5f); idx[(size_t)t * 3 + k] = t * 3 + k; }
        lvl[t] = (uint8_t)(3 + hash32(s + 99) % 5);
    }
    World world; world.size = ranks; world.send.resize((size_t)ranks, nullptr);
    std::vector<int> rc((size_t)ranks, -1);
    std::vector<std::thread> threads;
    for (int r = 0; r < ranks; ++r) threads.emplace_back([&, r] { rc[(size_t)r] = run_rank(&world, r, &alpha, W, H, &uv, &idx, &lvl); });
    for (auto& t : threads) t.join();
    int bad = 0;
    for (int r = 0; r < ranks; ++r) bad |= rc[(size_t)r];
    printf("%d ranks as threads on one GPU: %s\n", ranks, bad ? "FAILED" : "all == single-GPU");
    return bad ? 1 : 0;
}
