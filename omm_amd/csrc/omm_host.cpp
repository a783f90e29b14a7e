// omm_host.cpp -- the C ABI (include/omm_mi355x.h) and the host orchestration of a bake.
//
// Host code is C++ like the reference's (libraries/omm-lib/src/bake.cpp, bake_cpu_impl.cpp); it only
// validates, builds the work-item list and drives the HIP kernels.  There is NO CPU classification
// path here: without a working HIP device every bake returns ommResult_FAILURE with a Fatal log line.
#include "../../include/omm_mi355x.h"
#include "../../include/omm_mi355x_ext.h"
#include "bake_types.h"
#include "bake_kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>
#include <xmmintrin.h>
#include <emmintrin.h>

using namespace ommx;

namespace {

// ---- handle tagging (src/omm_handle.h:17-54) ----
enum HandleType : uintptr_t { kGpuBaker = 1, kCpuBaker = 3, kTexture = 4 };
template <class T> T* untag(const void* h) { return reinterpret_cast<T*>((uintptr_t)h & ~(uintptr_t)7); }
inline uintptr_t tag_of(const void* h) { return (uintptr_t)h & 7; }

// ---- allocator plumbing (src/std_allocator.h:45-117) ----
void* default_alloc(void*, size_t size, size_t alignment)
{
    void* p = nullptr;
    if (alignment < sizeof(void*)) alignment = sizeof(void*);
    if (posix_memalign(&p, alignment, size ? size : 1) != 0) return nullptr;
    return p;
}
void* default_realloc(void* u, void* mem, size_t size, size_t alignment)
{
    void* n = default_alloc(u, size, alignment);
    if (n && mem) { memcpy(n, mem, size); free(mem); } // conservative: size of the old block is unknown
    return n;
}
void default_free(void*, void* mem) { free(mem); }

struct Allocator {
    ommAllocate alloc = default_alloc; ommReallocate realloc_ = default_realloc; ommFree free_ = default_free; void* user = nullptr;
    void* allocate(size_t bytes, size_t align = 16) const { return alloc(user, bytes ? bytes : 1, align); }
    void release(void* p) const { if (p) free_(user, p); }
    template <class T, class... A> T* make(A&&... a) const { void* p = allocate(sizeof(T), alignof(T) < 16 ? 16 : alignof(T)); return p ? new (p) T(static_cast<A&&>(a)...) : nullptr; }
    template <class T> void destroy(T* p) const { if (p) { p->~T(); release(p); } }
};

// ---- logger (src/log.h:33-140): invalid arguments are reported at Fatal severity ----
struct Logger {
    ommMessageInterface iface = { nullptr, nullptr };
    bool has() const { return iface.messageCallback != nullptr; }
    void msg(ommMessageSeverity s, const char* m) const { if (iface.messageCallback) iface.messageCallback(s, m, iface.userArg); }
    ommResult invalid(const char* m) const { msg(ommMessageSeverity_Fatal, m); return ommResult_INVALID_ARGUMENT; }
    ommResult failure(const char* m) const { msg(ommMessageSeverity_Fatal, m); return ommResult_FAILURE; }
};

// ---- device arena: one grow-only HBM block per baker, reused across bakes ----
struct DeviceArena {
    std::mutex mu; uint8_t* base = nullptr; size_t cap = 0, used = 0;
    ~DeviceArena() { if (base) (void)hipFree(base); }
    bool reserve(size_t bytes) {
        if (bytes <= cap) { used = 0; return true; }
        if (base) { (void)hipFree(base); base = nullptr; cap = 0; }
        if (hipMalloc((void**)&base, bytes) != hipSuccess) { base = nullptr; (void)hipGetLastError(); return false; }
        cap = bytes; used = 0; return true;
    }
    template <class T> T* take(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        T* p = (T*)(base + used); used += bytes; return p;
    }
};
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Baker {
    Allocator mem; Logger log; ommBakerType type;
    DeviceArena arena;        // per-item tables + scratch
    DeviceArena statesArena;  // packed states of the active (non-uniform) items; guarded by arena.mu
    std::mutex timingsMu; ommxBakeTimings timings; bool haveTimings = false;
};

// HIP events on the bake's own stream (torch / the caller never see this stream)
struct EventTimer {
    hipStream_t s; hipEvent_t ev[16]; int n = 0;
    explicit EventTimer(hipStream_t st) : s(st) { for (auto& e : ev) e = nullptr; }
    ~EventTimer() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); }
    int mark() { if (n >= 16) return -1; if (hipEventCreate(&ev[n]) != hipSuccess) return -1; (void)hipEventRecord(ev[n], s); return n++; }
    float ms(int a, int b) const { float v = 0.f; if (a < 0 || b < 0 || hipEventElapsedTime(&v, ev[a], ev[b]) != hipSuccess) return 0.f; return v; }
};
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct TexMip { int w = 0, h = 0; void* texels = nullptr; uint32_t* sat = nullptr; };
struct Texture {
    Allocator mem; const Logger* log = nullptr;
    ommCpuTextureFormat format = ommCpuTextureFormat_MAX_NUM; ommCpuTextureFlags flags = ommCpuTextureFlags_None; float alphaCutoff = -1.f;
    std::vector<TexMip> mips;
    ~Texture() { for (auto& m : mips) { if (m.texels) (void)hipFree(m.texels); if (m.sat) (void)hipFree(m.sat); } }
};

struct BakeResult {
    Allocator mem;
    void* arrayData = nullptr; ommCpuOpacityMicromapDesc* descs = nullptr;
    ommCpuOpacityMicromapUsageCount* arrayHist = nullptr; ommCpuOpacityMicromapUsageCount* indexHist = nullptr;
    int32_t* index = nullptr;
    ommCpuBakeResultDesc desc;
    BakeResult() { memset(&desc, 0, sizeof desc); }
    ~BakeResult() { mem.release(arrayData); mem.release(descs); mem.release(arrayHist); mem.release(indexHist); mem.release(index); }
};

// ---- XXH64 of a constant byte stream: digests of uniform OMMs (bake_cpu_impl.cpp:1038-1040 applied to 4^level equal bytes) ----
struct UniformDigests {
    uint64_t v[kNumLevels * 4];
    UniformDigests() {
        const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
        auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
        auto round = [&](uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; };
        auto merge = [&](uint64_t h, uint64_t val) { h ^= round(0, val); return h * P1 + P4; };
        for (int l = 0; l < kNumLevels; ++l)
            for (int st = 0; st < 4; ++st) {
                const uint64_t len = (uint64_t)1 << (2 * l), seed = 42;
                const uint64_t w8 = 0x0101010101010101ULL * (uint64_t)st; const uint32_t w4 = 0x01010101u * (uint32_t)st;
                uint64_t h, rem = len;
                if (len >= 32) {
                    uint64_t a = seed + P1 + P2, b = seed + P2, c = seed, d = seed - P1;
                    for (uint64_t k = 0; k < len / 32; ++k) { a = round(a, w8); b = round(b, w8); c = round(c, w8); d = round(d, w8); }
                    h = rotl(a, 1) + rotl(b, 7) + rotl(c, 12) + rotl(d, 18);
                    h = merge(h, a); h = merge(h, b); h = merge(h, c); h = merge(h, d);
                    rem = 0; // 4^l is a multiple of 32 from level 3 on
                } else h = seed + P5;
                h += len;
                for (; rem >= 8; rem -= 8) { h ^= round(0, w8); h = rotl(h, 27) * P1 + P4; }
                if (rem >= 4) { h ^= (uint64_t)w4 * P1; h = rotl(h, 23) * P2 + P3; rem -= 4; }
                for (; rem > 0; --rem) { h ^= (uint64_t)st * P5; h = rotl(h, 11) * P1; }
                h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
                v[l * 4 + st] = h;
            }
    }
};
const UniformDigests& uniform_digests() { static const UniformDigests t; return t; }

// ---- x86 conversion semantics used by the reference's host-side arithmetic ----
inline int f2i(float f) { return _mm_cvtt_ss2si(_mm_set_ss(f)); }
inline uint32_t f2u(float f) { return (uint32_t)_mm_cvttss_si64(_mm_set_ss(f)); }

struct HostTri { float p[6]; };

float half_to_float(uint16_t h) // glm::unpackHalf2x16 element
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { e = 1; while (!(m & 0x400u)) { m <<= 1; e--; } m &= 0x3ffu; bits = sign | ((e + 112u) << 23) | (m << 13); }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

// util/geometry.h:191-239 + bake_cpu_impl.cpp:579-587
HostTri fetch_triangle(const ommCpuBakeInputDesc& d, uint32_t prim)
{
    uint32_t stride = d.texCoordStrideInBytes;
    if (stride == 0) stride = d.texCoordFormat == ommTexCoordFormat_UV32_FLOAT ? 8u : 4u;
    uint32_t idx[3];
    const size_t o = 3ull * prim;
    for (int k = 0; k < 3; ++k) {
        if (d.indexFormat == ommIndexFormat_UINT_8) idx[k] = ((const uint8_t*)d.indexBuffer)[o + k];
        else if (d.indexFormat == ommIndexFormat_UINT_16) idx[k] = ((const uint16_t*)d.indexBuffer)[o + k];
        else idx[k] = ((const uint32_t*)d.indexBuffer)[o + k];
    }
    HostTri t;
    for (int k = 0; k < 3; ++k) {
        const uint8_t* base = (const uint8_t*)d.texCoords + (size_t)stride * idx[k];
        if (d.texCoordFormat == ommTexCoordFormat_UV32_FLOAT) { memcpy(&t.p[2 * k], base, 8); }
        else {
            uint32_t v; memcpy(&v, base, 4);
            if (d.texCoordFormat == ommTexCoordFormat_UV16_UNORM) {
                t.p[2 * k] = (float)(v & 0xffffu) * 1.5259021896696421759314870504694e-5f;
                t.p[2 * k + 1] = (float)(v >> 16) * 1.5259021896696421759314870504694e-5f;
            } else if (d.texCoordFormat == ommTexCoordFormat_UV16_FLOAT) {
                t.p[2 * k] = half_to_float((uint16_t)(v & 0xffffu)); t.p[2 * k + 1] = half_to_float((uint16_t)(v >> 16));
            } else { t.p[2 * k] = 0; t.p[2 * k + 1] = 0; }
        }
    }
    return t;
}

bool tri_invalid(const HostTri& t) { for (float v : t.p) if (std::isnan(v) || std::isinf(v)) return true; return false; }
bool tri_degenerate(const HostTri& t) // util/geometry.h:44-47
{
    const float* p = t.p;
    const float area = 0.5f * fabsf(p[0] * (p[3] - p[5]) + p[2] * (p[5] - p[1]) + p[4] * (p[1] - p[3]));
    return (double)area < 1e-9;
}
float area2d(float ax, float ay, float bx, float by, float cx, float cy) // util/geometry.h:141-145
{
    const float v0x = cx - ax, v0y = cy - ay, v1x = bx - ax, v1y = by - ay;
    const float nx = v0y * 0.f - v1y * 0.f, ny = 0.f * v1x - 0.f * v0x, nz = v0x * v1y - v1x * v0y;
    return 0.5f * sqrtf(nx * nx + ny * ny + nz * nz);
}

// bake_cpu_impl.cpp:470-560
int32_t level_for_primitive(const ommCpuBakeInputDesc& d, uint32_t flags, uint32_t i, const HostTri& t, int w, int h)
{
    if (d.subdivisionLevels && d.subdivisionLevels[i] <= 12) return d.subdivisionLevels[i];
    if (!(d.dynamicSubdivisionScale > 0)) return d.maxSubdivisionLevel;
    const float fw = (float)(uint32_t)w, fh = (float)(uint32_t)h;
    const float* p = t.p;
    if (tri_degenerate(t) || (flags & (1u << 11))) { // edge heuristic (glibc log2f, stays on the host)
        const float e0x = fw * (p[2] - p[0]), e0y = fh * (p[3] - p[1]);
        const float e1x = fw * (p[4] - p[0]), e1y = fh * (p[5] - p[1]);
        const float e2x = fw * (p[4] - p[2]), e2y = fh * (p[5] - p[3]);
        const float l0 = e0x * e0x + e0y * e0y, l1 = e1x * e1x + e1y * e1y, l2 = e2x * e2x + e2y * e2y;
        float eMax = l0; if (eMax < l1) eMax = l1; if (eMax < l2) eMax = l2;
        const float n = (double)eMax < 1e-6 ? 0 : log2f(eMax) / 2.f - log2f(d.dynamicSubdivisionScale);
        int lvl = f2i(ceilf(n));
        if (lvl < 0) lvl = 0; if (lvl > (int)d.maxSubdivisionLevel) lvl = d.maxSubdivisionLevel;
        return lvl;
    }
    const float area = area2d(p[0] * fw, p[1] * fh, p[2] * fw, p[3] * fh, p[4] * fw, p[5] * fh);
    const float target = d.dynamicSubdivisionScale * d.dynamicSubdivisionScale;
    uint32_t v = f2u(area / target);
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
    static const uint32_t bm[5] = { 0xAAAAAAAAu, 0xCCCCCCCCu, 0xF0F0F0F0u, 0xFF00FF00u, 0xFFFF0000u };
    uint32_t r = (v & bm[0]) != 0;
    for (uint32_t k = 4; k > 0; k--) r |= (uint32_t)((v & bm[k]) != 0) << k;
    const uint32_t lvl = r >> 1;
    return (int32_t)(lvl < d.maxSubdivisionLevel ? lvl : d.maxSubdivisionLevel);
}

// UV-dedup key: the reference keys its map by a 64-bit hash chain over (p0,p1,p2,level,format) and
// trusts it (bake_cpu_impl.cpp:626-649); modulo 2^-64 collisions that is equality of the tuple with
// +0 == -0 (std::hash<float>).  The tuple itself is the key here.
struct UvKey { uint32_t k[8]; bool operator==(const UvKey& o) const { return memcmp(k, o.k, sizeof k) == 0; } };
struct UvKeyHash {
    size_t operator()(const UvKey& a) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 8; ++i) { h ^= a.k[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
        return (size_t)h;
    }
};

const char* special_name(int s)
{
    switch (s) { case -1: return "Fully Transparent"; case -2: return "Fully Opaque"; case -3: return "Fully Unknown Transparent"; case -4: return "Fully Unknown Opaque"; default: return "Unknown State"; }
}
const char* state_name(int s) { switch (s) { case 0: return "Transparent"; case 1: return "Opaque"; case 2: return "UnknownTransparent"; case 3: return "UnknownOpaque"; default: return "Unknown"; } }
const char* format_name(int f) { return f == 1 ? "OC1_2_State" : (f == 2 ? "OC1_4_State" : "Unknown"); }
bool compatible(int state, int format) { return format == ommFormat_OC1_2_State ? (state == 0 || state == 1) : true; }

// bake_cpu_impl.cpp:235-290 -- message strings are pinned by support/tests/test_omm_log.cpp:146-209
ommResult validate_desc(const Baker& b, const ommCpuBakeInputDesc& d)
{
    const Logger& L = b.log; const uint32_t flags = (uint32_t)d.bakeFlags; char buf[256];
    if (d.texture == 0) return L.invalid("[Invalid Argument] - texture is not set");
    if (tag_of(d.texture) != kTexture) return L.invalid("[Invalid Argument] - desc.texture is of incorrect type");
    if (d.alphaMode == ommAlphaMode_MAX_NUM) return L.invalid("[Invalid Argument] - alphaMode is not set");
    if (d.runtimeSamplerDesc.addressingMode == ommTextureAddressMode_MAX_NUM) return L.invalid("[Invalid Argument] - runtimeSamplerDesc.addressingMode is not set");
    if (d.runtimeSamplerDesc.filter == ommTextureFilterMode_MAX_NUM) return L.invalid("[Invalid Argument] - runtimeSamplerDesc.filter is not set");
    if (d.texCoordFormat == ommTexCoordFormat_MAX_NUM) return L.invalid("[Invalid Argument] - texCoordFormat is not set");
    if (d.texCoords == nullptr) return L.invalid("[Invalid Argument] - texCoords is not set");
    if (d.indexFormat == ommIndexFormat_MAX_NUM) return L.invalid("[Invalid Argument] - indexFormat is not set");
    if (d.indexBuffer == nullptr) return L.invalid("[Invalid Argument] - indexBuffer is not set");
    if (d.indexCount == 0) return L.invalid("[Invalid Argument] - indexCount is not set");
    if (d.maxSubdivisionLevel > kMaxLevel) {
        snprintf(buf, sizeof buf, "[Invalid Argument] - maxSubdivisionLevel (%d) is greater than maximum supported (%d)", d.maxSubdivisionLevel, kMaxLevel);
        return L.invalid(buf);
    }
    if ((flags & ((1u << 4) | (1u << 10))) && (flags & (1u << 3)))
        return L.invalid("[Invalid Argument] - EnableNearDuplicateDetection or EnableNearDuplicateDetectionBruteForce is used together with DisableDuplicateDetection");
    if ((flags & (1u << 5)) && !L.has())
        return L.invalid("[Invalid Argument] - EnableValidation is set but no message callback was provided");
    const Texture* tex = untag<Texture>(d.texture);
    if (tex->alphaCutoff >= 0.f && tex->alphaCutoff != d.alphaCutoff) {
        snprintf(buf, sizeof buf, "[Invalid Argument] - Texture object alpha cutoff threshold (%.6f) is different from alpha cutoff threshold in bake input (%.6f)", tex->alphaCutoff, d.alphaCutoff);
        return L.invalid(buf);
    }
    if (!compatible(d.alphaCutoffGreater, d.format)) {
        snprintf(buf, sizeof buf, "[Invalid Argument] - alphaCutoffGreater=%s is not compatible with %s", state_name(d.alphaCutoffGreater), format_name(d.format));
        return L.invalid(buf);
    }
    if (!compatible(d.alphaCutoffLessEqual, d.format)) {
        snprintf(buf, sizeof buf, "[Invalid Argument] - alphaCutoffLessEqual=%s is not compatible with %s", state_name(d.alphaCutoffLessEqual), format_name(d.format));
        return L.invalid(buf);
    }
    return ommResult_SUCCESS;
}

uint32_t ctz32(uint32_t n) { if (!n) return 32; uint32_t c = 0; while (!(n & 1)) { c++; n >>= 1; } return c; }
bool is_pow2(int x) { return x > 0 && !(x & (x - 1)); }

#define HIP_OK(call) ((call) == hipSuccess)

// The bake proper: bake_cpu_impl.cpp:1923-1985 re-organised for the device.
ommResult bake_impl(Baker& baker, const ommCpuBakeInputDesc& d, ommCpuBakeResult* out)
{
    const Logger& L = baker.log;
    const uint32_t flags = (uint32_t)d.bakeFlags;
    const Texture& tex = *untag<Texture>(d.texture);
    const double t0 = now_ms();

    // ---- scope fences (documented in DESIGN.md) ----
    if ((flags & ((1u << 4) | (1u << 10))) != 0)
        { L.msg(ommMessageSeverity_Fatal, "[Not Implemented] - near-duplicate merging (EnableNearDuplicateDetection) is not available in the MI355X baker yet"); return ommResult_NOT_IMPLEMENTED; }
    if (d.maxArrayDataSize != 0xFFFFFFFFu)
        { L.msg(ommMessageSeverity_Fatal, "[Not Implemented] - maxArrayDataSize budgets are not available in the MI355X baker yet"); return ommResult_NOT_IMPLEMENTED; }
    if ((flags & ((1u << 7) | (1u << 8) | (1u << 9) | (1u << 11))) != 0)
        { L.msg(ommMessageSeverity_Fatal, "[Not Implemented] - internal bake flags (bits 7-11) are not supported"); return ommResult_NOT_IMPLEMENTED; }
    const uint32_t triCount = d.indexCount / 3u;
    if (d.formats) // the reference sizes its arrays from the global format only (bake_cpu_impl.cpp:1763-1772): mixed formats corrupt its heap
        for (uint32_t i = 0; i < triCount; ++i)
            if (d.formats[i] != ommFormat_INVALID && d.formats[i] != d.format)
                return L.failure("[Failure] - per-triangle formats that differ from the global format are not supported");

    // ---- SetupWorkItems (bake_cpu_impl.cpp:589-660), host side ----
    std::vector<HostTri> itemUv; std::vector<uint8_t> itemLevel, itemDegenerate; std::vector<int32_t> triToItem(triCount ? triCount : 1, -1);
    {
        std::unordered_map<UvKey, uint32_t, UvKeyHash> seen;
        seen.reserve((size_t)triCount * 2);
        itemUv.reserve(triCount); itemLevel.reserve(triCount); itemDegenerate.reserve(triCount);
        uint32_t numDisabled = 0;
        const bool noDedup = (flags & (1u << 3)) != 0;
        for (uint32_t i = 0; i < triCount; ++i) {
            const HostTri t = fetch_triangle(d, i);
            const int32_t lvl = level_for_primitive(d, flags, i, t, tex.mips[0].w, tex.mips[0].h);
            if (lvl == 0xE || tri_invalid(t)) { numDisabled++; continue; }
            UvKey key;
            for (int k = 0; k < 6; ++k) { const float f = t.p[k] == 0.f ? 0.f : t.p[k]; memcpy(&key.k[k], &f, 4); }
            key.k[6] = (uint32_t)lvl; key.k[7] = (uint32_t)d.format;
            auto it = noDedup ? seen.end() : seen.find(key);
            if (it == seen.end()) {
                if (lvl > kMaxLevel) return L.invalid("[Invalid Argument] - subdivisionLevel for primitive (i) is (d) which exceeds kMaxSubdivLevel(12)");
                const uint32_t id = (uint32_t)itemUv.size();
                if (!noDedup) seen.emplace(key, id);
                itemUv.push_back(t); itemLevel.push_back((uint8_t)lvl); itemDegenerate.push_back(tri_degenerate(t) ? 1 : 0);
                triToItem[i] = (int32_t)id;
            } else triToItem[i] = (int32_t)it->second;
        }
        if ((flags & (1u << 5)) && numDisabled != 0) {
            char buf[256];
            snprintf(buf, sizeof buf, "[Info] - The workload consists of %d unclassifiable triangles, these will be classified as unresolvedTriState = %s.", numDisabled, special_name(d.unresolvedTriState));
            L.msg(ommMessageSeverity_Info, buf);
        }
    }
    const uint32_t U = (uint32_t)itemUv.size();

    // ---- ValidateWorkloadSize (bake_cpu_impl.cpp:662-713) ----
    {
        const bool limit = d.maxWorkloadSize != 0xFFFFFFFFFFFFFFFFull;
        if ((flags & (1u << 5)) || limit) {
            const float fw = (float)tex.mips[0].w, fh = (float)tex.mips[0].h;
            uint64_t workload = 0;
            for (uint32_t i = 0; i < U; ++i) {
                const float* p = itemUv[i].p;
                const float lox = std::min(std::min(p[0], p[2]), p[4]), loy = std::min(std::min(p[1], p[3]), p[5]);
                const float hix = std::max(std::max(p[0], p[2]), p[4]), hiy = std::max(std::max(p[1], p[3]), p[5]);
                const int ax = f2i((hix - lox) * fw), ay = f2i((hiy - loy) * fh);
                workload += (uint64_t)(int64_t)(int32_t)((uint32_t)ax * (uint32_t)ay);
            }
            if (limit && workload > d.maxWorkloadSize) return ommResult_WORKLOAD_TOO_BIG;
            if ((flags & (1u << 5)) && workload > (1ull << 27)) {
                char buf[256];
                snprintf(buf, sizeof buf, "[Perf Warning] - The workload consists of %lld work items (number of texels to classify), which corresponds to roughly %lld 1024x1024 textures."
                         " This is unusually large and may result in long bake times.", (long long)workload, (long long)(workload >> 20));
                L.msg(ommMessageSeverity_PerfWarning, buf);
            }
        }
    }

    const double tSetup = now_ms();
    // ---- device layout ----
    const int bits = (int)d.format;
    std::vector<uint32_t> levelCount(kNumLevels, 0), levelStart(kNumLevels + 1, 0);
    for (uint32_t i = 0; i < U; ++i) levelCount[itemLevel[i]]++;
    for (int l = 0; l < kNumLevels; ++l) levelStart[l + 1] = levelStart[l] + levelCount[l];
    std::vector<uint32_t> itemIds(U ? U : 1);
    { std::vector<uint32_t> cur(levelStart.begin(), levelStart.end() - 1); for (uint32_t i = 0; i < U; ++i) itemIds[cur[itemLevel[i]]++] = i; }

    const size_t scratchBytes = tail_scratch_bytes(U, triCount);
    const size_t perItem32 = pad256((size_t)(U ? U : 1) * 4), perItem64 = pad256((size_t)(U ? U : 1) * 8);
    size_t need = pad256((size_t)(U ? U : 1) * 24) /*uv*/ + 2 * pad256(U ? U : 1) /*level,degenerate*/ + perItem64 /*stateOfs*/ + perItem32 /*itemIds*/
                + pad256((size_t)(triCount ? triCount : 1) * 4) * 2 /*triToItem, indexBuffer*/ + perItem32 * 9 /*mask, known, special, rep, order, dstOfs, sizes, itemValue, activeIds*/ + pad256(U ? U : 1) /*active*/
                + perItem64 /*digests*/ + 2048 /*histograms, error flag*/ + pad256(scratchBytes) + 8192;

    std::unique_lock<std::mutex> lock(baker.arena.mu, std::try_to_lock);
    DeviceArena local, localStates; // concurrent bakes on one baker get private arenas
    DeviceArena* arena = lock.owns_lock() ? &baker.arena : &local;
    DeviceArena* statesArena = lock.owns_lock() ? &baker.statesArena : &localStates;
    if (!arena->reserve(need)) return L.failure("[Failure] - out of device memory for the bake working set");

    hipStream_t stream = nullptr;
    if (!HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking))) return L.failure("[Failure] - no usable HIP device (the MI355X baker has no CPU fallback)");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ stream };

    float* dUv = arena->take<float>((size_t)(U ? U : 1) * 6);
    uint8_t* dLevel = arena->take<uint8_t>(U ? U : 1); uint8_t* dDegen = arena->take<uint8_t>(U ? U : 1);
    uint64_t* dStateOfs = arena->take<uint64_t>(U ? U : 1); uint32_t* dItemIds = arena->take<uint32_t>(U ? U : 1);
    int32_t* dTriToItem = arena->take<int32_t>(triCount ? triCount : 1); int32_t* dIndex = arena->take<int32_t>(triCount ? triCount : 1);
    uint32_t* dMask = arena->take<uint32_t>(U ? U : 1); uint32_t* dKnown = arena->take<uint32_t>(U ? U : 1);
    int32_t* dSpecial = arena->take<int32_t>(U ? U : 1); uint32_t* dRep = arena->take<uint32_t>(U ? U : 1);
    uint32_t* dOrder = arena->take<uint32_t>(U ? U : 1); uint32_t* dDstOfs = arena->take<uint32_t>(U ? U : 1);
    uint32_t* dSizes = arena->take<uint32_t>(U ? U : 1); int32_t* dItemValue = arena->take<int32_t>(U ? U : 1);
    uint64_t* dDigests = arena->take<uint64_t>(U ? U : 1);
    uint32_t* dArrayHist = arena->take<uint32_t>(kNumLevels); uint32_t* dIndexHist = arena->take<uint32_t>(kNumLevels);
    uint32_t* dErr = arena->take<uint32_t>(1);
    uint8_t* dScratch = arena->take<uint8_t>(scratchBytes);
    uint32_t* dActiveIds = arena->take<uint32_t>(U ? U : 1); uint8_t* dActive = arena->take<uint8_t>(U ? U : 1);
    uint64_t* dUniformDigest = arena->take<uint64_t>(kNumLevels * 4);

    EventTimer et(stream);
    const int e0 = et.mark();
    bool ok = true;
    if (U) {
        ok &= HIP_OK(hipMemcpyAsync(dUv, itemUv.data(), (size_t)U * 24, hipMemcpyHostToDevice, stream));
        ok &= HIP_OK(hipMemcpyAsync(dLevel, itemLevel.data(), U, hipMemcpyHostToDevice, stream));
        ok &= HIP_OK(hipMemcpyAsync(dDegen, itemDegenerate.data(), U, hipMemcpyHostToDevice, stream));
        ok &= HIP_OK(hipMemcpyAsync(dItemIds, itemIds.data(), (size_t)U * 4, hipMemcpyHostToDevice, stream));
        ok &= HIP_OK(hipMemsetAsync(dKnown, 0, (size_t)U * 4, stream));
    }
    ok &= HIP_OK(hipMemcpyAsync(dUniformDigest, uniform_digests().v, sizeof(uint64_t) * kNumLevels * 4, hipMemcpyHostToDevice, stream));
    if (triCount) ok &= HIP_OK(hipMemcpyAsync(dTriToItem, triToItem.data(), (size_t)triCount * 4, hipMemcpyHostToDevice, stream));
    if (!ok) return L.failure("[Failure] - host to device transfer failed");

    // ---- classification: ResampleCoarse + ResampleFine (bake_cpu_impl.cpp:715-1029) ----
    ClassifyParams P; memset(&P, 0, sizeof P);
    P.mipCount = (int)tex.mips.size();
    for (int m = 0; m < P.mipCount; ++m) {
        DevMip& dm = P.mips[m]; const TexMip& tm = tex.mips[m];
        dm.texels = tm.texels; dm.sat = tm.sat; dm.w = tm.w; dm.h = tm.h;
        dm.log2w = (int)ctz32((uint32_t)tm.w); dm.log2h = (int)ctz32((uint32_t)tm.h);
        dm.pow2 = is_pow2(tm.w) && is_pow2(tm.h);
        dm.fw = (float)tm.w; dm.fh = (float)tm.h; dm.rw = 1.f / (float)tm.w; dm.rh = 1.f / (float)tm.h;
    }
    P.pow2Dispatch = P.mips[0].pow2;
    P.texIsFp32 = tex.format == ommCpuTextureFormat_FP32;
    P.addrMode = d.runtimeSamplerDesc.addressingMode;
    P.filterLinear = d.runtimeSamplerDesc.filter == ommTextureFilterMode_Linear;
    P.format = bits; P.promotion = d.unknownStatePromotion; P.stateGT = d.alphaCutoffGreater; P.stateLE = d.alphaCutoffLessEqual;
    P.useCoarse = tex.mips[0].sat != nullptr && P.mipCount == 1 && P.filterLinear;
    P.cutoff = d.alphaCutoff; P.borderAlpha = d.runtimeSamplerDesc.borderAlpha;
    P.wantKnownCount = d.rejectionThreshold > 0.f;

    const int e1 = et.mark();
    // level-0 hierarchical query per item, then compaction of the items that still need per-micro-triangle work
    launch_triage(P, dUv, U, dMask, dActive, stream);
    uint32_t activeStart[kNumLevels + 1]; uint64_t stateBytes = 0;
    if (!HIP_OK(run_prep(dItemIds, dActive, dLevel, bits, U, levelStart.data(), dActiveIds, dStateOfs, dScratch, scratchBytes, activeStart, &stateBytes, stream)))
        return L.failure("[Failure] - device work-list compaction failed");
    if (!statesArena->reserve(stateBytes ? stateBytes : 256)) return L.failure("[Failure] - out of device memory for the packed micro-triangle states");
    uint8_t* dStates = statesArena->base;
    const int e1b = et.mark();
    ItemArrays A; A.uv = dUv; A.degenerate = dDegen; A.stateOfs = dStateOfs; A.states = dStates; A.stateMask = dMask; A.knownCount = dKnown;
    for (int l = 0; l < kNumLevels; ++l)
        launch_classify(P, A, dActiveIds + activeStart[l], activeStart[l + 1] - activeStart[l], (uint32_t)l, stream);
    const int e2 = et.mark();
    // ---- CalcDigest (bake_cpu_impl.cpp:1038-1040): active items here, uniform ones from the table in the tail ----
    if (!(flags & (1u << 3)))
        for (int l = 0; l < kNumLevels; ++l)
            launch_digest(dStates, dStateOfs, dActiveIds + activeStart[l], activeStart[l + 1] - activeStart[l], (uint32_t)l, (uint32_t)bits, dDigests, stream);
    if (!HIP_OK(hipGetLastError())) return L.failure("[Failure] - kernel launch failed");

    const int e3 = et.mark();
    // ---- promote / dedup / sort / offsets on the device ----
    TailInputs ti; memset(&ti, 0, sizeof ti);
    ti.numItems = U; ti.numTris = triCount; ti.uv = dUv; ti.level = dLevel; ti.stateMask = dMask; ti.knownCount = dKnown; ti.digests = dDigests;
    ti.uniformDigest = dUniformDigest; ti.triToItem = dTriToItem; ti.format = bits;
    ti.disableSpecial = (flags & (1u << 1)) != 0; ti.disableDedup = (flags & (1u << 3)) != 0;
    ti.rejectionThreshold = d.rejectionThreshold; ti.unresolved = (int32_t)d.unresolvedTriState; ti.errorFlag = dErr;
    TailOutputs to; memset(&to, 0, sizeof to);
    to.special = dSpecial; to.rep = dRep; to.order = dOrder; to.dstOfs = dDstOfs; to.sizes = dSizes; to.itemValue = dItemValue;
    to.indexBuffer = dIndex; to.arrayHist = dArrayHist; to.indexHist = dIndexHist;
    TailCounts counts;
    if (!HIP_OK(run_tail(ti, to, dScratch, scratchBytes, &counts, stream))) return L.failure("[Failure] - device tail failed");
    if (counts.arrayDataSize > 0xFFFFFFFFull) return ommResult_FAILURE; // bake_cpu_impl.cpp:1774-1775
    const int e4 = et.mark();

    // ---- Serialize (bake_cpu_impl.cpp:1756-1920): gather on device, copy out through the user's allocator ----
    BakeResult* res = baker.mem.make<BakeResult>();
    if (!res) return ommResult_FAILURE;
    res->mem = baker.mem;
    const uint32_t E = counts.numOmms;
    uint32_t hostHist[2 * kNumLevels];
    ok = true;
    uint8_t* dArray = nullptr; ommCpuOpacityMicromapDesc* dDescs = nullptr;
    int e5 = e4;
    if (E) {
        res->arrayData = baker.mem.allocate((size_t)counts.arrayDataSize, 64);
        res->descs = (ommCpuOpacityMicromapDesc*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapDesc) * (size_t)E, 16);
        ok &= res->arrayData && res->descs;
        ok = ok && HIP_OK(hipMalloc((void**)&dArray, (size_t)counts.arrayDataSize)) && HIP_OK(hipMalloc((void**)&dDescs, sizeof(ommCpuOpacityMicromapDesc) * (size_t)E));
        if (ok) {
            launch_gather_omms(dStates, dStateOfs, dActive, dMask, dLevel, bits, dOrder, dDstOfs, dSizes, E, dArray, stream);
            launch_write_descs(dOrder, dDstOfs, dLevel, bits, E, dDescs, stream);
            e5 = et.mark();
            ok &= HIP_OK(hipMemcpyAsync(res->arrayData, dArray, (size_t)counts.arrayDataSize, hipMemcpyDeviceToHost, stream));
            ok &= HIP_OK(hipMemcpyAsync(res->descs, dDescs, sizeof(ommCpuOpacityMicromapDesc) * (size_t)E, hipMemcpyDeviceToHost, stream));
        }
    }
    res->index = (int32_t*)baker.mem.allocate(sizeof(int32_t) * (size_t)(triCount ? triCount : 1), 16);
    ok &= res->index != nullptr;
    if (ok && triCount) ok &= HIP_OK(hipMemcpyAsync(res->index, dIndex, (size_t)triCount * 4, hipMemcpyDeviceToHost, stream));
    if (ok) ok &= HIP_OK(hipMemcpyAsync(hostHist, dArrayHist, sizeof(uint32_t) * kNumLevels, hipMemcpyDeviceToHost, stream));
    if (ok) ok &= HIP_OK(hipMemcpyAsync(hostHist + kNumLevels, dIndexHist, sizeof(uint32_t) * kNumLevels, hipMemcpyDeviceToHost, stream));
    const int e6 = et.mark();
    if (ok) ok &= HIP_OK(hipStreamSynchronize(stream));
    if (dArray) (void)hipFree(dArray);
    if (dDescs) (void)hipFree(dDescs);
    if (!ok) { baker.mem.destroy(res); return L.failure("[Failure] - device to host transfer of the bake result failed"); }

    // histograms: format {2-state, 4-state} x level ascending, non-zero entries only (:1833-1850); one global format here
    res->arrayHist = (ommCpuOpacityMicromapUsageCount*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels, 16);
    res->indexHist = (ommCpuOpacityMicromapUsageCount*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels, 16);
    uint32_t nAH = 0, nIH = 0;
    for (uint32_t l = 0; l < (uint32_t)kNumLevels; ++l) {
        if (hostHist[l]) { res->arrayHist[nAH].count = hostHist[l]; res->arrayHist[nAH].subdivisionLevel = (uint16_t)l; res->arrayHist[nAH].format = (uint16_t)bits; nAH++; }
        if (hostHist[kNumLevels + l]) { res->indexHist[nIH].count = hostHist[kNumLevels + l]; res->indexHist[nIH].subdivisionLevel = (uint16_t)l; res->indexHist[nIH].format = (uint16_t)bits; nIH++; }
    }
    // index narrowing in place (:1872-1902)
    ommIndexFormat ifmt = ommIndexFormat_UINT_32;
    const bool allow8 = (flags & (1u << 6)) != 0, force32 = (flags & (1u << 2)) != 0;
    if (allow8 && triCount <= 127 && !force32) { int8_t* p8 = (int8_t*)res->index; for (uint32_t i = 0; i < triCount; ++i) { const int32_t v = res->index[i]; p8[i] = (int8_t)v; } ifmt = ommIndexFormat_UINT_8; }
    else if (triCount <= 32767 && !force32) { int16_t* p16 = (int16_t*)res->index; for (uint32_t i = 0; i < triCount; ++i) { const int32_t v = res->index[i]; p16[i] = (int16_t)v; } ifmt = ommIndexFormat_UINT_16; }

    res->desc.arrayData = E ? res->arrayData : nullptr; res->desc.arrayDataSize = E ? (uint32_t)counts.arrayDataSize : 0;
    res->desc.descArray = E ? res->descs : nullptr; res->desc.descArrayCount = E;
    res->desc.descArrayHistogram = res->arrayHist; res->desc.descArrayHistogramCount = nAH;
    res->desc.indexBuffer = res->index; res->desc.indexCount = triCount; res->desc.indexFormat = ifmt;
    res->desc.indexHistogram = res->indexHist; res->desc.indexHistogramCount = nIH;
    {
        ommxBakeTimings tm; memset(&tm, 0, sizeof tm);
        tm.hostSetupMs = (float)(tSetup - t0); tm.uploadMs = et.ms(e0, e1); tm.triageMs = et.ms(e1, e1b); tm.classifyMs = et.ms(e1b, e2); tm.digestMs = et.ms(e2, e3);
        tm.tailMs = et.ms(e3, e4); tm.gatherMs = et.ms(e4, e5); tm.downloadMs = et.ms(e5, e6); tm.totalMs = (float)(now_ms() - t0);
        for (uint32_t i = 0; i < U; ++i) tm.microTriangles += (uint64_t)1 << (2 * itemLevel[i]);
        tm.uniqueItems = U; tm.stateBytes = stateBytes; tm.activeItems = activeStart[kNumLevels];
        for (int l = 0; l < kNumLevels; ++l) tm.classifyLaunches += activeStart[l + 1] != activeStart[l];
        std::lock_guard<std::mutex> g(baker.timingsMu); baker.timings = tm; baker.haveTimings = true;
    }
    *out = (ommCpuBakeResult)res;
    return ommResult_SUCCESS;
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
OMM_MI355X_API ommLibraryDesc ommGetLibraryDesc(void)
{
    ommLibraryDesc d = { OMM_VERSION_MAJOR, OMM_VERSION_MINOR, OMM_VERSION_BUILD };
    return d;
}

OMM_MI355X_API ommResult ommCreateBaker(const ommBakerCreationDesc* desc, ommBaker* outBaker)
{
    if (desc == nullptr) return ommResult_INVALID_ARGUMENT;
    if (desc->type != ommBakerType_CPU && desc->type != ommBakerType_GPU) return ommResult_INVALID_ARGUMENT;
    Allocator mem;
    if (desc->memoryAllocatorInterface.allocate != nullptr) {
        mem.alloc = desc->memoryAllocatorInterface.allocate; mem.realloc_ = desc->memoryAllocatorInterface.reallocate;
        mem.free_ = desc->memoryAllocatorInterface.free; mem.user = desc->memoryAllocatorInterface.userArg;
    }
    Baker* b = mem.make<Baker>();
    if (!b) return ommResult_FAILURE;
    b->mem = mem; b->log.iface = desc->messageInterface; b->type = desc->type;
    *outBaker = (ommBaker)((uintptr_t)b | (desc->type == ommBakerType_CPU ? kCpuBaker : kGpuBaker));
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommDestroyBaker(ommBaker baker)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    const uintptr_t t = tag_of(baker);
    if (t != kCpuBaker && t != kGpuBaker) return ommResult_FAILURE;
    Baker* b = untag<Baker>(baker);
    const Allocator mem = b->mem;
    mem.destroy(b);
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommCpuCreateTexture(ommBaker baker, const ommCpuTextureDesc* desc, ommCpuTexture* outTexture)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    const Logger& L = b->log;
    if (desc == 0) return L.invalid("texture desc was not set");
    if (tag_of(baker) != kCpuBaker) return L.invalid("Baker was not created as the right type");
    // texture_impl.cpp:44-65
    if (desc->mipCount == 0) return L.invalid("[Invalid Arg] - mipCount must be non-zero");
    if (desc->format == ommCpuTextureFormat_MAX_NUM) return L.invalid("[Invalid Arg] - format is not set");
    for (uint32_t i = 0; i < desc->mipCount; ++i) {
        if (!desc->mips[i].textureData) return L.invalid("[Invalid Arg] - mips.textureData is not set");
        if (desc->mips[i].width == 0) return L.invalid("[Invalid Arg] - mips.width must be non-zero");
        if (desc->mips[i].height == 0) return L.invalid("[Invalid Arg] - mips.height must be non-zero");
        if (desc->mips[i].width > 65536) return L.invalid("[Invalid Arg] - mips.width must be less than kMaxDim.x (65536)");
        if (desc->mips[i].height > 65536) return L.invalid("[Invalid Arg] - mips.height must be less than kMaxDim.y (65536)");
    }
    if (desc->mipCount > (uint32_t)kMaxMips) return L.invalid("[Invalid Arg] - more than 17 mips");
    Texture* t = b->mem.make<Texture>();
    if (!t) return ommResult_FAILURE;
    t->mem = b->mem; t->log = &b->log; t->format = desc->format; t->flags = desc->flags; t->alphaCutoff = desc->alphaCutoff;
    const bool linear = ((uint32_t)desc->flags & (uint32_t)ommCpuTextureFlags_DisableZOrder) != 0;
    const size_t px = desc->format == ommCpuTextureFormat_FP32 ? 4 : 1;
    const bool enableSAT = desc->alphaCutoff >= 0; // texture_impl.cpp:91
    bool ok = true;
    std::vector<uint8_t> staging;
    for (uint32_t mi = 0; mi < desc->mipCount && ok; ++mi) {
        const ommCpuTextureMipDesc& md = desc->mips[mi];
        TexMip m; m.w = (int)md.width; m.h = (int)md.height;
        const size_t rowBytes = px * (size_t)m.w, bytes = rowBytes * (size_t)m.h;
        // rowPitch is in bytes for DisableZOrder textures and in texels otherwise (texture_impl.cpp:141-142,169,179)
        const size_t pitch = linear ? (md.rowPitch == 0 ? rowBytes : (size_t)md.rowPitch) : px * (md.rowPitch == 0 ? (size_t)md.width : (size_t)md.rowPitch);
        const uint8_t* src = (const uint8_t*)md.textureData;
        if (pitch != rowBytes) {
            staging.resize(bytes);
            for (int j = 0; j < m.h; ++j) memcpy(staging.data() + rowBytes * (size_t)j, src + pitch * (size_t)j, rowBytes);
            src = staging.data();
        }
        ok = HIP_OK(hipMalloc(&m.texels, bytes)) && HIP_OK(hipMemcpy(m.texels, src, bytes, hipMemcpyHostToDevice));
        if (ok && enableSAT) {
            ok = HIP_OK(hipMalloc((void**)&m.sat, sizeof(uint32_t) * (size_t)m.w * (size_t)m.h));
            if (ok) { launch_sat_build(m.texels, desc->format == ommCpuTextureFormat_FP32, m.sat, m.w, m.h, desc->alphaCutoff, nullptr); ok = HIP_OK(hipGetLastError()); }
        }
        t->mips.push_back(m);
    }
    if (ok) ok = HIP_OK(hipDeviceSynchronize());
    if (!ok) { (void)hipGetLastError(); b->mem.destroy(t); return L.failure("[Failure] - could not create the texture on the HIP device (no CPU fallback)"); }
    *outTexture = (ommCpuTexture)((uintptr_t)t | kTexture);
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommCpuGetTextureDesc(ommCpuTexture texture, ommCpuTextureDesc* outDesc)
{
    if (texture == 0) return ommResult_INVALID_ARGUMENT;
    Texture* t = untag<Texture>(texture);
    if (t == 0 || outDesc == nullptr) return ommResult_INVALID_ARGUMENT;
    outDesc->format = t->format; outDesc->flags = t->flags; outDesc->alphaCutoff = t->alphaCutoff; outDesc->mipCount = (uint32_t)t->mips.size();
    if (outDesc->mips == nullptr) return ommResult_SUCCESS;
    const size_t px = t->format == ommCpuTextureFormat_FP32 ? 4 : 1;
    for (uint32_t i = 0; i < outDesc->mipCount; ++i) { // texture_impl.cpp:280-325
        ommCpuTextureMipDesc& m = const_cast<ommCpuTextureMipDesc&>(outDesc->mips[i]);
        m.width = (uint32_t)t->mips[i].w; m.height = (uint32_t)t->mips[i].h; m.rowPitch = (uint32_t)t->mips[i].w;
        if (m.textureData != nullptr)
            if (!HIP_OK(hipMemcpy(const_cast<void*>(m.textureData), t->mips[i].texels, px * (size_t)m.width * m.height, hipMemcpyDeviceToHost))) return ommResult_FAILURE;
    }
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommCpuDestroyTexture(ommBaker baker, ommCpuTexture texture)
{
    if (texture == 0) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    if (tag_of(baker) != kCpuBaker) return b ? b->log.invalid("Baker was not created as the right type") : ommResult_INVALID_ARGUMENT;
    Texture* t = untag<Texture>(texture);
    b->mem.destroy(t);
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommCpuBake(ommBaker baker, const ommCpuBakeInputDesc* desc, ommCpuBakeResult* outBakeResult)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    if (desc == 0) return b->log.invalid("input desc was not set");
    if (tag_of(baker) != kCpuBaker) return b->log.invalid("Baker was not created as the right type");
    if (desc->texture == 0) return b->log.invalid("[Invalid Argument] - ommCpuBakeInputDesc has no texture set"); // bake_cpu_impl.cpp:97-103
    // the dispatch table lookup precedes ValidateDesc (bake_cpu_impl.cpp:297-304): unknown sampler enums -> FAILURE
    if (tag_of(desc->texture) == kTexture &&
        ((unsigned)desc->runtimeSamplerDesc.addressingMode >= (unsigned)ommTextureAddressMode_MAX_NUM ||
         (unsigned)desc->runtimeSamplerDesc.filter >= (unsigned)ommTextureFilterMode_MAX_NUM))
        return ommResult_FAILURE;
    const ommResult v = validate_desc(*b, *desc);
    if (v != ommResult_SUCCESS) return v;
    return bake_impl(*b, *desc, outBakeResult);
}

OMM_MI355X_API ommResult ommCpuDestroyBakeResult(ommCpuBakeResult bakeResult)
{
    if (bakeResult == 0) return ommResult_INVALID_ARGUMENT;
    BakeResult* r = (BakeResult*)bakeResult;
    const Allocator mem = r->mem;
    mem.destroy(r);
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommCpuGetBakeResultDesc(ommCpuBakeResult bakeResult, const ommCpuBakeResultDesc** desc)
{
    if (bakeResult == 0) return ommResult_INVALID_ARGUMENT;
    if (desc == nullptr) return ommResult_INVALID_ARGUMENT;
    *desc = &((BakeResult*)bakeResult)->desc;
    return ommResult_SUCCESS;
}

// debug_impl.cpp:512-641 (host-side parse of a finished result; knownAreaMetric needs the per-triangle areas -> 0 here)
OMM_MI355X_API ommResult ommDebugGetStats(ommBaker baker, const ommCpuBakeResultDesc* res, ommDebugStats* out)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    if (res == nullptr || out == nullptr) return ommResult_INVALID_ARGUMENT;
    ommDebugStats st; memset(&st, 0, sizeof st);
    std::vector<uint32_t> refs((size_t)res->descArrayCount + 1, 0);
    for (uint32_t i = 0; i < res->indexCount; ++i) {
        int32_t v;
        if (res->indexFormat == ommIndexFormat_UINT_8) v = ((const int8_t*)res->indexBuffer)[i];
        else if (res->indexFormat == ommIndexFormat_UINT_16) v = ((const int16_t*)res->indexBuffer)[i];
        else v = ((const int32_t*)res->indexBuffer)[i];
        if (v == ommSpecialIndex_FullyTransparent) st.totalFullyTransparent++;
        else if (v == ommSpecialIndex_FullyOpaque) st.totalFullyOpaque++;
        else if (v == ommSpecialIndex_FullyUnknownTransparent) st.totalFullyUnknownTransparent++;
        else if (v == ommSpecialIndex_FullyUnknownOpaque) st.totalFullyUnknownOpaque++;
        else if (v >= 0 && (uint32_t)v < res->descArrayCount) refs[(size_t)v]++;
    }
    for (uint32_t i = 0; i < res->descArrayCount; ++i) {
        if (!refs[i]) continue;
        const ommCpuOpacityMicromapDesc& dd = res->descArray[i];
        const uint8_t* data = (const uint8_t*)res->arrayData + dd.offset;
        const uint32_t nM = 1u << (dd.subdivisionLevel << 1);
        const uint32_t is2 = dd.format == ommFormat_OC1_2_State;
        uint64_t c[4] = { 0, 0, 0, 0 };
        for (uint32_t u = 0; u < nM; ++u) {
            const uint8_t v = data[u >> (2 + is2)];
            c[is2 ? ((v >> (u & 7)) & 1u) : ((v >> ((u << 1) & 7)) & 3u)]++;
        }
        st.totalTransparent += (uint64_t)refs[i] * c[0]; st.totalOpaque += (uint64_t)refs[i] * c[1];
        st.totalUnknownTransparent += (uint64_t)refs[i] * c[2]; st.totalUnknownOpaque += (uint64_t)refs[i] * c[3];
    }
    *out = st;
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxGetLastBakeTimings(ommBaker baker, ommxBakeTimings* out)
{
    if (baker == 0 || out == nullptr || tag_of(baker) != kCpuBaker) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    std::lock_guard<std::mutex> g(b->timingsMu);
    if (!b->haveTimings) return ommResult_FAILURE;
    *out = b->timings;
    return ommResult_SUCCESS;
}
