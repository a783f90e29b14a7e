// bake_types.h -- plain structs shared by the host orchestration and the HIP kernels.
// Domain vocabulary follows the reference: work item = one unique (UV triangle, level, format),
// micro-triangle = one of its 4^level bird-curve cells, state = 2-bit opacity class.
#pragma once
#include <stdint.h>

namespace ommx {

constexpr int kMaxLevel = 12;       // defines.h:25
constexpr int kNumLevels = 13;
constexpr int kMaxMips = 17;        // 65536 = 2^16 -> at most 17 mips
constexpr int kTexCoordBorder = 0x7FFFFFFE; // util/texture.h:23

// One mip level as the kernels see it.  Texels are row-major, tightly packed (Load(x,y) of
// texture_impl.h:178-202 is layout-independent, so the HBM layout is ours to choose).
struct DevMip {
    const void*     texels;   // u8 or f32, w*h
    const uint32_t* sat;      // inclusive summed-area table of (alpha > cutoff), or null
    int   w, h, log2w, log2h, pow2;
    float fw, fh, rw, rh;     // size as float, 1.f/size (texture_impl.cpp:97-105)
};

// Everything one classification launch needs besides the item list.
struct ClassifyParams {
    DevMip mips[kMaxMips];
    int   mipCount;
    int   texIsFp32;
    int   addrMode;           // ommTextureAddressMode
    int   filterLinear;
    int   format;             // 1 = 2-state, 2 = 4-state (global desc.format: bake_cpu_impl.cpp:907)
    int   promotion;          // ommUnknownStatePromotion
    int   stateGT, stateLE;
    int   useCoarse;          // texture has SAT && 1 mip && linear (bake_cpu_impl.cpp:723-727,746)
    float cutoff, borderAlpha;
    int   wantKnownCount;     // rejectionThreshold > 0
    int   noFine;             // internal flag DisableFineClassification (bake_cpu_impl.cpp:45,822-823): ResampleFine is skipped, whatever the coarse pass
                              // left unresolved keeps the initial state UnknownOpaque (:427)
    int   altKernel;          // internal flags DisableLevelLineIntersection (bit 8) / + EnableAABBTesting (bit 7), bake_cpu_impl.cpp:44-45,915-966: 0 = the level-line
                              // kernel; 1 = ConservativeBilinearKernel over the micro-triangle's raster; 2 = the same over the two triangles of its bounding box
                              // (Linear filter only; no centre vote, mip 0 only)
    int   pow2Dispatch;       // SizeIsPow2() of mip 0: the template flag of the reference's kernels (bake_cpu_impl.cpp:299);
                              // TextureImpl::Bilinear alone uses the per-mip flag (texture_impl.cpp:266)
};

// Per work item, device resident (structure of arrays).
// the level-line statistic is bumped once per tile (millions per bake): striped over 256 cache lines, summed on the host
constexpr int kFineSlots = 256, kFineStride = 16;
struct ItemArrays {
    const float*    uv;        // 6 floats per item: p0.x p0.y p1.x p1.y p2.x p2.y
    const uint8_t*  degenerate;// area < 1e-9 (util/geometry.h:44-47)
    const uint64_t* stateOfs;  // byte offset of the item's packed states in `states`
    uint8_t*        states;    // packed 1-/2-bit states, LSB first (bake_cpu_impl.cpp:1806-1816)
    uint32_t*       stateMask; // OR of (1 << state) over the item's micro-triangles
    uint32_t*       knownCount;// number of T/O micro-triangles (only when wantKnownCount)
    unsigned long long* fineCount; // statistics: micro-triangles that went through the level-line pass; kFineSlots counters, kFineStride apart
};

} // namespace ommx
