// classify_device.h -- per-micro-triangle classification for gfx950, bit-exact with the
// reference CPU baker.
//
// Every expression keeps the reference's fp32 association order; this translation unit is
// compiled with -ffp-contract=off (no v_fma contraction), IEEE-correct division/sqrt (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt) and fp32 denormals on (gfx950 default), so
// the VALU results equal what the reference's -msse4.1 build computes on x86.
// Citations are relative to /root/reference/libraries/omm-lib/src/.
#pragma once
#include <hip/hip_runtime.h>
#include "bake_types.h"
#include "region_curve.h"

namespace ommx {

struct V2 { float x, y; };
__device__ __forceinline__ V2 mk2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }

// x86 cvttss2si: NaN / out-of-range -> INT_MIN ("integer indefinite"); v_cvt_i32_f32 saturates instead.
__device__ __forceinline__ int cvt_trunc_x86(float f)
{
    return (f >= -2147483648.f && f < 2147483648.f) ? (int)f : (int)0x80000000;
}
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float vlen(float x, float y) { return __builtin_sqrtf(x * x + y * y); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (hi < v ? hi : v); }

// UV-space micro-triangle with the cached edge vectors of util/geometry.h:59-75.
struct MicroTri {
    V2 p0, p1, p2;
    V2 p0p2, p1p0, p2p1;
    V2 lo, hi; // aabb_s, aabb_e
};

__device__ __forceinline__ void finish_tri(MicroTri& t)
{
    t.p0p2 = mk2(t.p0.x - t.p2.x, t.p0.y - t.p2.y);
    t.p1p0 = mk2(t.p1.x - t.p0.x, t.p1.y - t.p0.y);
    t.p2p1 = mk2(t.p2.x - t.p1.x, t.p2.y - t.p1.y);
    t.lo = mk2(std_min(std_min(t.p0.x, t.p1.x), t.p2.x), std_min(std_min(t.p0.y, t.p1.y), t.p2.y));
    t.hi = mk2(std_max(std_max(t.p0.x, t.p1.x), t.p2.x), std_max(std_max(t.p0.y, t.p1.y), t.p2.y));
}

// ---- bird curve: micro-triangle index -> barycentrics (util/bird.h:34-118) ----
__host__ __device__ __forceinline__ uint32_t even_bits(uint32_t x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x0000ffffu;
    return x;
}
__host__ __device__ __forceinline__ uint32_t prefix_xor(uint32_t x)
{
    x ^= x >> 1; x ^= x >> 2; x ^= x >> 4; x ^= x >> 8;
    return x;
}

// util/geometry.h:241-248 with bc = (1-u-v, u, v)
__device__ __forceinline__ V2 bary_point(const float* __restrict__ tri, float u, float v)
{
    const float bx = 1.f - u - v;
    return mk2(tri[0] * bx + tri[2] * u + tri[4] * v, tri[1] * bx + tri[3] * u + tri[5] * v);
}

// util/bird.h:170-182
__device__ __forceinline__ MicroTri micro_triangle(const float* __restrict__ tri, uint32_t index, uint32_t level)
{
    MicroTri t;
    if (level == 0) {
        t.p0 = bary_point(tri, 0.f, 0.f); t.p1 = bary_point(tri, 1.f, 0.f); t.p2 = bary_point(tri, 0.f, 1.f);
    } else {
        const uint32_t b0 = even_bits(index), b1 = even_bits(index >> 1);
        const uint32_t fx = prefix_xor(b0), fy = prefix_xor(b0 & ~b1);
        const uint32_t tt = fy ^ b1;
        uint32_t iu = (fx & ~tt) | (b0 & ~tt) | (~b0 & ~fx & tt);
        uint32_t iv = fy ^ b0;
        uint32_t iw = (~fx & ~tt) | (b0 & ~tt) | (~b0 & fx & tt);
        const uint32_t mask = (1u << level) - 1u;
        iu &= mask; iv &= mask; iw &= mask;
        const bool upright = ((iu ^ iv ^ iw) & 1u) != 0;
        if (!upright) { iu += 1; iv += 1; }
        const float ls = __uint_as_float((127u - level) << 23);
        float du = 1.f * ls, dv = 1.f * ls;
        const float u = (float)iu * ls, v = (float)iv * ls;
        if (!upright) { du = -du; dv = -dv; }
        t.p0 = bary_point(tri, u, v); t.p1 = bary_point(tri, u + du, v); t.p2 = bary_point(tri, u, v + dv);
    }
    finish_tri(t);
    return t;
}

// ---- bird curve, split at the 64-group boundary ----
// index = group * 64 + l.  Every step of the decode above is bitwise except the two prefix-xors, and bit i of a prefix-xor is the xor of the
// bits >= i: the low three bits of fx / fy are the prefix-xor of the low three input bits, flipped when the high part has odd parity.
// Hence the integer barycentrics of micro-triangle l of a group are
//     iu = (IU << 3 | lo_u) + !upright,   iv = (IV << 3 | lo_v) + !upright
// with (IU, IV) = the masked, un-adjusted (iu, iv) of the GROUP's own decode at level N - 3, and (lo_u, lo_v, upright) a function of l and
// of the two parities (pX, pY) = low bits of the group's fx / fy only: a 4 x 64-entry table.  The group part is computed once per group
// (by the lane that also asks the SAT about the group), the table once per workgroup; the per-micro-triangle decode shrinks from ~55
// integer instructions to two LDS reads and a dozen bit operations.  Same integers, hence the same vertices.
struct BirdGroup { uint32_t word; };   // IU | IV << 12 | (pX | pY << 1) << 24
__host__ __device__ __forceinline__ BirdGroup bird_group(uint32_t group, uint32_t groupLevel)   // groupLevel = N - 3 >= 0
{
    BirdGroup g;
    if (groupLevel == 0) { g.word = 0u; return g; }
    const uint32_t b0 = even_bits(group), b1 = even_bits(group >> 1);
    const uint32_t fx = prefix_xor(b0), fy = prefix_xor(b0 & ~b1);
    const uint32_t tt = fy ^ b1;
    const uint32_t mask = (1u << groupLevel) - 1u;
    const uint32_t iu = ((fx & ~tt) | (b0 & ~tt) | (~b0 & ~fx & tt)) & mask, iv = (fy ^ b0) & mask;
    g.word = iu | (iv << 12) | ((fx & 1u) << 24) | ((fy & 1u) << 25);
    return g;
}
// table entry for (ctx = pX | pY << 1, l): lo_u | lo_v << 3 | upright << 6
__host__ __device__ __forceinline__ uint32_t bird_table_entry(uint32_t ctx, uint32_t l)
{
    const uint32_t c0 = (l & 1u) | ((l >> 1) & 2u) | ((l >> 2) & 4u), c1 = ((l >> 1) & 1u) | ((l >> 2) & 2u) | ((l >> 3) & 4u);   // even / odd bits of l
    auto pxor3 = [](uint32_t x) { x ^= x >> 1; x ^= x >> 2; return x & 7u; };
    const uint32_t fx = pxor3(c0) ^ ((ctx & 1u) ? 7u : 0u), fy = pxor3(c0 & ~c1 & 7u) ^ ((ctx & 2u) ? 7u : 0u);
    const uint32_t tt = fy ^ c1;
    const uint32_t iu = ((fx & ~tt) | (c0 & ~tt) | (~c0 & ~fx & tt)) & 7u, iv = (fy ^ c0) & 7u, iw = ((~fx & ~tt) | (c0 & ~tt) | (~c0 & fx & tt)) & 7u;
    return iu | (iv << 3) | (((iu ^ iv ^ iw) & 1u) << 6);
}
// micro-triangle l (0..63) of a group at level N >= 3, from the group word and the table entry: the same MicroTri as micro_triangle(tri, group * 64 + l, N)
__device__ __forceinline__ MicroTri micro_triangle_grouped(const float* __restrict__ tri, BirdGroup g, uint32_t entry, uint32_t level)
{
    const bool upright = (entry >> 6) != 0u;
    const uint32_t adj = upright ? 0u : 1u;
    const uint32_t iu = ((((g.word & 0xFFFu) << 3) | (entry & 7u)) + adj), iv = (((((g.word >> 12) & 0xFFFu)) << 3) | ((entry >> 3) & 7u)) + adj;
    const float ls = __uint_as_float((127u - level) << 23);
    float du = 1.f * ls, dv = 1.f * ls;
    const float u = (float)iu * ls, v = (float)iv * ls;
    if (!upright) { du = -du; dv = -dv; }
    MicroTri t;
    t.p0 = bary_point(tri, u, v); t.p1 = bary_point(tri, u + du, v); t.p2 = bary_point(tri, u, v + dv);
    finish_tri(t);
    return t;
}

// Address-mode policy: the reference instantiates its kernels per (address mode, pow2) pair (bake_cpu_impl.cpp:128-228);
// ModeStatic<> does the same for the common pairs so the switch in tex_coord() folds away, ModeDynamic reads the desc.
struct ModeDynamic {
    __device__ static __forceinline__ int addr(const ClassifyParams& P) { return P.addrMode; }
    __device__ static __forceinline__ int pow2(const ClassifyParams& P) { return P.pow2Dispatch; }
};
template <int AM, int P2> struct ModeStatic {
    __device__ static __forceinline__ int addr(const ClassifyParams&) { return AM; }
    __device__ static __forceinline__ int pow2(const ClassifyParams&) { return P2; }
};

// ---- texture addressing (util/texture.h:34-91) ----
__device__ __forceinline__ int tex_coord(int mode, int pow2, int x, int size, int sizeLog2)
{
    switch (mode) {
    case 0: // Wrap
        return pow2 ? (int)((uint32_t)x & (uint32_t)(size - 1)) : (int)((uint32_t)x % (uint32_t)size);
    case 1: // Mirror
        if (pow2) {
            const int xa = (x < 0 ? -x : x) - (x < 0 ? 1 : 0);
            const int flipped = (xa >> sizeLog2) & 1;
            const int w = (int)((uint32_t)xa & (uint32_t)(size - 1));
            return flipped ? size - w - 1 : w;
        } else {
            const int xa = cvt_trunc_x86(__builtin_fabsf((float)x + 0.5f));
            const uint32_t flipped = ((uint32_t)(xa / size)) % 2u;
            const int w = (int)((uint32_t)xa % (uint32_t)size);
            return flipped ? size - w - 1 : w;
        }
    case 2: // Clamp
        return clampi(x, 0, size - 1);
    case 3: // Border
        return (x >= size || x < 0) ? kTexCoordBorder : x;
    case 4: // MirrorOnce
        return clampi(cvt_trunc_x86(__builtin_fabsf((float)x + 0.5f)), 0, size - 1);
    default:
        return 0x7FFFFFFF;
    }
}

// LDS copy of the part of mip 0 (texel values already converted to float exactly as Load() does, and the summed-area
// table) that a tile of micro-triangles can touch.  Coordinates are ADDRESSED texel coordinates; the window only exists
// when the address mode maps the tile's footprint without a seam (w == 0 otherwise).  A fetch outside the window falls
// back to HBM, so the window is a pure cache: it never changes a value.
typedef __attribute__((address_space(3))) const float    lds_float;
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
struct TexWindow {
    lds_float* tex;        // w * h floats, row-major (LDS)
    lds_u32*   sat;        // (w + 1) * (h + 1): entry (i, j) = SAT(sx - 1 + i, sy - 1 + j), 0 where the coordinate is -1 (LDS)
    const void*     base;  // texels of mip 0 (the window belongs to that mip only)
    int sx, sy, w, h;
};

// texture_impl.h:178-202
template <bool FP32>
__device__ __forceinline__ float load_texel(const DevMip& m, int x, int y, const TexWindow& W)
{
    const uint32_t wx = (uint32_t)(x - W.sx), wy = (uint32_t)(y - W.sy);
    if (wx < (uint32_t)W.w && wy < (uint32_t)W.h && m.texels == W.base) return W.tex[wx + wy * (uint32_t)W.w];
    const size_t idx = (size_t)x + (size_t)y * (size_t)m.w;
    if (FP32) return ((const float*)m.texels)[idx];
    return (float)((const uint8_t*)m.texels)[idx] * (1.f / 255.f);
}

template <bool FP32>
__device__ __forceinline__ float load_texel_border(const DevMip& m, int x, int y, float borderAlpha, const TexWindow& W)
{
    return (x == kTexCoordBorder || y == kTexCoordBorder) ? borderAlpha : load_texel<FP32>(m, x, y, W);
}

// The 2 x 2 texels of the cell whose low corner is the (unaddressed) texel (px, py): 00, 01 (y+1), 11, 10 (x+1), each addressed like
// the reference does (GatherTexCoord4, util/texture.h:130-148).  When the whole cell lies in the tile's LDS window -- where
// addressing is a pure translation, which is re-checked here through x1 == x0 + 1, y1 == y0 + 1 -- the four values come from one
// index computation; otherwise every texel goes through load_texel() / the border rule on its own.
template <bool FP32, class MD, bool PAIRS = false>
__device__ __forceinline__ void fetch_cell(const ClassifyParams& P, const DevMip& m, int pow2, int px, int py, const TexWindow& W,
                                           float& g00, float& g01, float& g11, float& g10)
{
    const int x0 = tex_coord(MD::addr(P), pow2, px, m.w, m.log2w), y0 = tex_coord(MD::addr(P), pow2, py, m.h, m.log2h);
    const int x1 = tex_coord(MD::addr(P), pow2, px + 1, m.w, m.log2w), y1 = tex_coord(MD::addr(P), pow2, py + 1, m.h, m.log2h);
#ifndef OMMX_NO_CELL_FETCH
    {
        const uint32_t wx = (uint32_t)(x0 - W.sx), wy = (uint32_t)(y0 - W.sy);
        if (wx + 1u < (uint32_t)W.w && wy + 1u < (uint32_t)W.h && x1 == x0 + 1 && y1 == y0 + 1 && m.texels == W.base) {
            const uint32_t i = wx + wy * (uint32_t)W.w;
            g00 = W.tex[i]; g10 = W.tex[i + 1u]; g01 = W.tex[i + (uint32_t)W.w]; g11 = W.tex[i + (uint32_t)W.w + 1u];
            return;
        }
    }
#endif
    // outside the window: the two texels of a row are neighbours in memory unless the address mode folds the cell at a seam -- one load per row
    // (the deferred generic pass does nothing but such fetches: classify_generic 28.6 -> 27.8 ms)
    // (PAIRS: only where the caller asks for it -- in the persistent kernel the extra path costs more registers than the rare fetch outside the window saves)
    if (PAIRS && MD::addr(P) != 3 && x1 == x0 + 1) {
        const size_t i0 = (size_t)x0 + (size_t)y0 * (size_t)m.w, i1 = (size_t)x0 + (size_t)y1 * (size_t)m.w;
        if (FP32) {
            typedef float __attribute__((ext_vector_type(2), aligned(4))) f32x2u;
            const f32x2u r0 = *(const f32x2u*)((const float*)m.texels + i0), r1 = *(const f32x2u*)((const float*)m.texels + i1);
            g00 = r0.x; g10 = r0.y; g01 = r1.x; g11 = r1.y;
        } else {
            typedef uint16_t __attribute__((aligned(1))) u16u;
            const uint32_t r0 = *(const u16u*)((const uint8_t*)m.texels + i0), r1 = *(const u16u*)((const uint8_t*)m.texels + i1);
            g00 = (float)(r0 & 0xFFu) * (1.f / 255.f); g10 = (float)(r0 >> 8) * (1.f / 255.f);
            g01 = (float)(r1 & 0xFFu) * (1.f / 255.f); g11 = (float)(r1 >> 8) * (1.f / 255.f);
        }
        return;
    }
    if (MD::addr(P) == 3) {
        g00 = load_texel_border<FP32>(m, x0, y0, P.borderAlpha, W); g01 = load_texel_border<FP32>(m, x0, y1, P.borderAlpha, W);
        g11 = load_texel_border<FP32>(m, x1, y1, P.borderAlpha, W); g10 = load_texel_border<FP32>(m, x1, y0, P.borderAlpha, W);
    } else {
        g00 = load_texel<FP32>(m, x0, y0, W); g01 = load_texel<FP32>(m, x0, y1, W);
        g11 = load_texel<FP32>(m, x1, y1, W); g10 = load_texel<FP32>(m, x1, y0, W);
    }
}

// one summed-area-table entry of mip 0 (x, y >= 0)
__device__ __forceinline__ uint32_t load_sat(const DevMip& m, int x, int y, const TexWindow& W)
{
    const uint32_t wx = (uint32_t)(x - W.sx + 1), wy = (uint32_t)(y - W.sy + 1);
    if (wx <= (uint32_t)W.w && wy <= (uint32_t)W.h && W.w != 0) return W.sat[wx + wy * (uint32_t)(W.w + 1)];
    return m.sat[(size_t)x + (size_t)y * (size_t)m.w];
}

// texture_impl.h:110-125
__device__ __forceinline__ uint32_t sat_sum(const DevMip& m, int sx, int sy, int ex, int ey, const TexWindow& W)
{
#ifndef OMMX_NO_CELL_FETCH
    {   // all four corners in the LDS window: its entry (i, j) is SAT(W.sx - 1 + i, W.sy - 1 + j) with the zero row / column for -1
        // already stored, so the "sx > 0 / sy > 0" selections below are table look-ups
        const uint32_t i0 = (uint32_t)(sx - W.sx), j0 = (uint32_t)(sy - W.sy), i1 = (uint32_t)(ex - W.sx + 1), j1 = (uint32_t)(ey - W.sy + 1);
        const uint32_t pitch = (uint32_t)W.w + 1u;
        if (W.w != 0 && i0 <= (uint32_t)W.w && i1 <= (uint32_t)W.w && j0 <= (uint32_t)W.h && j1 <= (uint32_t)W.h)
            return W.sat[i1 + j1 * pitch] + W.sat[i0 + j0 * pitch] - W.sat[i1 + j0 * pitch] - W.sat[i0 + j1 * pitch];
    }
#endif
    const uint32_t A = (sx > 0 && sy > 0) ? load_sat(m, sx - 1, sy - 1, W) : 0u;
    const uint32_t B = sy > 0 ? load_sat(m, ex, sy - 1, W) : 0u;
    const uint32_t C = sx > 0 ? load_sat(m, sx - 1, ey, W) : 0u;
    const uint32_t D = load_sat(m, ex, ey, W);
    return D + A - B - C;
}

// texture_impl.cpp:261-278 (centre vote).  Border texels take borderAlpha (the reference reads out of
// bounds there -- documented fence).
template <bool FP32, class MD, bool PAIRS = false>
__device__ __forceinline__ float bilinear(const ClassifyParams& P, const DevMip& m, V2 p, const TexWindow& W)
{
    const float px = p.x * m.fw - 0.5f, py = p.y * m.fh - 0.5f;
    const float fx = __builtin_floorf(px), fy = __builtin_floorf(py);
    const int ix = cvt_trunc_x86(fx), iy = cvt_trunc_x86(fy);
    float a, b, c, d; // 00, 01, 10, 11 (the sentinel of Border addressing reads borderAlpha: documented fence)
    fetch_cell<FP32, MD, PAIRS>(P, m, m.pow2, ix, iy, W, a, b, d, c);
    const float wx = px - fx, wy = py - fy;
    const float ac = a * (1.f - wx) + c * wx;
    const float bd = b * (1.f - wx) + d * wx;
    return ac * (1.f - wy) + bd * wy;
}

// ---- coverage -> state (bake_kernels_cpu.h:25-61) ----
__device__ __forceinline__ int state_from_coverage(const ClassifyParams& P, uint32_t above, uint32_t below)
{
    if (above != 0 && below != 0) {
        if (P.format == 2) {
            if (P.promotion == 1) return 3;
            if (P.promotion == 2) return 2;
            return above >= below ? (P.stateGT | 2) : (P.stateLE | 2);
        }
        if (P.promotion == 1) return 1;
        if (P.promotion == 2) return 0;
        return above >= below ? P.stateGT : P.stateLE;
    }
    return above == 0 ? P.stateLE : P.stateGT;
}
__device__ __forceinline__ bool state_is_unknown(int s) { return s >= 2; }
// one sample's vote.  Written without an if/else on the two counters: clang turns `if (c) above++; else below++;` into a pointer
// select over a 2-dword private array, i.e. scratch (HBM-backed) loads and stores in the hottest loop.
__device__ __forceinline__ void vote(bool isAbove, uint32_t& above, uint32_t& below)
{
    above += isAbove ? 1u : 0u;
    below += isAbove ? 0u : 1u;
}

// ---- geometry predicates ----
// util/geometry.h:101-114
__device__ __forceinline__ bool point_in_triangle(const MicroTri& t, float px, float py)
{
    const float s = t.p0p2.x * (py - t.p2.y) - t.p0p2.y * (px - t.p2.x);
    const float tt = t.p1p0.x * (py - t.p0.y) - t.p1p0.y * (px - t.p0.x);
    if (((s < 0) != (tt < 0)) && s != 0 && tt != 0) return false;
    const float d = t.p2p1.x * (py - t.p1.y) - t.p2p1.y * (px - t.p1.x);
    return d == 0 || ((d < 0) == (s + tt <= 0));
}

// The same predicate without the early exit (all three cross products, bitwise combination): in the single-texel pass the four corner
// tests of a micro-triangle are straight-line code for the whole wave -- the early exit saved five instructions per corner and cost two
// divergent regions (exec save / restore + branch) each; without it classify_tiles runs 30.6 -> 29.4 ms.
__device__ __forceinline__ bool point_in_triangle_flat(const MicroTri& t, float px, float py)
{
    const float s = t.p0p2.x * (py - t.p2.y) - t.p0p2.y * (px - t.p2.x);
    const float tt = t.p1p0.x * (py - t.p0.y) - t.p1p0.y * (px - t.p0.x);
    const float d = t.p2p1.x * (py - t.p1.y) - t.p2p1.y * (px - t.p1.x);
    const bool reject = ((s < 0) != (tt < 0)) & (s != 0) & (tt != 0);
    const bool accept = (d == 0) | ((d < 0) == (s + tt <= 0));
    return !reject & accept;
}
__device__ __forceinline__ bool near_zero(float v, float eps) { return v < eps && v > -eps; }
__device__ __forceinline__ bool in_unit_square(float x, float y) { return x >= 0.f && x <= 1.f && y >= 0.f && y <= 1.f; }

// Edge::IsPointOnEdge (bake_kernels_cpu.h:115-142): |P-a0| + |P-a1| - |a1-a0| within 1e-5 of zero -- three IEEE square roots.
// Only ~1 % of the candidate roots are anywhere near the segment, and a wave pays for the square roots as soon as ONE lane needs
// them, so a sqrt-free test first discards points that are PROVABLY off the segment:
//   with s = projection of P on the edge direction, slack S = |P-a0| + |P-a1| - L >= 2 (s - L) beyond the far end and >= -2 s before
//   the near end; s - L = (w - L^2) / L with w = (P-a0).(a1-a0), and L <= M = |dx| + |dy|.  The fp32 evaluation of the reference's
//   expression is within 5.4e-7 (|P-a0| + |P-a1|) of S and that of w - L^2 within 2.4e-7 M (U + M), U = |ux| + |uy|; the threshold
//   T = M (1e-4 + 1e-5 (U + M)) leaves a >10x margin over both, so "w - L^2 > T or -w > T" implies the reference's value exceeds
//   1e-5 (measured: smallest |value| among discarded points 2.2e-4 over 2.5e7 calls, no disagreement).  NaN/inf compare false
//   and take the exact path.  Everything not discarded is evaluated exactly as the reference does.
__device__ __forceinline__ bool point_on_edge(V2 a0, V2 a1, float x, float y)
{
    const float ux = x - a0.x, uy = y - a0.y;
    const float dx = a1.x - a0.x, dy = a1.y - a0.y;
#ifndef OMMX_NO_EDGE_PREFILTER
    const float w = ux * dx + uy * dy, l2 = dx * dx + dy * dy;
    const float M = __builtin_fabsf(dx) + __builtin_fabsf(dy), U = __builtin_fabsf(ux) + __builtin_fabsf(uy);
    const float T = M * (1e-4f + 1e-5f * (U + M));
    if (w - l2 > T || -w > T) return false;
#endif
    return near_zero(vlen(ux, uy) + vlen(x - a1.x, y - a1.y) - vlen(dx, dy), 1e-5f);
}

// Root filter.  A root of the level curve on the edge's carrier line counts only if the reference's point
// P = (x, y) passes in_unit_square AND IsPointOnEdge.  Measured on the bench workload 99.5 % of the roots lie outside the edge's x-range,
// yet each costs an IEEE division (the most expensive operation of the kernel) before that is known.  The filter decides from the
// division's OPERANDS whether its correctly rounded quotient x = RN(n / c) can lie in the only interval where the reference can accept:
//   (i)  accepted  =>  P in [0,1]^2 and computed slack < 1e-5  =>  true slack S(P) = |P-a0| + |P-a1| - L <= eps = 2.4e-5 (the fp32
//        evaluation error of the slack is <= 5.4e-7 (S + L), section 5.3, so S <= 1.11e-5 for L <= 2: eps allows 13x that error)  =>  P
//        lies in the ellipse with foci a0, a1 and major axis L + eps, whose x-extent is sqrt((kd/2)^2 + B^2) <= kd/2 + B around the
//        midpoint, B = sqrt(L eps / 2 + eps^2 / 4) < 4.9e-3 for L <= M = |dx| + |dy| <= 2.
//        Hence x in I = [max(0, a0.x - 5e-3), min(1, a1.x + 5e-3)].
//   (ii) with s = sign(c): n s > fl((hi + 1e-3) |c|)  =>  n / c > (hi + 1e-3)(1 - 2^-24) > hi + 9e-4  =>  RN(n / c) > hi (rounding is
//        monotone); likewise below lo.  |c| >= 1e-6 and |hi + 1e-3| >= 1e-3, |lo - 1e-3| is 0 or >= 1e-10: no product underflows.
// NaN / Inf operands compare false and take the exact path; so do edges with M > 2.  Everything not rejected is computed exactly as before.
struct RootFilter { float lo, hi; bool on; };
__device__ __forceinline__ RootFilter root_filter(V2 a0, V2 a1) // a0.x <= a1.x
{
    RootFilter f;
#ifndef OMMX_NO_ROOT_FILTER
    const float M = __builtin_fabsf(a1.x - a0.x) + __builtin_fabsf(a1.y - a0.y);
    f.on = M <= 2.f;
    float lo = a0.x - 5e-3f; lo = lo > 0.f ? lo : 0.f;
    float hi = a1.x + 5e-3f; hi = hi < 1.f ? hi : 1.f;
    f.lo = lo - 1e-3f; f.hi = hi + 1e-3f;
#else
    f.on = false; f.lo = f.hi = 0.f;
#endif
    return f;
}
// true when RN(n / c) is provably outside the acceptance interval
__device__ __forceinline__ bool root_rejected(const RootFilter& f, float n, float c)
{
    const float ac = __builtin_fabsf(c), ns = c < 0.f ? -n : n;
    return f.on & ((ns > f.hi * ac) | (ns < f.lo * ac));   // (bitwise on purpose: no short-circuit regions around two compares, 28.5 -> 28.1 ms)
}

// bake_kernels_cpu.h:144-238 -- does segment (a0,a1) cross the level curve
// h.x + h.y*x + h.z*y + h.w*x*y = 0 inside the unit texel?
__device__ __forceinline__ bool edge_crosses_level_curve(V2 a0, V2 a1, float ha, float hb, float hc, float hd)
{
    if (a0.x > a1.x) { V2 t = a0; a0 = a1; a1 = t; }
    // Edge::_length (bake_kernels_cpu.h:115-133) is only consumed by IsPointOnEdge: evaluated lazily, same value
#define ON_EDGE(X, Y) point_on_edge(a0, a1, (X), (Y))
    const float kd = a1.x - a0.x;
    if (near_zero(kd, 1e-6f)) {
        const float x = a0.x;
        const float c0 = hd * x + hc;
        const float c1 = ha + hb * x;
        if (near_zero(c0, 1e-6f)) return false;
        const float y = -c1 / c0;
        return in_unit_square(x, y) && ON_EDGE(x, y);
    }
    const RootFilter rf = root_filter(a0, a1);
    const float k = (a1.y - a0.y) / kd;
    const float m = a1.y - a1.x * k;
    const float c0 = hd * k;
    const float c1 = hc * k + hd * m + hb;
    const float c2 = ha + hc * m;
    if (near_zero(c0, 1e-6f)) {
        if (near_zero(c1, 1e-6f)) return false;
        if (root_rejected(rf, -c2, c1)) return false;
        const float x = -c2 / c1;
        const float y = k * x + m;
        return in_unit_square(x, y) && ON_EDGE(x, y);
    }
    const float inner = c1 * c1 - 4 * c0 * c2;
    if (inner > 0.f) {
        const float root = __builtin_sqrtf(inner);
        const float n0 = 0.5f * (-c1 + root), n1 = 0.5f * (-c1 - root);
        bool i0 = false, i1 = false;
        if (!root_rejected(rf, n0, c0)) { const float x0 = n0 / c0, y0 = k * x0 + m; i0 = in_unit_square(x0, y0) && ON_EDGE(x0, y0); }
        if (!root_rejected(rf, n1, c0)) { const float x1 = n1 / c0, y1 = k * x1 + m; i1 = in_unit_square(x1, y1) && ON_EDGE(x1, y1); }
        return i0 || i1;
    }
    return false;
#undef ON_EDGE
}

// ---- curve exclusion: can ANY of the three edge tests of a micro-triangle succeed in this texel?   ----
// An edge test succeeds only if a computed root (x^, y^) lies in the unit texel and on the edge (IsPointOnEdge).  Whichever branch of
// TestEdgeHyperbolaIntersection produced it, such a point (i) lies within 5e-3 of the edge's bounding box (the ellipse argument of the root
// filter above, edges with |dx| + |dy| <= 2) and (ii) is an approximate zero of the bilinear patch f(x, y) = ha + hb x + hc y + hd x y: with
// K >= |slope| of every edge, k^ / m^ the computed slope / intercept, c0..c2 the computed coefficients of f along the carrier line, forward
// error analysis of the three branches gives |f(x^, y^)| <= R with
//     vertical edge     R <= 5.2 e S                                              (S = |ha| + |hb| + |hc| + |hd|, e = 6e-8 >= 2^-24)
//     |c0| < 1e-6       R <= 1.0001e-6 + e (|hd| K + 3 C1 + 3 C2 + 1.01 (|hc| + |hd|)(K + 1))
//     quadratic         R <= e (1.01 C1^2 / |c0| + 6.1 C2 + 5.1 C1 + 5.1 |hd| K + 1.01 (|hc| + |hd|)(K + 1))
// where C1 >= |c1|, C2 >= |c2| are bounds in terms of K and |coordinates| <= 2.5 (the residual of the quadratic formula is (rho^2 - D) / 4 c0 for
// the computed root rho of the computed discriminant: no cancellation enters it; the 1 / |c0| term is the genuine ill-conditioning of a small
// leading coefficient, and |c0| >= max(1e-6, |hd| K_min) in that branch).  f is bilinear, so over the bounding box of the micro-triangle fattened
// by 6e-3 its extremes sit in the four corners: if they have one sign and the smallest |f| among them exceeds the sum of the three bounds plus the
// fp32 evaluation error of the corners (66 e S), no edge test can succeed, and all three -- a division and usually a square root each -- are
// skipped.  The two slopes bounds take one v_rcp_f32 each (1 ulp), widened by 2e-6.  Anything the test does not cover (micro-triangles wider than
// a texel, a nearly vertical edge, NaN) answers false and is evaluated exactly.  Audited like the two filters above: the audit build of the oracle
// evaluates this predicate (with reciprocals on the permissive side of anything v_rcp_f32 can return) next to the three edge tests for every
// call: 5.5 M calls on the bench workloads, 88 % excluded, no disagreement (tests/test_edge_prefilter_audit.py).
__device__ __forceinline__ bool curve_excluded(V2 r0, V2 r1, V2 r2, float ha, float hb, float hc, float hd)
{
    const float E = 6.0e-8f;
    const float lx = __builtin_fminf(__builtin_fminf(r0.x, r1.x), r2.x), hx = __builtin_fmaxf(__builtin_fmaxf(r0.x, r1.x), r2.x);
    const float ly = __builtin_fminf(__builtin_fminf(r0.y, r1.y), r2.y), hy = __builtin_fmaxf(__builtin_fmaxf(r0.y, r1.y), r2.y);
    const float dxmax = hx - lx, dymax = hy - ly;
    const float dxmin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(r1.x - r0.x), __builtin_fabsf(r2.x - r1.x)), __builtin_fabsf(r0.x - r2.x));
    const float dymin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(r1.y - r0.y), __builtin_fabsf(r2.y - r1.y)), __builtin_fabsf(r0.y - r2.y));
    const bool ok = (dxmax <= 1.f) & (dymax <= 1.f) & (lx >= -1.5f) & (hx <= 2.5f) & (ly >= -1.5f) & (hy <= 2.5f) & (dxmin >= 1e-4f);
    const float Kub = (dymax * __builtin_amdgcn_rcpf(dxmin)) * 1.000002f, Klb = (dymin * __builtin_amdgcn_rcpf(dxmax)) * 0.999998f;
    const float aa = __builtin_fabsf(ha), ab = __builtin_fabsf(hb), ac = __builtin_fabsf(hc), ad = __builtin_fabsf(hd);
    const float S = aa + ab + ac + ad;
    const float Mb = 2.5f * (1.f + Kub) * 1.000001f;
    const float C1b = (ac * Kub + ad * Mb + ab) * 1.000001f;
    const float C2b = (aa + ac * Mb) * 1.000001f;
    const float c0lb = __builtin_fmaxf(ad * Klb * 0.999999f, 0.999999e-6f);
    const float rest = 1.0001e-6f + E * (9.1f * C2b + 8.1f * C1b + 6.1f * ad * Kub + 2.02f * (ac + ad) * (Kub + 1.f)) + 66.f * E * S + 1e-30f;
    const float rho = 6e-3f;
    const float x0 = lx - rho, x1 = hx + rho, y0 = ly - rho, y1 = hy + rho;
    const float g0 = ha + hc * y0, h0 = hb + hd * y0, g1 = ha + hc * y1, h1 = hb + hd * y1;
    const float f00 = g0 + x0 * h0, f10 = g0 + x1 * h0, f01 = g1 + x0 * h1, f11 = g1 + x1 * h1;
    const float fmn = __builtin_fminf(__builtin_fminf(f00, f10), __builtin_fminf(f01, f11)), fmx = __builtin_fmaxf(__builtin_fmaxf(f00, f10), __builtin_fmaxf(f01, f11));
    const float F = __builtin_fmaxf(fmn, -fmx) - rest;
    return ok & (F > 0.f) & (F * c0lb > 1.01f * E * C1b * C1b);
}

// ---- cell exclusion: curve_excluded()'s question for micro-triangles LARGER than the texel (the deferred generic pass: classify_generic) ----
// r0..r2 are the vertices in the texel's frame (the texel is [0,1]^2), up to tens of texels away.  An edge test succeeds only if its computed root
// (x^, y^) passes in_unit_square -- an exact test on the computed values -- so the region that matters is the unit cell itself, where the extremes of the
// bilinear f sit in the four corners: f00 = ha, f10 = ha + hb, f01 = ha + hc, f11 = ha + hb + hc + hd.  The approximate-zero bounds R of the three
// branches (above) hold with the coordinate bound 2.5 replaced by Rc = the largest |coordinate| of the three vertices: it enters only through the
// intercept |m| <= Rc (1 + K) of an edge's carrier line and, with it, the bounds C1, C2 on the line's coefficients.  Slopes are bounded per edge
// (one v_rcp_f32 each, widened by 2e-6) instead of over the triangle: asset triangles have axis-aligned edges, for which a common bound is useless --
//     |kd| < 1e-6       the reference's vertical branch (same float, same decision): R = 5.2 e S;
//     dy == 0           k = +-0 exactly, so c0 = +-0 and the reference takes its linear branch: R_linear with K = 0;
//     1e-6 <= |kd| < 1e-4   steep, the bounds explode: not excluded.
// If min |f| over the corners, less the evaluation error of the corners (66 e S), exceeds every edge's bound, the level curve cannot be found inside this
// texel by any of the three edge tests, and they are skipped.  The kernel evaluates the bounds with e = 4.8e-7, EIGHT times the rounding unit they were
// derived for: the bounds are tiny against a typical |f| (the audit's exclusion rate barely moves between 8x and 1x), so the margin is free.  Audited
// like curve_excluded: the audit build of the oracle evaluates this predicate next to the three edge tests for every visit of the level-line kernel,
// at 8x (shipped), at 1x (the derivation) and at 0.25x (a probe below it, which does find exclusions next to a crossing: the derivation is not slack
// by a large factor, the shipped form is) -- tests/test_edge_prefilter_audit.py.
__device__ __forceinline__ bool cell_excluded(V2 r0, V2 r1, V2 r2, float ha, float hb, float hc, float hd)
{
    const float E = 4.8e-7f;
    const float f10 = ha + hb, f01 = ha + hc, f11 = f10 + (hc + hd);
    const float fmn = __builtin_fminf(__builtin_fminf(ha, f10), __builtin_fminf(f01, f11)), fmx = __builtin_fmaxf(__builtin_fmaxf(ha, f10), __builtin_fmaxf(f01, f11));
    const float aa = __builtin_fabsf(ha), ab = __builtin_fabsf(hb), ac = __builtin_fabsf(hc), ad = __builtin_fabsf(hd);
    const float S = aa + ab + ac + ad;
    const float F = __builtin_fmaxf(fmn, -fmx) - 66.f * E * S;
    const float Rc = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(r0.x), __builtin_fabsf(r0.y)), __builtin_fmaxf(__builtin_fabsf(r1.x), __builtin_fabsf(r1.y))),
                                     __builtin_fmaxf(__builtin_fabsf(r2.x), __builtin_fabsf(r2.y)));
    bool ok = (Rc <= 64.f) & (F > 0.f);
    const V2 ea[3] = { r0, r1, r2 }, eb[3] = { r1, r2, r0 };
    #pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float dx = __builtin_fabsf(eb[e].x - ea[e].x), dy = __builtin_fabsf(eb[e].y - ea[e].y);
        const bool vertical = dx < 1e-6f;
        const float Kub = (dy * __builtin_amdgcn_rcpf(dx)) * 1.000002f, Klb = (dy * __builtin_amdgcn_rcpf(dx)) * 0.999998f;
        const float Mb = Rc * (1.f + Kub) * 1.000001f;
        const float C1b = (ac * Kub + ad * Mb + ab) * 1.000001f;
        const float C2b = (aa + ac * Mb) * 1.000001f;
        const float rest = 1.0001e-6f + E * (9.1f * C2b + 8.1f * C1b + 6.1f * ad * Kub + 2.02f * (ac + ad) * (Kub + 1.f)) + 1e-30f;
        const float c0lb = __builtin_fmaxf(ad * Klb * 0.999999f, 0.999999e-6f);
        const float Fe = F - rest;
        const bool sloped = (dx >= 1e-4f) & (Fe > 0.f) & ((dy == 0.f) | (Fe * c0lb > 1.01f * E * C1b * C1b));
        ok = ok & (vertical ? (F > 5.2f * E * S + 1e-30f) : sloped);
    }
    return ok;
}

// bake_kernels_cpu.h:241-399 : one texel of the bilinear footprint grid.  Adds to (above, below).
template <bool FP32, bool DEGENERATE, class MD>
__device__ __forceinline__ void level_line_texel(const ClassifyParams& P, const DevMip& m, const MicroTri& tIn, int px, int py,
                                                 uint32_t& above, uint32_t& below, const TexWindow& W)
{
    const float pfx = (float)px + 0.5f, pfy = (float)py + 0.5f;
    float gx, gy, gz, gw; // 00, 01, 11, 10
    fetch_cell<FP32, MD>(P, m, MD::pow2(P), px, py, W, gx, gy, gz, gw);
    if (!DEGENERATE) {
        const float ipx = pfx * m.rw, ipy = pfy * m.rh;
        const bool o0 = P.cutoff < gx, o1 = P.cutoff < gy, o2 = P.cutoff < gz, o3 = P.cutoff < gw;
        // Register pressure: the vertex values pass through an empty asm so that clang cannot hoist the per-texel products and
        // edge vectors out of the raster loops (LICM keeps ~27 more loop-invariant VGPRs live otherwise: 121 VGPRs = 4 waves/SIMD,
        // or 100+ bytes/lane of HBM-backed scratch when bounded).  With it the kernel needs 75 VGPRs, no scratch, 6 waves/SIMD
        // (68.7 -> 59.8 ms on the bench workload).  Values are unchanged -- the same fp32 expressions are evaluated per texel.
#ifndef OMMX_NO_LAUNDER
        MicroTri t = tIn;
        asm volatile("" : "+v"(t.p0.x), "+v"(t.p0.y), "+v"(t.p1.x), "+v"(t.p1.y), "+v"(t.p2.x), "+v"(t.p2.y));
        t.p0p2 = mk2(t.p0.x - t.p2.x, t.p0.y - t.p2.y); t.p1p0 = mk2(t.p1.x - t.p0.x, t.p1.y - t.p0.y); t.p2p1 = mk2(t.p2.x - t.p1.x, t.p2.y - t.p1.y);
#else
        const MicroTri& t = tIn;
#endif
        // (ipx is never a zero, so the reference's "+ 0.f" on the unchanged coordinate of each corner is the identity)
        const bool in0 = point_in_triangle_flat(t, ipx, ipy);
        const bool in1 = point_in_triangle_flat(t, ipx, ipy + m.rh);
        const bool in2 = point_in_triangle_flat(t, ipx + m.rw, ipy + m.rh);
        const bool in3 = point_in_triangle_flat(t, ipx + m.rw, ipy);
        const bool isO = (in0 && o0) || (in1 && o1) || (in2 && o2) || (in3 && o3);
        const bool isT = (in0 && !o0) || (in1 && !o1) || (in2 && !o2) || (in3 && !o3);
        if (isO) above += 1;
        if (isT) below += 1;
        if (isO && isT) return;
    }
    const float a = gx;
    const float b = gw - gx;
    const float c = gy - gx;
    const float d = gx + gz - gy - gw;
    if (near_zero(b, 1e-6f) && near_zero(c, 1e-6f) && near_zero(d, 1e-6f)) {
        vote(P.cutoff < a, above, below);
        return;
    }
    const float ha = a - P.cutoff;
    if (DEGENERATE) {
        const MicroTri& t = tIn;
        const V2 q0 = mk2(m.fw * t.lo.x - pfx, m.fh * t.lo.y - pfy);
        const V2 q1 = mk2(m.fw * t.hi.x - pfx, m.fh * t.hi.y - pfy);
        if (edge_crosses_level_curve(q0, q1, ha, b, c, d)) { above += 1; below += 1; }
        return;
    }
#ifndef OMMX_NO_LAUNDER
    MicroTri t = tIn;
    asm volatile("" : "+v"(t.p0.x), "+v"(t.p0.y), "+v"(t.p1.x), "+v"(t.p1.y), "+v"(t.p2.x), "+v"(t.p2.y));
#else
    const MicroTri& t = tIn;
#endif
    const V2 q0 = mk2(m.fw * t.p0.x - pfx, m.fh * t.p0.y - pfy);
    const V2 q1 = mk2(m.fw * t.p1.x - pfx, m.fh * t.p1.y - pfy);
    const V2 q2 = mk2(m.fw * t.p2.x - pfx, m.fh * t.p2.y - pfy);
    // (all three edges, no short-circuit: as in the single-texel pass, the divergent regions cost more than the tests they skip: 29.3 -> 28.5 ms)
    const bool x0 = edge_crosses_level_curve(q0, q1, ha, b, c, d), x1 = edge_crosses_level_curve(q1, q2, ha, b, c, d), x2 = edge_crosses_level_curve(q2, q0, ha, b, c, d);
    if (x0 | x1 | x2) {
        above += 1; below += 1;
    }
}

// bake_cpu_impl.cpp:994-1009
template <bool FP32, class MD>
__device__ __forceinline__ void nearest_texel(const ClassifyParams& P, const DevMip& m, int px, int py, uint32_t& above, uint32_t& below, const TexWindow& W)
{
    const int cx = tex_coord(MD::addr(P), MD::pow2(P), px, m.w, m.log2w), cy = tex_coord(MD::addr(P), MD::pow2(P), py, m.h, m.log2h);
    const bool border = MD::addr(P) == 3 && (cx == kTexCoordBorder || cy == kTexCoordBorder);
    const float alpha = border ? P.borderAlpha : load_texel<FP32>(m, cx, cy, W);
    vote(P.cutoff < alpha, above, below);
}

// bake_kernels_cpu.h:404-452 (ConservativeBilinearKernel, the reference's alternative to the level-line kernel behind internal flag bit 8): the four texels of
// the cell at int(pixel + 0.5) -- a conversion that truncates towards zero, so negative pixels take the cell one further in -- vote by their extremes
template <bool FP32, class MD>
__device__ __forceinline__ void conservative_bilinear_texel(const ClassifyParams& P, const DevMip& m, int px, int py, uint32_t& above, uint32_t& below, const TexWindow& W)
{
    const int ix = cvt_trunc_x86((float)px + 0.5f), iy = cvt_trunc_x86((float)py + 0.5f);
    const int x0 = tex_coord(MD::addr(P), MD::pow2(P), ix, m.w, m.log2w), y0 = tex_coord(MD::addr(P), MD::pow2(P), iy, m.h, m.log2h);
    const int x1 = tex_coord(MD::addr(P), MD::pow2(P), ix + 1, m.w, m.log2w), y1 = tex_coord(MD::addr(P), MD::pow2(P), iy + 1, m.h, m.log2h);
    float gx, gy, gz, gw;   // 00, 01, 11, 10
    if (MD::addr(P) == 3) {
        gx = load_texel_border<FP32>(m, x0, y0, P.borderAlpha, W); gy = load_texel_border<FP32>(m, x0, y1, P.borderAlpha, W);
        gz = load_texel_border<FP32>(m, x1, y1, P.borderAlpha, W); gw = load_texel_border<FP32>(m, x1, y0, P.borderAlpha, W);
    } else {
        gx = load_texel<FP32>(m, x0, y0, W); gy = load_texel<FP32>(m, x0, y1, W); gz = load_texel<FP32>(m, x1, y1, W); gw = load_texel<FP32>(m, x1, y0, W);
    }
    const float mn = std_min(std_min(std_min(gx, gy), gz), gw), mx = std_max(std_max(std_max(gx, gy), gz), gw);
    above += (P.cutoff < mx) ? 1u : 0u;
    below += (P.cutoff > mn) ? 1u : 0u;
}

// ---- conservative rasterisation of one micro-triangle (util/cpu_raster.h:20-52,117-124,277-383) ----
struct EdgeEq { float nx, ny, c, bias; };
__device__ __forceinline__ EdgeEq edge_eq(V2 p, V2 q)
{
    EdgeEq e;
    e.nx = q.y - p.y; e.ny = p.x - q.x;
    e.c = -(e.nx * p.x + e.ny * p.y);
    e.bias = 0.f; // unused; the conservative offsets are re-added in reference order below
    return e;
}
__device__ __forceinline__ float eval_cons(const EdgeEq& e, float sx, float sy)
{
    const float ev = (e.nx * sx + e.ny * sy) + e.c;
    const float bx = e.nx > 0 ? 0.f : e.nx;
    const float by = e.ny > 0 ? 0.f : e.ny;
    return ev + bx * 1.f + by * 1.f;
}

// KIND: 0 = level-line (linear filter), 1 = nearest vote, 2 = ConservativeBilinearKernel (internal flag bit 8)
template <bool FP32, int KIND, class MD>
__device__ __forceinline__ void raster_micro_triangle(const ClassifyParams& P, const DevMip& m, const MicroTri& t, float off,
                                                      uint32_t& above, uint32_t& below, const TexWindow& W)
{
    // util/geometry.h:49-55 : winding from the fp64 cross product of fp32 edge vectors
    const double ax = (double)(t.p2.x - t.p0.x), ay = (double)(t.p2.y - t.p0.y);
    const double bx = (double)(t.p1.x - t.p0.x), by = (double)(t.p1.y - t.p0.y);
    const bool ccw = (ax * by - bx * ay) < 0;
    V2 a = mk2(t.p0.x * m.fw + off, t.p0.y * m.fh + off);
    const V2 b = mk2(t.p1.x * m.fw + off, t.p1.y * m.fh + off);
    V2 c = mk2(t.p2.x * m.fw + off, t.p2.y * m.fh + off);
    if (!ccw) { V2 s = a; a = c; c = s; }
    const float lox = std_min(std_min(a.x, b.x), c.x), loy = std_min(std_min(a.y, b.y), c.y);
    const float hix = std_max(std_max(a.x, b.x), c.x), hiy = std_max(std_max(a.y, b.y), c.y);
    const int minx = cvt_trunc_x86(__builtin_floorf(lox)), miny = cvt_trunc_x86(__builtin_floorf(loy));
    const int maxx = cvt_trunc_x86(__builtin_ceilf(hix)), maxy = cvt_trunc_x86(__builtin_ceilf(hiy));
#ifdef OMMX_NO_LAUNDER
    const EdgeEq e0 = edge_eq(a, b), e1 = edge_eq(b, c), e2 = edge_eq(c, a);
#endif
    // Only the Nearest promotion looks at the counts (bake_kernels_cpu.h:38,49); for the forced promotions the state is
    // final as soon as both counters are non-zero, so the remaining texels cannot change the result.
    const bool countsMatter = P.promotion == 0;
    for (int y = miny; y < maxy; ++y) {
        bool wasInside = false;
        for (int x = minx; x < maxx; ++x) {
            const float sx = (float)x, sy = (float)y;
#ifndef OMMX_NO_LAUNDER
            // edge equations are re-derived per texel from the (opaque) vertices instead of living in 15 VGPRs across both loops
            // -- same expressions, same values; see the register-pressure note in level_line_texel()
            V2 t0 = t.p0, t1 = t.p1, t2 = t.p2;
            asm volatile("" : "+v"(t0.x), "+v"(t0.y), "+v"(t1.x), "+v"(t1.y), "+v"(t2.x), "+v"(t2.y));
            V2 la = mk2(t0.x * m.fw + off, t0.y * m.fh + off);
            const V2 lb = mk2(t1.x * m.fw + off, t1.y * m.fh + off);
            V2 lc = mk2(t2.x * m.fw + off, t2.y * m.fh + off);
            if (!ccw) { V2 sw = la; la = lc; lc = sw; }
            const EdgeEq e0 = edge_eq(la, lb), e1 = edge_eq(lb, lc), e2 = edge_eq(lc, la);
#endif
        const bool inside = eval_cons(e0, sx, sy) < 0.f && eval_cons(e1, sx, sy) < 0.f && eval_cons(e2, sx, sy) < 0.f;
            if (inside) {
                if (KIND == 0) level_line_texel<FP32, false, MD>(P, m, t, x, y, above, below, W);
                else if (KIND == 1) nearest_texel<FP32, MD>(P, m, x, y, above, below, W);
                else conservative_bilinear_texel<FP32, MD>(P, m, x, y, above, below, W);
                if (!countsMatter && above != 0 && below != 0) return;
                wasInside = true;
            } else if (wasInside) break;
        }
    }
}

// conservative line walk for degenerate work items (util/cpu_raster.h:486-555)
template <bool FP32, class MD>
__device__ __forceinline__ void raster_micro_segment(const ClassifyParams& P, const DevMip& m, const MicroTri& t,
                                                     uint32_t& above, uint32_t& below, const TexWindow& W)
{
    V2 p0 = mk2(t.lo.x * m.fw + -0.5f, t.lo.y * m.fh + -0.5f);
    V2 p1 = mk2(t.hi.x * m.fw + -0.5f, t.hi.y * m.fh + -0.5f);
    if (p0.x > p1.x) { V2 s = p0; p0 = p1; p1 = s; }
    const float dx = p1.x - p0.x, dy = p1.y - p0.y;
    int x = cvt_trunc_x86(__builtin_floorf(p0.x));
    int y = cvt_trunc_x86(__builtin_floorf(p0.y));
    const int stepX = dx > 0 ? 1 : (dx < 0 ? -1 : 0);
    const int stepY = dy > 0 ? 1 : (dy < 0 ? -1 : 0);
    const float inf = __builtin_inff();
    const float tDeltaX = stepX != 0 ? 1.f / __builtin_fabsf(dx) : inf;
    const float tDeltaY = stepY != 0 ? 1.f / __builtin_fabsf(dy) : inf;
    float tMaxX = inf, tMaxY = inf;
    if (stepX != 0) tMaxX = (((float)x + (stepX > 0 ? 1.f : 0.f)) - p0.x) / dx;
    if (stepY != 0) tMaxY = (((float)y + (stepY > 0 ? 1.f : 0.f)) - p0.y) / dy;
    if (stepX == 0 && stepY == 0) { level_line_texel<FP32, true, MD>(P, m, t, x, y, above, below, W); return; }
    const int yMin = cvt_trunc_x86(std_min(__builtin_floorf(p0.y), __builtin_floorf(p1.y)));
    const int yMax = cvt_trunc_x86(std_max(__builtin_ceilf(p0.y), __builtin_ceilf(p1.y)));
    const int xMin = cvt_trunc_x86(std_min(__builtin_floorf(p0.x), __builtin_floorf(p1.x)));
    const int xMax = cvt_trunc_x86(std_max(__builtin_ceilf(p0.x), __builtin_ceilf(p1.x)));
    const bool countsMatter = P.promotion == 0;
    while (x >= xMin && x <= xMax && y >= yMin && y <= yMax) {
        level_line_texel<FP32, true, MD>(P, m, t, x, y, above, below, W);
        if (!countsMatter && above != 0 && below != 0) return;
        if (tMaxX < tMaxY) { x += stepX; tMaxX += tDeltaX; }
        else { y += stepY; tMaxY += tDeltaY; }
    }
}

// ---- coarse pass: summed-area-table test of one micro-triangle (bake_cpu_impl.cpp:749-801) ----
// returns -1 when the micro-triangle stays unresolved
template <class MD>
__device__ __forceinline__ int coarse_state(const ClassifyParams& P, const MicroTri& t, const TexWindow& W)
{
    const DevMip& m = P.mips[0];
    if (cvt_trunc_x86(t.lo.x) != cvt_trunc_x86(t.hi.x) || cvt_trunc_x86(t.lo.y) != cvt_trunc_x86(t.hi.y)) return -1;
    const float fsx = t.lo.x * m.fw - 0.5f, fsy = t.lo.y * m.fh - 0.5f;
    const float fex = t.hi.x * m.fw - 0.5f, fey = t.hi.y * m.fh - 0.5f;
    const int sx = tex_coord(MD::addr(P), MD::pow2(P), cvt_trunc_x86(__builtin_floorf(fsx)), m.w, m.log2w);
    const int sy = tex_coord(MD::addr(P), MD::pow2(P), cvt_trunc_x86(__builtin_floorf(fsy)), m.h, m.log2h);
    const int ex = tex_coord(MD::addr(P), MD::pow2(P), cvt_trunc_x86(__builtin_floorf(fex)) + 1, m.w, m.log2w);
    const int ey = tex_coord(MD::addr(P), MD::pow2(P), cvt_trunc_x86(__builtin_floorf(fey)) + 1, m.h, m.log2h);
    if (ex < sx || ey < sy) return -1;
    if (!(sx >= 0 && sy >= 0 && sx < m.w && sy < m.h)) return -1;
    if (!(ex >= 0 && ey >= 0 && ex < m.w && ey < m.h)) return -1;
    const uint32_t area = (uint32_t)((ex - sx + 1) * (ey - sy + 1));
    const uint32_t sa = sat_sum(m, sx, sy, ex, ey, W);
    if (sa == 0) return P.stateLE;
    if (sa == area) return P.stateGT;
    return -1;
}

// coarse_state() for items under the FINITE precondition of the single-texel pass (every |uv| <= 16384: nothing is NaN, every value
// converted to int is below 2^31 in magnitude): the box comes from three-operand minima / maxima of the vertices (they differ from the
// reference's min / max chains only in the sign of a zero, which the truncation and the "* size - 0.5" erase), the conversions are
// plain, and the reference's four early exits are ONE predicate in front of the summed-area look-up.
template <class MD>
__device__ __forceinline__ int coarse_state_finite(const ClassifyParams& P, const MicroTri& t, const TexWindow& W)
{
    const DevMip& m = P.mips[0];
    const float lox = __builtin_fminf(__builtin_fminf(t.p0.x, t.p1.x), t.p2.x), loy = __builtin_fminf(__builtin_fminf(t.p0.y, t.p1.y), t.p2.y);
    const float hix = __builtin_fmaxf(__builtin_fmaxf(t.p0.x, t.p1.x), t.p2.x), hiy = __builtin_fmaxf(__builtin_fmaxf(t.p0.y, t.p1.y), t.p2.y);
    const bool sameTile = ((int)lox == (int)hix) & ((int)loy == (int)hiy);
    const float fsx = lox * m.fw - 0.5f, fsy = loy * m.fh - 0.5f;
    const float fex = hix * m.fw - 0.5f, fey = hiy * m.fh - 0.5f;
    const int sx = tex_coord(MD::addr(P), MD::pow2(P), (int)__builtin_floorf(fsx), m.w, m.log2w);
    const int sy = tex_coord(MD::addr(P), MD::pow2(P), (int)__builtin_floorf(fsy), m.h, m.log2h);
    const int ex = tex_coord(MD::addr(P), MD::pow2(P), (int)__builtin_floorf(fex) + 1, m.w, m.log2w);
    const int ey = tex_coord(MD::addr(P), MD::pow2(P), (int)__builtin_floorf(fey) + 1, m.h, m.log2h);
    // unsigned compares fold "0 <= v" and "v < size" into one test each
    const bool ok = sameTile & !(ex < sx) & !(ey < sy) & ((uint32_t)sx < (uint32_t)m.w) & ((uint32_t)sy < (uint32_t)m.h) & ((uint32_t)ex < (uint32_t)m.w) & ((uint32_t)ey < (uint32_t)m.h);
    int st = -1;
    if (ok) {
        const uint32_t area = (uint32_t)((ex - sx + 1) * (ey - sy + 1));
        const uint32_t sa = sat_sum(m, sx, sy, ex, ey, W);
        st = sa == 0 ? P.stateLE : (sa == area ? P.stateGT : -1);
    }
    return st;
}

// ---- hierarchical shortcut: is a whole bird-curve sub-triangle provably resolved by the coarse pass? ----
// `sub` is an ancestor (level k <= N) of micro-triangles of one work item; `maxAbs` = largest |coordinate| of the item's
// vertices.  Returns the state every descendant micro-triangle gets from coarse_state(), or -1 when that cannot be
// guaranteed.  Argument: every descendant's fp32 vertices lie within 12 ulp(maxAbs) of the
// ancestor's fp32 AABB, so with the AABB grown by 64 ulp each descendant's (a) UV-tile test, (b) texel rectangle
// [floor(lo*W-.5), floor(hi*W-.5)+1] -- all operations monotone -- is contained in the ancestor's; if the address mode maps
// the ancestor's rectangle without a seam and the SAT says the rectangle is uniformly <= / > cutoff, so does it for every
// sub-rectangle.  States that the reference's fine pass would revisit (value 3, bake_cpu_impl.cpp:861) are not shortcut.
struct TexRect { int sx, sy, ex, ey; bool ok; };

// addressed texel rectangle that contains the coarse-pass rectangle of every descendant of `sub`; ok == false when that
// cannot be guaranteed (huge coordinates, UV-tile crossing, address-mode seam)
template <class MD>
__device__ __forceinline__ TexRect region_rect(const ClassifyParams& P, const MicroTri& sub, float maxAbs)
{
    const DevMip& m = P.mips[0];
    TexRect r; r.sx = r.sy = r.ex = r.ey = 0; r.ok = false;
    if (!(maxAbs <= 16384.f)) return r;
    const float grow = maxAbs * 7.62939453125e-06f + 1e-30f; // 2^-17 * maxAbs  (= 64 ulp)
    const float lx = sub.lo.x - grow, ly = sub.lo.y - grow, hx = sub.hi.x + grow, hy = sub.hi.y + grow;
    if (cvt_trunc_x86(lx) != cvt_trunc_x86(hx) || cvt_trunc_x86(ly) != cvt_trunc_x86(hy)) return r;
    const int X0 = cvt_trunc_x86(__builtin_floorf(lx * m.fw - 0.5f)), Y0 = cvt_trunc_x86(__builtin_floorf(ly * m.fh - 0.5f));
    const int X1 = cvt_trunc_x86(__builtin_floorf(hx * m.fw - 0.5f)) + 1, Y1 = cvt_trunc_x86(__builtin_floorf(hy * m.fh - 0.5f)) + 1;
    if (X1 - X0 >= m.w || Y1 - Y0 >= m.h) return r;
    if (MD::addr(P) == 0) { // Wrap: any single period is fine as long as the rectangle does not cross the seam
        r.sx = tex_coord(0, MD::pow2(P), X0, m.w, m.log2w); r.ex = tex_coord(0, MD::pow2(P), X1, m.w, m.log2w);
        r.sy = tex_coord(0, MD::pow2(P), Y0, m.h, m.log2h); r.ey = tex_coord(0, MD::pow2(P), Y1, m.h, m.log2h);
        if (r.ex - r.sx != X1 - X0 || r.ey - r.sy != Y1 - Y0) return r;
    } else {               // other modes: only the untouched interior [0,W) x [0,H)
        if (X0 < 0 || Y0 < 0 || X1 >= m.w || Y1 >= m.h) return r;
        r.sx = X0; r.sy = Y0; r.ex = X1; r.ey = Y1;
    }
    r.ok = true;
    return r;
}

constexpr int kRegionEdgeFreeBase = 0x100;   // region_curve_state(): -(base + 16 above + mask) = "inside ONE cell, every vote on that side except, possibly, the
                                             // corner votes of the cell corners in `mask` (wrong side): test PointInTriangle for those per micro-triangle"
constexpr int kRegionUnknown = -1;     // descendants must be tested one by one
constexpr int kRegionAllOpen = -2;     // EVERY descendant stays unresolved in the coarse pass (region_state_ex only)

template <class MD, bool WANT_OPEN>
__device__ __forceinline__ int region_state_impl(const ClassifyParams& P, const MicroTri& sub, float maxAbs, const TexWindow& W)
{
    const TexRect r = region_rect<MD>(P, sub, maxAbs);
    if (!r.ok) return kRegionUnknown;
    const uint32_t area = (uint32_t)((r.ex - r.sx + 1) * (r.ey - r.sy + 1));
    const uint32_t sa = sat_sum(P.mips[0], r.sx, r.sy, r.ex, r.ey, W);
    int st = kRegionUnknown;
    if (sa == 0) st = P.stateLE; else if (sa == area) st = P.stateGT;
    // The converse shortcut.  A descendant's texel rectangle is at least 2 x 2 ([floor(lo), floor(hi) + 1], hi >= lo) and is contained
    // in the ancestor's; when the ancestor's rectangle is itself 2 x 2 the two coincide, so a non-uniform SAT answer for the
    // ancestor is the answer of every descendant: none of them is resolved by the coarse pass (their other early-outs,
    // bake_cpu_impl.cpp:760-775, also leave them unresolved) and all go to the level-line pass without being tested.
    else if (WANT_OPEN && area == 4u) return kRegionAllOpen;
    return st == 3 ? kRegionUnknown : st;
}
template <class MD>
__device__ __forceinline__ int region_state(const ClassifyParams& P, const MicroTri& sub, float maxAbs, const TexWindow& W)
{
    return region_state_impl<MD, false>(P, sub, maxAbs, W);
}
template <class MD>
__device__ __forceinline__ int region_state_ex(const ClassifyParams& P, const MicroTri& sub, float maxAbs, const TexWindow& W)
{
    return region_state_impl<MD, true>(P, sub, maxAbs, W);
}

// ---- hierarchical shortcut of the FINE pass: a bird-curve sub-triangle the level curve provably cannot reach (region_curve.h) ----
// `sub` is an ancestor (any level <= N, the micro-triangle itself included) of micro-triangles of a NON-DEGENERATE work item with the vertices `uv` and the
// subdivision level `level`; `sh` = rc_shape() of that item.  Returns the state every descendant ends with -- whichever of the reference's passes classifies
// it: the votes of ResampleFine all fall on one side of the cutoff, and a descendant that ResampleCoarse resolves has all its texels on that side --
// or -1.  Independent of the summed-area table: it also culls bakes of textures without one.  Texels come from HBM / L2 (at most 5 x 5 per call).
__device__ __forceinline__ bool region_curve_applies(const ClassifyParams& P) { return P.filterLinear != 0 && P.mipCount == 1 && P.noFine == 0 && P.altKernel == 0; }
struct RcTex { const void* texels; int w, h; float fw, fh; int addr, pow2, fp32; float cutoff; int stateGT, stateLE; };   // what the test reads of ClassifyParams
template <class MD>
__device__ __forceinline__ RcTex rc_tex(const ClassifyParams& P, bool fp32)
{
    const DevMip& m = P.mips[0];
    RcTex t; t.texels = m.texels; t.w = m.w; t.h = m.h; t.fw = m.fw; t.fh = m.fh; t.addr = MD::addr(P); t.pow2 = MD::pow2(P); t.fp32 = fp32 ? 1 : 0;
    t.cutoff = P.cutoff; t.stateGT = P.stateGT; t.stateLE = P.stateLE;
    return t;
}
__device__ __forceinline__ int region_curve_state_impl(const RcTex& T, const RcShape& sh, float lox, float loy, float hix, float hiy, float maxAbs)
{
    if (!sh.ok) return -1;
    const RcFrame f = rc_frame(lox, loy, hix, hiy, maxAbs, T.fw, T.fh, T.w, T.h, T.addr, T.pow2);
    if (!f.ok) return -1;
    int sign = 0; uint32_t wrong = 0;
    for (int j = 0; j < f.ny; ++j)
        for (int i = 0; i < f.nx; ++i) {
            const size_t i00 = (size_t)(f.sx + i) + (size_t)(f.sy + j) * (size_t)T.w, i01 = i00 + (size_t)T.w;
            float g00, g10, g01, g11;
            if (T.fp32) { const float* t = (const float*)T.texels; g00 = t[i00]; g10 = t[i00 + 1]; g01 = t[i01]; g11 = t[i01 + 1]; }
            else { const uint8_t* t = (const uint8_t*)T.texels; g00 = (float)t[i00] * (1.f / 255.f); g10 = (float)t[i00 + 1] * (1.f / 255.f); g01 = (float)t[i01] * (1.f / 255.f); g11 = (float)t[i01 + 1] * (1.f / 255.f); }
            const int c = rc_cell(&sh, g00, g10, g01, g11, T.cutoff, f.bx0 - (float)i, f.bx1 - (float)i, f.by0 - (float)j, f.by1 - (float)j);
            if (c == 0) return -1;
            if (c == 2) continue;
            if (sign != 0 && sign != c) return -1;
            sign = c;
            // corners of this cell whose texel is on the other side, in the order the level-line kernel tests them: (x, y), (x, y+1), (x+1, y+1), (x+1, y)
            const uint32_t above = (T.cutoff < g00 ? 1u : 0u) | (T.cutoff < g01 ? 2u : 0u) | (T.cutoff < g11 ? 4u : 0u) | (T.cutoff < g10 ? 8u : 0u);
            wrong = c > 0 ? (~above & 15u) : above;
        }
    if (sign == 0) return -1;
    const int st = sign > 0 ? T.stateGT : T.stateLE;
    if (st == 3) return -1;   // (as region_state(): the value the reference's fine pass would revisit is not shortcut)
    if (sh.fat) return st;
    // the corner bound does not hold for this work item (RcShape::fat): edges and centre vote are settled, the corner votes of the other side are not.  For
    // a sub-triangle inside ONE cell the caller can finish the job per micro-triangle (PointInTriangle of the cell's wrong-side corners): kRegionEdgeFree code
    if (f.nx != 1 || f.ny != 1) return -1;
    return -(kRegionEdgeFreeBase + (sign > 0 ? 16 : 0) + (int)wrong);
}
template <class MD>
__device__ __forceinline__ int region_curve_state(const ClassifyParams& P, bool fp32, const RcShape& sh, const MicroTri& sub, float maxAbs)
{
    return region_curve_state_impl(rc_tex<MD>(P, fp32), sh, sub.lo.x, sub.lo.y, sub.hi.x, sub.hi.y, maxAbs);
}
// the same as a real call: inside the persistent classify_tiles kernel the test runs once per 64-group in ONE wave of the tile's set-up phase, and inlined it
// costs the whole kernel registers (25 -> 49 spilled VGPRs); as a callee it has an allocation of its own
__device__ __attribute__((noinline)) int region_curve_state_call(RcTex T, RcShape sh, float lox, float loy, float hix, float hiy, float maxAbs)
{
    return region_curve_state_impl(T, sh, lox, loy, hix, hiy, maxAbs);
}

// ---- fine pass of a micro-triangle whose conservative raster covers ONE texel of mip 0 (Linear filter, non-degenerate item) ----
// At the bench configuration 95 % of the micro-triangles that reach the level-line pass are far smaller than a texel (level 8 on an
// 8-texel triangle: 1/32 texel), so their raster bounding box [floor(lo), ceil(hi)) is a single texel and that texel is also the cell
// of the centre vote.  For those the generic code (fine_state -> bilinear + raster_micro_triangle -> level_line_texel) degenerates to
// a fixed sequence; this function is that sequence written straight-line: no texel loops, one 2x2 texel fetch shared by the centre
// vote and the level-line kernel, the raster-space vertices Q = P * size - 0.5 computed once.  Every fp32 expression is the one the
// generic path evaluates (same operands, same association), so the state is bit-identical; micro-triangles that do not fit the pattern
// return -1 and take the generic path.
//   * Q: the centre vote uses `p * size - 0.5f` (texture_impl.cpp:264), the rasteriser `p * size + (-0.5f)` (cpu_raster.h:298-299):
//     the same IEEE operation.
//   * winding: sign of the fp64 difference of two products of fp32 values (geometry.h:49-55).  The products are exact in fp64, and
//     rounding to fp32 is monotone, so whenever the fp32-rounded products differ their order IS the exact order; fp64 is evaluated
//     only when they round to the same float.
// The three edge tests are NOT evaluated here: curve_excluded() settles 88 % of the micro-triangles that get that far; for the others the answer is
// kNeedsEdges | (above >= below) << 2 | the state without a crossing, and single_texel_edges() below finishes them in a second, compacted pass.
constexpr int kNeedsEdges = 0x80;
template <bool FP32, class MD>
// `shape`: three words of the work item's RcShape (region_curve.h: rhoX, rhoY as floats, ok & fat as a word), in LDS.
__device__ __forceinline__ int fine_single_texel(const ClassifyParams& P, const MicroTri& t, const TexWindow& W, const lds_u32* shape)
{
    const DevMip& m = P.mips[0];
    const float q0x = t.p0.x * m.fw - 0.5f, q0y = t.p0.y * m.fh - 0.5f;
    const float q1x = t.p1.x * m.fw - 0.5f, q1y = t.p1.y * m.fh - 0.5f;
    const float q2x = t.p2.x * m.fw - 0.5f, q2y = t.p2.y * m.fh - 0.5f;
    // winding (util/geometry.h:49-55): ccw = (double)(p2-p0).x * (double)(p1-p0).y - (double)(p1-p0).x * (double)(p2-p0).y < 0
    const float ax = t.p2.x - t.p0.x, ay = t.p2.y - t.p0.y, bx = t.p1.x - t.p0.x, by = t.p1.y - t.p0.y;
    const float l32 = ax * by, r32 = bx * ay;
    bool ccw = l32 < r32;
    const bool tie = !(l32 < r32) && !(r32 < l32);   // equal or unordered in fp32: decide in fp64 like the reference
    if (__any(tie)) {                                // (wave-uniform branch: the fp64 sequence is skipped by all but a few waves)
        const bool ccw64 = ((double)ax * (double)by - (double)bx * (double)ay) < 0;
        ccw = tie ? ccw64 : ccw;
    }
    V2 a = mk2(q0x, q0y); const V2 b = mk2(q1x, q1y); V2 c = mk2(q2x, q2y);
    if (!ccw) { V2 s = a; a = c; c = s; }
    // FINITE (the caller's guarantee: every |uv| of the item <= 16384, so |Q| <= 16384 * 65536 + 0.5 < 2^31 and nothing is NaN): the
    // min(min(a, b), c) chains of the reference reduce to three-operand minima / maxima (v_min3_f32 / v_max3_f32) -- the two differ only for
    // NaN operands and in the sign of a zero, which floor / ceil -> int erases -- and none of the six float -> int conversions below can
    // leave the int range, so the x86 "integer indefinite" rule of cvt_trunc_x86() cannot apply and the plain conversion gives the same integers
    const float lox = __builtin_fminf(__builtin_fminf(a.x, b.x), c.x), loy = __builtin_fminf(__builtin_fminf(a.y, b.y), c.y);
    const float hix = __builtin_fmaxf(__builtin_fmaxf(a.x, b.x), c.x), hiy = __builtin_fmaxf(__builtin_fmaxf(a.y, b.y), c.y);
    const int minx = (int)__builtin_floorf(lox), miny = (int)__builtin_floorf(loy);
    const int maxx = (int)__builtin_ceilf(hix), maxy = (int)__builtin_ceilf(hiy);
    const float fx = __builtin_floorf(q0x), fy = __builtin_floorf(q0y);
    const int ix = (int)fx, iy = (int)fy;
    // one texel, which is also the centre-vote cell
    if (!(maxx - minx == 1 && maxy - miny == 1 && ix == minx && iy == miny)) return -1;

    float g00, g01, g11, g10;
    fetch_cell<FP32, MD>(P, m, MD::pow2(P), minx, miny, W, g00, g01, g11, g10);
    // (TextureImpl::Bilinear addresses with the per-mip pow2 flag, the level-line kernel with the dispatch flag = mip 0's: the same here)
    uint32_t above = 0, below = 0; bool needsEdges = false;
    {   // centre vote: TextureImpl::Bilinear at p0 (texture_impl.cpp:261-278); a = 00, b = 01, c = 10, d = 11
        const float wx = q0x - fx, wy = q0y - fy;
        const float ac = g00 * (1.f - wx) + g10 * wx;
        const float bd = g01 * (1.f - wx) + g11 * wx;
        vote(P.cutoff < ac * (1.f - wy) + bd * wy, above, below);
    }
    // conservative coverage test of the texel (cpu_raster.h:309-335 at x = minx, y = miny)
    const float sx = (float)minx, sy = (float)miny;
    const EdgeEq e0 = edge_eq(a, b), e1 = edge_eq(b, c), e2 = edge_eq(c, a);
    const bool inside = eval_cons(e0, sx, sy) < 0.f && eval_cons(e1, sx, sy) < 0.f && eval_cons(e2, sx, sy) < 0.f;
    if (inside) {
        // LevelLineIntersectionKernel::run (bake_kernels_cpu.h:241-399), as level_line_texel<.., false, ..> above
        const float pfx = sx + 0.5f, pfy = sy + 0.5f;
        const float ipx = pfx * m.rw, ipy = pfy * m.rh;
        const bool o0 = P.cutoff < g00, o1 = P.cutoff < g01, o2 = P.cutoff < g11, o3 = P.cutoff < g10;
        // The four corner votes.  A micro-triangle of this pass is a small fraction of a texel: a corner of its cell lies inside it once in a thousand.  When
        // the work item has the corner bound and all four corners are outside the micro-triangle's box fattened by rho, PointInTriangle cannot report
        // one inside (rc_corners_far, region_curve.h: audited on every cell visit of the oracle's level-line kernel) -- the four tests, 100 of this pass's 460
        // vector instructions, run only in waves where a lane is near a corner.
        bool isO = false, isT = false;
#ifdef OMMX_EXP_SKIP_ALL_CORNERS   // (timing experiment: wrong results)
        const bool nearCorner = false;
#elif !defined(OMMX_NO_CORNER_SKIP)
        bool nearCorner = true;
        if (shape[2] != 0u) {   // (wave-uniform: the chunk's work item has the corner bound)
            // the vertices in the cell's coordinates, as the level-line kernel forms them (bake_kernels_cpu.h:378-379): their box is rc_corners_far()'s frame
            const float r0x = m.fw * t.p0.x - pfx, r0y = m.fh * t.p0.y - pfy, r1x = m.fw * t.p1.x - pfx, r1y = m.fh * t.p1.y - pfy, r2x = m.fw * t.p2.x - pfx, r2y = m.fh * t.p2.y - pfy;
            RcShape sh; sh.Kub = 0.f; sh.Klb = 0.f; sh.rhoX = __uint_as_float(shape[0]); sh.rhoY = __uint_as_float(shape[1]); sh.ok = sh.fat = 1;
            nearCorner = !rc_corners_far(&sh, __builtin_fminf(__builtin_fminf(r0x, r1x), r2x), __builtin_fmaxf(__builtin_fmaxf(r0x, r1x), r2x),
                                         __builtin_fminf(__builtin_fminf(r0y, r1y), r2y), __builtin_fmaxf(__builtin_fmaxf(r0y, r1y), r2y));
        }
#else
        const bool nearCorner = true;   // (A/B builds: the four tests for every micro-triangle, as in rounds 1 - 5)
#endif
        if (__any(nearCorner)) {   // (wave-uniform branch; the tests themselves stay straight-line code)
            // (ipx is never a zero, so the reference's "+ 0.f" on the unchanged coordinate of each corner is the identity)
            const bool in0 = point_in_triangle_flat(t, ipx, ipy);
            const bool in1 = point_in_triangle_flat(t, ipx, ipy + m.rh);
            const bool in2 = point_in_triangle_flat(t, ipx + m.rw, ipy + m.rh);
            const bool in3 = point_in_triangle_flat(t, ipx + m.rw, ipy);
            isO = nearCorner && ((in0 && o0) || (in1 && o1) || (in2 && o2) || (in3 && o3));
            isT = nearCorner && ((in0 && !o0) || (in1 && !o1) || (in2 && !o2) || (in3 && !o3));
        }
        if (isO) above += 1;
        if (isT) below += 1;
        if (!(isO && isT)) {
            const float sa = g00, sb = g10 - g00, sc = g01 - g00, sd = g00 + g11 - g01 - g10;
            if (near_zero(sb, 1e-6f) && near_zero(sc, 1e-6f) && near_zero(sd, 1e-6f)) vote(P.cutoff < sa, above, below);
            else {
                const V2 r0 = mk2(m.fw * t.p0.x - pfx, m.fh * t.p0.y - pfy);
                const V2 r1 = mk2(m.fw * t.p1.x - pfx, m.fh * t.p1.y - pfy);
                const V2 r2 = mk2(m.fw * t.p2.x - pfx, m.fh * t.p2.y - pfy);
                needsEdges = !curve_excluded(r0, r1, r2, sa - P.cutoff, sb, sc, sd);   // (88 % of the micro-triangles are provably not touched by the level curve)
            }
        }
    }
    const int st = state_from_coverage(P, above, below);
    return needsEdges ? (kNeedsEdges | (above >= below ? 4 : 0) | st) : st;
}

// Second pass for the micro-triangles fine_single_texel() answered kNeedsEdges for (`code` = its answer): the three edge tests of
// LevelLineIntersectionKernel::run (bake_kernels_cpu.h:376-396) in the single texel of the micro-triangle, from the same expressions as there.
template <bool FP32, class MD>
__device__ __forceinline__ int single_texel_edges(const ClassifyParams& P, const MicroTri& t, const TexWindow& W, int code)
{
    const DevMip& m = P.mips[0];
    // the texel is the centre-vote cell (the first pass checked that): floor(p0 * size - 0.5)
    const int minx = (int)__builtin_floorf(t.p0.x * m.fw - 0.5f), miny = (int)__builtin_floorf(t.p0.y * m.fh - 0.5f);
    float g00, g01, g11, g10;
    fetch_cell<FP32, MD>(P, m, MD::pow2(P), minx, miny, W, g00, g01, g11, g10);
    const float pfx = (float)minx + 0.5f, pfy = (float)miny + 0.5f;
    const float sb = g10 - g00, sc = g01 - g00, sd = g00 + g11 - g01 - g10, ha = g00 - P.cutoff;
    const V2 r0 = mk2(m.fw * t.p0.x - pfx, m.fh * t.p0.y - pfy);
    const V2 r1 = mk2(m.fw * t.p1.x - pfx, m.fh * t.p1.y - pfy);
    const V2 r2 = mk2(m.fw * t.p2.x - pfx, m.fh * t.p2.y - pfy);
    // all three edges are evaluated by all lanes: the reference stops at the first crossing edge, but crossings are rare and the lanes of a wave do not
    // agree on them, so the short-circuit only adds divergent regions (33.1 -> 32.1 ms)
    const bool x0 = edge_crosses_level_curve(r0, r1, ha, sb, sc, sd), x1 = edge_crosses_level_curve(r1, r2, ha, sb, sc, sd), x2 = edge_crosses_level_curve(r2, r0, ha, sb, sc, sd);
    // a crossing adds one to both counters: both non-zero, and (above + 1 >= below + 1) == (above >= below)
    return (x0 | x1 | x2) ? state_from_coverage(P, (code & 4) ? 2u : 1u, (code & 4) ? 1u : 2u) : (code & 3);
}

// ---- fine pass for one micro-triangle (bake_cpu_impl.cpp:859-914 linear, :983-1022 nearest) ----
template <bool FP32, class MD>
__device__ __forceinline__ int fine_state(const ClassifyParams& P, const MicroTri& t, bool degenerate, const TexWindow& W)
{
    uint32_t above = 0, below = 0;
    if (P.filterLinear && P.altKernel) {   // bake_cpu_impl.cpp:915-966: no centre vote, mip 0 only, degenerate items like the others
        const DevMip& m = P.mips[0];
        if (P.altKernel == 2) {            // EnableAABBTesting: the two triangles of the micro-triangle's bounding box
            MicroTri t0, t1;
            t0.p0 = t.lo; t0.p1 = mk2(t.hi.x, t.lo.y); t0.p2 = mk2(t.lo.x, t.hi.y); finish_tri(t0);
            t1.p0 = t.hi; t1.p1 = mk2(t.hi.x, t.lo.y); t1.p2 = mk2(t.lo.x, t.hi.y); finish_tri(t1);
            raster_micro_triangle<FP32, 2, MD>(P, m, t0, -0.5f, above, below, W);
            raster_micro_triangle<FP32, 2, MD>(P, m, t1, -0.5f, above, below, W);
        } else raster_micro_triangle<FP32, 2, MD>(P, m, t, -0.5f, above, below, W);
        return state_from_coverage(P, above, below);
    }
    for (int mip = 0; mip < P.mipCount; ++mip) {
        const DevMip& m = P.mips[mip];
        if (P.filterLinear) {
            vote(P.cutoff < bilinear<FP32, MD>(P, m, t.p0, W), above, below);
            if (!degenerate) raster_micro_triangle<FP32, 0, MD>(P, m, t, -0.5f, above, below, W);
            else raster_micro_segment<FP32, MD>(P, m, t, above, below, W);
        } else {
            raster_micro_triangle<FP32, 1, MD>(P, m, t, 0.f, above, below, W);
        }
        if (state_is_unknown(state_from_coverage(P, above, below))) break;
    }
    return state_from_coverage(P, above, below);
}

} // namespace ommx
