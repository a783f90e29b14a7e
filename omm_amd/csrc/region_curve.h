// region_curve.h -- "curve-free region" test: can the level curve alpha == cutoff reach ANY micro-triangle of a bird-curve sub-triangle?
//
// The reference classifies a micro-triangle (Linear filter, one mip, non-degenerate work item) from votes (bake_cpu_impl.cpp:859-914,
// bake_kernels_cpu.h:241-399): the centre vote (bilinear sample at p0 against the cutoff), per visited texel cell the votes of the cell
// corners that lie inside the micro-triangle, the vote of a flat cell's first texel, and "both sides" when one of the three edges crosses the
// level curve of the cell's bilinear patch f(x, y) = ha + hb x + hc y + hd x y inside the unit cell.  If, for EVERY cell a sub-triangle's
// micro-triangles can touch, |f| stays above the error bounds below on the part D of that cell the (fattened) sub-triangle can reach, and the
// sign of f there is the same in all those cells, then every one of those votes falls on that side: every descendant micro-triangle ends with
// the pure state of that side, whatever path (coarse or fine) the reference takes for it.  One test settles 64, 4096 or 4^N micro-triangles.
//
// Why each vote is on side s (D = sub-triangle's box in the cell's coordinates, fattened by rho, clipped to [-0.02, 1.02]^2):
//   edges    curve_excluded()'s theorem (classify_device.h) with D in the place of the micro-triangle's own fattened box: an accepted
//            root lies in the unit cell, within 4.9e-3 of its edge (inside D) and is an approximate zero of f (|f| <= rest there).  The slopes that enter
//            `rest` are bounded for ALL micro-triangles of the work item at once: their edges are +-2^-N times the item's edges, each component off by at
//            most 8 u (u = size x maxAbs x 2^-24, half a unit in the last place of a raster coordinate).
//   centre   the sample point lies in D (within 2 u of the vertex r0); its fp32 evaluation differs from f + cutoff by < 6 e (sum |texel| + |cutoff|).
//   corners  a cell corner inside D has f's sign there (it IS a corner of the unit cell, so it survives the clip).  A corner outside D is at least
//            g = 1/64 texel away from every micro-triangle; PointInTriangle's three fp32 cross products (util/geometry.h:101-114) are each within
//            8 e (|a1 b1| + |a2 b2|) of their exact value, and an outside point at distance >= g from a triangle whose smallest angle has
//            sin >= 0.005 has two cross products of opposite exact sign that are both larger than that: the function returns false.
//   flat     a flat cell (|hb|, |hc|, |hd| < 1e-6) votes by its first texel: |f - ha| < 3.1e-6 on the cell, so min |f| > 2e-5 on D fixes sign(ha).
//   coarse   a micro-triangle the summed-area-table pass resolves has all texels of its cells on one side: f has that sign on the whole cell.
// Audited, not only argued: the audit build of the oracle (oracle/Makefile: libomm_oracle_audit.so) includes THIS header, evaluates the test for every
// sub-triangle of every level of every work item it bakes and compares with the states the reference algorithm produced for the descendants
// (tests/test_region_curve_audit.py).  Plain C99 / C++: no HIP types, so the device code and the oracle compile the same expressions (no FMA
// contraction on either side: -ffp-contract=off).
#ifndef OMMX_REGION_CURVE_H
#define OMMX_REGION_CURVE_H
#include <stdint.h>

/* Bit-exactness of the verdicts between the device code and the audited host compilation rests on both evaluating the SAME fp32 expressions: no FMA
 * contraction, no fast-math.  Both Makefiles pass -ffp-contract=off (omm_amd/csrc/Makefile, oracle/Makefile); for clang (hipcc: host and device side) the
 * pragma below pins it in the source as well, whatever flags a different build passes. */
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
#if defined(__FAST_MATH__)
#error "region_curve.h must not be compiled with -ffast-math: its error bounds assume IEEE fp32 evaluation"
#endif
#if defined(__HIPCC__)
#define OMMX_RC_FN __host__ __device__ static inline
#else
#define OMMX_RC_FN static inline
#endif

typedef struct RcShape { float Kub, Klb, rhoX, rhoY; int ok, fat; } RcShape;            /* per work item (level included).  ok: the edge / centre-vote bounds hold; fat: the corner (point-in-triangle) bound too */
typedef struct RcFrame { int X0, Y0, sx, sy, nx, ny, ok; float bx0, bx1, by0, by1; } RcFrame;   /* per sub-triangle: cells [X0, X0 + nx) x [Y0, Y0 + ny) */

OMMX_RC_FN float rc_abs(float v) { return v < 0.f ? -v : v; }
OMMX_RC_FN float rc_min(float a, float b) { return b < a ? b : a; }
OMMX_RC_FN float rc_max(float a, float b) { return a < b ? b : a; }
OMMX_RC_FN int rc_trunc(float f) { return (f >= -2147483648.f && f < 2147483648.f) ? (int)f : (int)0x80000000; }   /* cvttss2si */
OMMX_RC_FN float rc_floor(float f) { return __builtin_floorf(f); }

#define OMMX_RC_MAX_CELLS 4   /* cells per axis a sub-triangle may touch (5 x 5 texels) */

/* The fattening rho of a work item, per axis (rc_shape() below stores it; classify_tiles recomputes it per chunk from the record's coordinates): 6e-3 of
 * curve_excluded()'s ellipse argument + 22 u of vertex rounding + g = 1/64 texel of the corner bound, u = size x maxAbs x 2^-24. */
OMMX_RC_FN void rc_rho(const float* uv, float fw, float fh, float* ux, float* uy, float* rhoX, float* rhoY)
{
    const float mx = rc_max(rc_max(rc_abs(uv[0]), rc_abs(uv[2])), rc_abs(uv[4])), my = rc_max(rc_max(rc_abs(uv[1]), rc_abs(uv[3])), rc_abs(uv[5]));
    *ux = fw * mx * 5.9604645e-8f; *uy = fh * my * 5.9604645e-8f;
    *rhoX = 6e-3f + 22.f * *ux + 0.015625f; *rhoY = 6e-3f + 22.f * *uy + 0.015625f;
}

/* Slope and shape bounds shared by every micro-triangle of a work item (uv: its six floats; level: its subdivision level; w, h: texture size). */
OMMX_RC_FN RcShape rc_shape(const float* uv, float fw, float fh, int w, int h, uint32_t level)
{
    RcShape s; s.Kub = 0.f; s.Klb = 0.f; s.rhoX = 0.f; s.rhoY = 0.f; s.ok = 0; s.fat = 0;
    /* u: bound of ONE rounding of a raster-space coordinate of this item (2^-24 relative), per axis */
    float ux, uy, rhoX, rhoY;
    rc_rho(uv, fw, fh, &ux, &uy, &rhoX, &rhoY);
    if (!(ux <= 1e-3f && uy <= 1e-3f)) return s;
    const float sc = 1.f / (float)(1u << (level > 12u ? 12u : level));
    const float ex0 = fw * (uv[2] - uv[0]) * sc, ey0 = fh * (uv[3] - uv[1]) * sc;
    const float ex1 = fw * (uv[4] - uv[2]) * sc, ey1 = fh * (uv[5] - uv[3]) * sc;
    const float ex2 = fw * (uv[0] - uv[4]) * sc, ey2 = fh * (uv[1] - uv[5]) * sc;
    /* a vertex of a micro-triangle, in raster units, is within 3 u of its exact place (three rounded products and two rounded sums of the barycentric
     * interpolation, util/geometry.h:241-248; the weights are exact), 4 u when the scaling by a non-power-of-two size rounds once more; the subtraction
     * of the cell origin is exact up to half an ulp of a value below 4.  An edge component is a difference of two such coordinates. */
    const float dX = ((w & (w - 1)) ? 8.f : 6.f) * ux + 5e-7f, dY = ((h & (h - 1)) ? 8.f : 6.f) * uy + 5e-7f;
    const float ax0 = rc_abs(ex0), ax1 = rc_abs(ex1), ax2 = rc_abs(ex2), ay0 = rc_abs(ey0), ay1 = rc_abs(ey1), ay2 = rc_abs(ey2);
    const float dxmin = rc_min(rc_min(ax0, ax1), ax2) * 0.99999f - dX, dxmax = rc_max(rc_max(ax0, ax1), ax2) * 1.00001f + dX;
    const float dymin = rc_max(rc_min(rc_min(ay0, ay1), ay2) * 0.99999f - dY, 0.f), dymax = rc_max(rc_max(ay0, ay1), ay2) * 1.00001f + dY;
    if (!(dxmax <= 1.f && dymax <= 1.f)) return s;
    /* (no lower bound on dxmin: an edge whose |dx| is below 1e-6 takes TestEdgeHyperbolaIntersection's vertical branch -- decided on the same fp32 value --
     *  whose residual is the smallest of the three; every other edge has |slope| <= dymax / 1e-6.  The bounds of rc_cell() hold for any slope bound; a large
     *  one only makes them weak.) */
    /* smallest angle: sin >= 2 A / Lmax^2, both on the pessimistic side of that perturbation (|cross(e + d, e' + d') - cross(e, e')| <= dX (|ey| + |ey'|) + dY (|ex| + |ex'|) + 2 dX dY) */
    const float c1 = ex0 * ey1, c2 = ey0 * ex1;
    const float twoA = rc_abs(c1 - c2) - 4e-7f * (rc_abs(c1) + rc_abs(c2));
    const float l0 = ex0 * ex0 + ey0 * ey0, l1 = ex1 * ex1 + ey1 * ey1, l2 = ex2 * ex2 + ey2 * ey2;
    const float L2 = rc_max(rc_max(l0, l1), l2) * 1.00001f;
    const float dl = rc_max(dX, dY);
    const float q01 = dX * (ay0 + ay1) + dY * (ax0 + ax1), q12 = dX * (ay1 + ay2) + dY * (ax1 + ax2), q20 = dX * (ay2 + ay0) + dY * (ax2 + ax0);
    const float twoAlb = twoA - rc_min(rc_min(q01, q12), q20) * 1.0001f - 2.f * dX * dY;
    const float L2ub = L2 + 3.f * dl * (dxmax + dymax) + 2.f * dl * dl;
    /* without that lower bound on the angles (thin work items; micro-triangles smaller than the rounding of their vertices) only the corner votes are open: the
     * edge tests and the centre vote do not depend on it.  Such a work item gets the weaker verdict of region_curve.h's callers: "every vote is on side s unless
     * PointInTriangle puts a cell corner of the OTHER side inside the micro-triangle", which the caller then evaluates per micro-triangle. */
    s.fat = twoAlb >= 0.005f * L2ub;
    s.Kub = (dymax / rc_max(dxmin, 1e-6f)) * 1.00001f; s.Klb = (dymin / dxmax) * 0.99999f;
    s.rhoX = rhoX; s.rhoY = rhoY;
    s.ok = 1;
    return s;
}

/* Wrap addressing of util/texture.h:34-45 (the only mode whose cells may leave [0, size)); other modes: identity, interior only */
OMMX_RC_FN int rc_wrap(int pow2, int x, int size) { return pow2 ? (int)((uint32_t)x & (uint32_t)(size - 1)) : (int)((uint32_t)x % (uint32_t)size); }

/* The cells a sub-triangle's micro-triangles can touch, as region_rect() of classify_device.h finds them (box grown by 64 ulp of maxAbs, same UV tile,
 * no seam of the address mode inside), and the sub-triangle's box in the coordinates of cell (X0, Y0).  lo / hi: the sub-triangle's fp32 AABB. */
OMMX_RC_FN RcFrame rc_frame(float lox, float loy, float hix, float hiy, float maxAbs, float fw, float fh, int w, int h, int addrMode, int pow2)
{
    RcFrame f; f.X0 = f.Y0 = f.sx = f.sy = f.nx = f.ny = 0; f.ok = 0; f.bx0 = f.bx1 = f.by0 = f.by1 = 0.f;
    if (!(maxAbs <= 16384.f)) return f;
    const float grow = maxAbs * 7.62939453125e-06f + 1e-30f;
    const float lx = lox - grow, ly = loy - grow, hx = hix + grow, hy = hiy + grow;
    if (rc_trunc(lx) != rc_trunc(hx) || rc_trunc(ly) != rc_trunc(hy)) return f;
    const int X0 = rc_trunc(rc_floor(lx * fw - 0.5f)), Y0 = rc_trunc(rc_floor(ly * fh - 0.5f));
    const int X1 = rc_trunc(rc_floor(hx * fw - 0.5f)) + 1, Y1 = rc_trunc(rc_floor(hy * fh - 0.5f)) + 1;
    const int nx = X1 - X0, ny = Y1 - Y0;
    if (nx < 1 || ny < 1 || nx > OMMX_RC_MAX_CELLS || ny > OMMX_RC_MAX_CELLS || nx >= w || ny >= h) return f;
    if (addrMode == 0) {
        f.sx = rc_wrap(pow2, X0, w); f.sy = rc_wrap(pow2, Y0, h);
        if (rc_wrap(pow2, X1, w) - f.sx != nx || rc_wrap(pow2, Y1, h) - f.sy != ny) return f;
    } else {
        if (X0 < 0 || Y0 < 0 || X1 >= w || Y1 >= h) return f;
        f.sx = X0; f.sy = Y0;
    }
    f.X0 = X0; f.Y0 = Y0; f.nx = nx; f.ny = ny;
    /* the (un-grown) box in the coordinates of cell (X0, Y0): r = size * p - (cell + 0.5), as bake_kernels_cpu.h:378-379 computes a vertex */
    const float pfx = (float)X0 + 0.5f, pfy = (float)Y0 + 0.5f;
    f.bx0 = fw * lox - pfx; f.bx1 = fw * hix - pfx; f.by0 = fh * loy - pfy; f.by1 = fh * hiy - pfy;
    f.ok = 1;
    return f;
}

/* One cell.  g00 g10 g01 g11: its texels as Load() returns them (x, y), (x+1, y), (x, y+1), (x+1, y+1); (bx0..by1): the sub-triangle's box in THIS
 * cell's coordinates.  Returns +1 / -1: every vote this cell can give a descendant is "above" / "below"; 2: no descendant touches the cell; 0: undecided. */
OMMX_RC_FN int rc_cell(const RcShape* sh, float g00, float g10, float g01, float g11, float cutoff, float bx0, float bx1, float by0, float by1)
{
    const float E = 6.0e-8f;
    const float x0 = rc_max(bx0 - sh->rhoX, -0.02f), x1 = rc_min(bx1 + sh->rhoX, 1.02f);
    const float y0 = rc_max(by0 - sh->rhoY, -0.02f), y1 = rc_min(by1 + sh->rhoY, 1.02f);
    if (x0 > x1 || y0 > y1) return 2;
    if (!(x0 <= x1 && y0 <= y1)) return 0;   /* NaN */
    const float hb = g10 - g00, hc = g01 - g00, hd = g00 + g11 - g01 - g10, ha = g00 - cutoff;   /* bake_kernels_cpu.h:337-341,377 */
    const float aa = rc_abs(ha), ab = rc_abs(hb), ac = rc_abs(hc), ad = rc_abs(hd);
    const int flat = (ab < 1e-6f) & (ac < 1e-6f) & (ad < 1e-6f);
    const float S = aa + ab + ac + ad;
    const float Kub = sh->Kub, Klb = sh->Klb;
    const float Mb = 2.5f * (1.f + Kub) * 1.000001f;
    const float C1b = (ac * Kub + ad * Mb + ab) * 1.000001f;
    const float C2b = (aa + ac * Mb) * 1.000001f;
    const float rest = 1.0001e-6f + E * (9.1f * C2b + 8.1f * C1b + 6.1f * ad * Kub + 2.02f * (ac + ad) * (Kub + 1.f)) + 66.f * E * S + 1e-30f;
    /* the quadratic branch's ill-conditioning term e C1(k)^2 / |c0(k)| of an edge of slope k, |c0| = |hd| k >= 1e-6 in that branch, C1(k) = |hc| k + |hd| 2.5 (1 + k) + |hb|:
     * convex in k above k* = 1e-6 / |hd|, so over the slopes [max(Klb, k*), Kub] the work item's edges can have it is largest at an end (curve_excluded() of
     * classify_device.h uses the cruder C1(Kub)^2 / (|hd| Klb), which needs Klb > 0: useless when an edge may be horizontal) */
    float G = 0.f;
    if (ad * Kub > 0.999999e-6f) {
        const float klo = rc_min(rc_max(Klb, 1.000001e-6f / ad), Kub);
        const float c1lo = (ac * klo + ad * 2.5f * (1.f + klo) + ab) * 1.000002f;
        G = rc_max(c1lo * c1lo / rc_max(ad * klo * 0.999999f, 0.999999e-6f), C1b * C1b / rc_max(ad * Kub * 0.999999f, 0.999999e-6f));
    }
    const float cv = 16.f * E * (rc_abs(g00) + rc_abs(g10) + rc_abs(g01) + rc_abs(g11) + rc_abs(cutoff));
    const float gy0 = ha + hc * y0, hy0 = hb + hd * y0, gy1 = ha + hc * y1, hy1 = hb + hd * y1;
    const float f00 = gy0 + x0 * hy0, f10 = gy0 + x1 * hy0, f01 = gy1 + x0 * hy1, f11 = gy1 + x1 * hy1;
    const float fmn = rc_min(rc_min(f00, f10), rc_min(f01, f11)), fmx = rc_max(rc_max(f00, f10), rc_max(f01, f11));
    const float F = rc_max(fmn, -fmx) - rest - cv;
    const int pass = (F > 1.01f * E * G) & (!flat | (F > 2e-5f));
    return pass ? (fmn > 0.f ? 1 : -1) : 0;
}

/* Round 6: the corner statement above for ONE micro-triangle and ONE cell, on its own (rc_cell() only uses it when its other bounds hold as well).
 * (bx0..by1): the box of the micro-triangle's vertices in the cell's coordinates, r = size x p - (cell + 0.5) as bake_kernels_cpu.h:378-379 computes them.
 * Returns 1 when NO corner of the unit cell can be reported inside the micro-triangle by PointInTriangle (util/geometry.h:101-114): the work item has the corner
 * bound (sh->fat: smallest angle, vertices resolved by fp32) and all four corners lie outside D = the box fattened by rho (rho = 6e-3 + 22 u + 1/64, the
 * fattening of rc_cell(): "a corner outside D is at least g = 1/64 texel away from every micro-triangle", the distance the cross-product bound needs).
 * A corner (cx, cy), cx, cy in {0, 1}, lies in the fattened box iff cx is in its x range and cy in its y range: none does when one of the two ranges holds
 * neither 0 nor 1.  NaN compares false: "not far".  The four point-in-triangle tests of a cell visit (100 of the single-texel pass's 460 vector instructions)
 * are skipped by classify_device.h: fine_single_texel() when this holds; audited by the oracle's audit build on every cell visit of the level-line kernel
 * (tests/test_edge_prefilter_audit.py: skipped visits in which one of the four tests succeeds -- none). */
OMMX_RC_FN int rc_corners_far(const RcShape* sh, float bx0, float bx1, float by0, float by1)
{
    const float rx = sh->rhoX, ry = sh->rhoY;
    const int farX = (bx0 - rx > 0.f) & (bx1 + rx < 1.f), farY = (by0 - ry > 0.f) & (by1 + ry < 1.f);
    return (sh->ok != 0) & (sh->fat != 0) & (farX | farY);
}

#endif
