// isa_budget.hip -- the per-micro-triangle pieces of classify_tiles as kernels of their own, so that their ISA can be counted by category
// (profiles/tools/isa_budget.py).  Built with the product's flags; never linked into the library.  Each kernel is ONE call of the device function per lane,
// inputs from memory, result to memory: the static instruction count of the kernel minus the fixed load / store frame is the cost per micro-triangle
// (both sides of data-dependent branches are counted: an upper bound for divergent code, exact for the straight-line pieces).
#include <hip/hip_runtime.h>
#include "../../omm_amd/csrc/bake_types.h"
#include "../../omm_amd/csrc/classify_device.h"
using namespace ommx;
typedef ModeStatic<0, 1> MD;
__device__ __forceinline__ TexWindow window_of(const float* tex, const uint32_t* sat, const void* base)
{
    TexWindow W; W.tex = (lds_float*)tex; W.sat = (lds_u32*)sat; W.base = base; W.sx = 3; W.sy = 5; W.w = 32; W.h = 32; return W;
}
#define FRAME __shared__ float s_tex[32 * 32]; __shared__ uint32_t s_sat[33 * 33]; __shared__ uint32_t s_gdec[64]; __shared__ uint8_t s_btab[256]; \
    s_tex[threadIdx.x] = uv[threadIdx.x]; s_sat[threadIdx.x] = (uint32_t)threadIdx.x; s_gdec[threadIdx.x & 63] = out[threadIdx.x]; s_btab[threadIdx.x] = (uint8_t)out[threadIdx.x + 1]; __syncthreads(); \
    const TexWindow W = window_of(s_tex, s_sat, P.mips[0].texels); const uint32_t i = threadIdx.x; float tri[6]; for (int k = 0; k < 6; ++k) tri[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(uv[k]))); \
    BirdGroup bg; bg.word = s_gdec[i >> 6]; const MicroTri t = micro_triangle_grouped(tri, bg, (uint32_t)s_btab[((bg.word >> 24) & 3u) * 64u + (i & 63u)], level);
// the frame alone: LDS set-up + split bird decode + the three interpolated vertices (what every piece below starts from)
extern "C" __global__ void k0_frame_decode_vertices(ClassifyParams P, const float* uv, uint32_t level, uint32_t* out) { FRAME out[i] = __float_as_uint(t.p0.x + t.p1.y + t.p2.x + t.lo.x + t.hi.y + t.p0p2.x + t.p1p0.y + t.p2p1.x); (void)W; }
extern "C" __global__ void k1_coarse_sat_test(ClassifyParams P, const float* uv, uint32_t level, uint32_t* out) { FRAME out[i] = (uint32_t)coarse_state_finite<MD>(P, t, W); }
extern "C" __global__ void k2_single_texel_pass(ClassifyParams P, const float* uv, uint32_t level, uint32_t* out) { FRAME out[i] = (uint32_t)fine_single_texel<false, MD>(P, t, W); }
extern "C" __global__ void k3_edge_tests(ClassifyParams P, const float* uv, uint32_t level, uint32_t* out) { FRAME out[i] = (uint32_t)single_texel_edges<false, MD>(P, t, W, (int)out[i]); }
// one curve-free-region test of a sub-triangle (one lane = one 64-group in phase 0c, one tile in triage_tiles)
extern "C" __global__ void k4_region_curve_test(ClassifyParams P, const float* uv, uint32_t level, uint32_t* out)
{
    const MicroTri sub = micro_triangle(uv, threadIdx.x, level - 3);
    float m = 0.f; for (int k = 0; k < 6; ++k) m = fmaxf(m, fabsf(uv[k]));
    out[threadIdx.x] = (uint32_t)region_curve_state<MD>(P, false, rc_shape(uv, P.mips[0].fw, P.mips[0].fh, P.mips[0].w, P.mips[0].h, level), sub, m);
}
extern "C" __global__ void k5_region_sat_query(ClassifyParams P, const float* uv, uint32_t level, uint32_t* out)
{
    const MicroTri sub = micro_triangle(uv, threadIdx.x, level - 3);
    float m = 0.f; for (int k = 0; k < 6; ++k) m = fmaxf(m, fabsf(uv[k]));
    TexWindow W; W.tex = (lds_float*)0; W.sat = (lds_u32*)0; W.base = nullptr; W.sx = W.sy = 0; W.w = W.h = 0;
    out[threadIdx.x] = (uint32_t)region_state_ex<MD>(P, sub, m, W);
}
