// valu_rates.hip -- issue cost of the instruction classes classify_tiles is made of, measured on the machine the bench runs on.
//
// The bake's dominant kernel is bound by VALU issue, not by HBM (DESIGN.md "Roofline statement").  To turn the SQ instruction
// counters of a profile into "where the cycles go" one needs the cost of one wave64 instruction of every class on a gfx950 SIMD.
// This tool measures it: every kernel runs ITER x 8 copies of ONE instruction (inline asm, 8 independent register chains) on
// every SIMD of the chip, once with 4 and once with 8 waves per SIMD; the time difference is four more waves' worth of instructions per
// SIMD (launch overhead and single-wave issue latency cancel), reported as SIMD cycles per wave64 instruction at the 2.4 GHz peak clock.
//
//   hipcc --offload-arch=gfx950 -O2 -o profiles/bin/valu_rates profiles/valu_rates.hip     (cross-compiles without a GPU)
//   profiles/bin/valu_rates > gpurun_out/valu_rates.json                                   (on the GPU box; committed as profiles/valu_rates_mi355x.json)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 16384;

#define REP8(S0, S1, S2, S3, S4, S5, S6, S7) S0 "\n" S1 "\n" S2 "\n" S3 "\n" S4 "\n" S5 "\n" S6 "\n" S7 "\n"

// f32 two-source op:  r_k = op(r_k, c)
#define KERNEL_F32(NAME, OP)                                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, long long* cyc)                                                     \
    {                                                                                                                            \
        float r0 = threadIdx.x * 1e-3f + 1.f, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f; \
        const float c = 1.0000001f;                                                                                              \
        const long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITER; ++i)                                                                                           \
            asm volatile(REP8(OP " %0, %0, %8", OP " %1, %1, %8", OP " %2, %2, %8", OP " %3, %3, %8", OP " %4, %4, %8",          \
                              OP " %5, %5, %8", OP " %6, %6, %8", OP " %7, %7, %8")                                              \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));             \
        const long long t1 = __builtin_readcyclecounter();                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                       \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                                \
    }
// f32 one-source op:  r_k = op(r_k)
#define KERNEL_F32_1(NAME, OP)                                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float* out, long long* cyc)                                                     \
    {                                                                                                                            \
        float r0 = threadIdx.x * 1e-3f + 1.f, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f; \
        const long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITER; ++i)                                                                                           \
            asm volatile(REP8(OP " %0, %0", OP " %1, %1", OP " %2, %2", OP " %3, %3", OP " %4, %4", OP " %5, %5", OP " %6, %6", OP " %7, %7") \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));                      \
        const long long t1 = __builtin_readcyclecounter();                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                       \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                                \
    }
// f32 three-source op:  r_k = op(r_k, c, r_k)
#define KERNEL_F32_3(NAME, OP)                                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float* out, long long* cyc)                                                     \
    {                                                                                                                            \
        float r0 = threadIdx.x * 1e-3f + 1.f, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f; \
        const float c = 0.5f;                                                                                                    \
        const long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITER; ++i)                                                                                           \
            asm volatile(REP8(OP " %0, %0, %8, %0", OP " %1, %1, %8, %1", OP " %2, %2, %8, %2", OP " %3, %3, %8, %3",            \
                              OP " %4, %4, %8, %4", OP " %5, %5, %8, %5", OP " %6, %6, %8, %6", OP " %7, %7, %8, %7")            \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));             \
        const long long t1 = __builtin_readcyclecounter();                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                       \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                                \
    }
// 64-bit register pairs (f64 and packed f32): r_k = op(r_k, c)
#define KERNEL_64(NAME, TYPE, OP, INIT, CVAL)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(float* out, long long* cyc)                                                     \
    {                                                                                                                            \
        TYPE r0 = INIT(1), r1 = INIT(2), r2 = INIT(3), r3 = INIT(4), r4 = INIT(5), r5 = INIT(6), r6 = INIT(7), r7 = INIT(8);     \
        const TYPE c = CVAL;                                                                                                     \
        const long long t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < ITER; ++i)                                                                                           \
            asm volatile(REP8(OP " %0, %0, %8", OP " %1, %1, %8", OP " %2, %2, %8", OP " %3, %3, %8", OP " %4, %4, %8",          \
                              OP " %5, %5, %8", OP " %6, %6, %8", OP " %7, %7, %8")                                              \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));             \
        const long long t1 = __builtin_readcyclecounter();                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8;                                                                       \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                                \
    }
// compare + select pair (the shape of every `cond ? a : b` on floats): counted as TWO instructions
__global__ __launch_bounds__(256) void k_cmp_cndmask(float* out, long long* cyc)
{
    float r0 = threadIdx.x * 1e-3f + 1.f, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f;
    const float c = 2.5f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i)
        asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                     "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c) : "vcc");
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// scalar ALU and an (always taken-through) exec-mask branch skeleton: s_and_saveexec + s_cbranch_execz + s_or restore = 3 "instructions"
__global__ __launch_bounds__(256) void k_salu(float* out, long long* cyc)
{
    uint32_t s0 = blockIdx.x, s1 = 3;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP8("s_add_u32 %0, %0, %1", "s_xor_b32 %0, %0, %1", "s_add_u32 %0, %0, %1", "s_xor_b32 %0, %0, %1",
                          "s_add_u32 %0, %0, %1", "s_xor_b32 %0, %0, %1", "s_add_u32 %0, %0, %1", "s_xor_b32 %0, %0, %1") : "+s"(s0) : "s"(s1) : "scc");
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s0;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_lds_read(float* out, long long* cyc)
{
    __shared__ float buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = (float)i;
    __syncthreads();
    float r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    const uint32_t a = (threadIdx.x * 4u) & 4095u;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP8("ds_read_b32 %0, %8", "ds_read_b32 %1, %8 offset:4", "ds_read_b32 %2, %8 offset:8", "ds_read_b32 %3, %8 offset:12",
                          "ds_read_b32 %4, %8 offset:16", "ds_read_b32 %5, %8 offset:20", "ds_read_b32 %6, %8 offset:24", "ds_read_b32 %7, %8 offset:28")
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a) : "memory");
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

KERNEL_F32(k_mul_f32, "v_mul_f32")
KERNEL_F32(k_add_f32, "v_add_f32")
KERNEL_F32(k_max_f32, "v_max_f32")
KERNEL_F32(k_and_b32, "v_and_b32")
KERNEL_F32(k_add_u32, "v_add_u32")
KERNEL_F32(k_mul_lo_u32, "v_mul_lo_u32")
KERNEL_F32(k_lshlrev_b32, "v_lshlrev_b32")
KERNEL_F32_3(k_fma_f32, "v_fma_f32")
KERNEL_F32_3(k_div_fixup_f32, "v_div_fixup_f32")
KERNEL_F32_1(k_sqrt_f32, "v_sqrt_f32")
KERNEL_F32_1(k_rcp_f32, "v_rcp_f32")
KERNEL_F32_1(k_rsq_f32, "v_rsq_f32")
KERNEL_F32_1(k_floor_f32, "v_floor_f32")
KERNEL_F32_1(k_cvt_i32_f32, "v_cvt_i32_f32")
KERNEL_F32_1(k_mov_b32, "v_mov_b32")
#define INITD(k) ((double)threadIdx.x * 1e-3 + (double)(k))
#define INITP(k) (v2f){ threadIdx.x * 1e-3f + (float)(k), (float)(k) }
#define SUM8 (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7)
KERNEL_64(k_mul_f64, double, "v_mul_f64", INITD, 1.0000001)
KERNEL_64(k_add_f64, double, "v_add_f64", INITD, 1.0000001)
#undef SUM8
#define SUM8 (r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x + r0.y + r7.y)
KERNEL_64(k_pk_mul_f32, v2f, "v_pk_mul_f32", INITP, ((v2f){ 1.0000001f, 0.9999999f }))
KERNEL_64(k_pk_add_f32, v2f, "v_pk_add_f32", INITP, ((v2f){ 1.0000001f, 0.9999999f }))
#undef SUM8

// the two IEEE sequences the compiler expands `a / b` and sqrtf(x) into under -fno-fast-math (what the bake really executes)
__global__ __launch_bounds__(256) void k_ieee_div(float* out, long long* cyc)
{
    float r0 = threadIdx.x * 1e-3f + 1.f, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f;
    const float c = out[0] + 1.0000001f; // run-time divisor
    const long long t0 = __builtin_readcyclecounter();
    #pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
        r0 = r0 / c; r1 = r1 / c; r2 = r2 / c; r3 = r3 / c;
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_ieee_sqrt(float* out, long long* cyc)
{
    float r0 = threadIdx.x * 1e-3f + 1.f, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f;
    const long long t0 = __builtin_readcyclecounter();
    #pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
        r0 = __builtin_sqrtf(r0) + 1.f; r1 = __builtin_sqrtf(r1) + 1.f; r2 = __builtin_sqrtf(r2) + 1.f; r3 = __builtin_sqrtf(r3) + 1.f;
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

struct Case { const char* name; void (*fn)(float*, long long*); int instrPerIter; const char* note; };

int main()
{
    const Case cases[] = {
        { "v_mul_f32", k_mul_f32, 8, "" }, { "v_add_f32", k_add_f32, 8, "" }, { "v_fma_f32", k_fma_f32, 8, "" }, { "v_max_f32", k_max_f32, 8, "" },
        { "v_pk_mul_f32", k_pk_mul_f32, 8, "2 fp32 results per lane" }, { "v_pk_add_f32", k_pk_add_f32, 8, "2 fp32 results per lane" },
        { "v_mov_b32", k_mov_b32, 8, "" }, { "v_and_b32", k_and_b32, 8, "" }, { "v_add_u32", k_add_u32, 8, "" }, { "v_lshlrev_b32", k_lshlrev_b32, 8, "" },
        { "v_mul_lo_u32", k_mul_lo_u32, 8, "" }, { "v_floor_f32", k_floor_f32, 8, "" }, { "v_cvt_i32_f32", k_cvt_i32_f32, 8, "" },
        { "v_sqrt_f32", k_sqrt_f32, 8, "transcendental" }, { "v_rcp_f32", k_rcp_f32, 8, "transcendental" }, { "v_rsq_f32", k_rsq_f32, 8, "transcendental" },
        { "v_div_fixup_f32", k_div_fixup_f32, 8, "" },
        { "v_mul_f64", k_mul_f64, 8, "" }, { "v_add_f64", k_add_f64, 8, "" },
        { "v_cmp_lt_f32+v_cndmask_b32", k_cmp_cndmask, 8, "4 pairs" }, { "s_add_u32/s_xor_b32", k_salu, 8, "scalar unit, dependent chain" },
        { "ds_read_b32", k_lds_read, 8, "conflict-free, waited per 8" },
        { "IEEE a/b (compiler expansion)", k_ieee_div, 4, "cycles per DIVISION" }, { "IEEE sqrtf (compiler expansion) + add", k_ieee_sqrt, 4, "cycles per SQRT (+1 add)" },
    };
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"iter\": %d, \"unit\": \"SIMD cycles per wave64 instruction at 2.4 GHz = (t[8 waves/SIMD] - t[4 waves/SIMD]) x 2.4e9 / (4 x instructions per wave)\", \"cases\": [\n", prop.gcnArchName, cus, ITER);
    bool first = true;
    for (const Case& c : cases) {
        // kernel time at 4 and at 8 waves per SIMD: the difference is 4 more waves' worth of instructions on every SIMD, free of launch
        // overhead and of the single-wave issue latency (a lone wave issues one VALU instruction per ~4.8 cycles, two already saturate it)
        double ms[2] = { 0, 0 };
        for (int v = 0; v < 2; ++v) {
            const int wavesPerSimd = v == 0 ? 4 : 8;
            const int blocks = cus * wavesPerSimd;          // 256 threads = 4 waves = one per SIMD of a CU
            const size_t threads = (size_t)blocks * 256;
            float* out = nullptr; long long* cyc = nullptr;
            CHECK(hipMalloc((void**)&out, threads * sizeof(float))); CHECK(hipMalloc((void**)&cyc, threads / 64 * sizeof(long long)));
            CHECK(hipMemset(out, 0, threads * sizeof(float)));
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, cyc);   // warm-up
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, cyc);
                CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
                float t = 0; CHECK(hipEventElapsedTime(&t, e0, e1)); best = t < best ? t : best;
            }
            ms[v] = best;
            CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1)); CHECK(hipFree(out)); CHECK(hipFree(cyc));
        }
        const double perInstr = (ms[1] - ms[0]) * 1e-3 * 2.4e9 / ((double)ITER * c.instrPerIter * 4.0);
        printf("%s  {\"op\": \"%s\", \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.2f, \"ms_4_waves\": %.4f, \"ms_8_waves\": %.4f, \"note\": \"%s\"}",
               first ? "" : ",\n", c.name, perInstr, ms[0], ms[1], c.note);
        first = false;
    }
    printf("\n]}\n");
    return 0;
}
