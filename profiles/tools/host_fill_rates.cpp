// host_fill_rates.cpp -- how fast can host threads EXPAND a codec stream (tail_kernels.hip: one nibble per 16-byte unit = which state it repeats, or raw)
// into the caller's arrayData?  Write-only traffic on the destination; the question behind "send the ommCpuBake result over PCIe as codec streams".
// (profiles/tools; not part of the product)   build: g++ -O2 -pthread -o host_fill_rates host_fill_rates.cpp ; run on the GPU box
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <emmintrin.h>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const uint32_t kPattern[4] = { 0u, 0x55555555u, 0xAAAAAAAAu, 0xFFFFFFFFu };
// decode codec blocks [b0, b1): 256 units of 16 bytes each; ofs[b] = first raw unit of block b
template <bool NT>
static void decode(uint8_t* dst, const uint8_t* codes, const uint32_t* ofs, const uint8_t* raw, size_t b0, size_t b1)
{
    for (size_t b = b0; b < b1; ++b) {
        const uint8_t* c = codes + b * 128; const __m128i* r = (const __m128i*)(raw + 16ull * ofs[b]); __m128i* d = (__m128i*)(dst + b * 4096);
        for (int k = 0; k < 128; ++k) {
            const uint32_t two = c[k];
            for (int h = 0; h < 2; ++h) {
                const uint32_t code = (two >> (4 * h)) & 15u;
                const __m128i v = code < 4u ? _mm_set1_epi32((int)kPattern[code]) : _mm_loadu_si128(r++);
                if (NT) _mm_stream_si128(d++, v); else _mm_storeu_si128(d++, v);
            }
        }
    }
    if (NT) _mm_sfence();
}
template <bool NT>
static double run(const char* what, uint8_t* dst, const uint8_t* codes, const uint32_t* ofs, const uint8_t* raw, size_t blocks, unsigned nt)
{
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        const double t0 = now();
        std::vector<std::thread> th;
        for (unsigned p = 0; p < nt; ++p) th.emplace_back([=] { decode<NT>(dst, codes, ofs, raw, blocks * p / nt, blocks * (p + 1) / nt); });
        for (auto& t : th) t.join();
        const double dt = now() - t0; if (dt < best) best = dt;
    }
    printf("%-44s %3u threads  %.2f ms  %.1f GB/s written\n", what, nt, best, blocks * 4096.0 / best / 1e6);
    return best;
}
static void fill(const char* what, uint8_t* dst, size_t bytes, unsigned nt)
{
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        const double t0 = now();
        std::vector<std::thread> th;
        for (unsigned p = 0; p < nt; ++p) th.emplace_back([=] { const size_t lo = bytes * p / nt / 4096 * 4096, hi = bytes * (p + 1) / nt / 4096 * 4096; memset(dst + lo, 0x55, hi - lo); });
        for (auto& t : th) t.join();
        const double dt = now() - t0; if (dt < best) best = dt;
    }
    printf("%-44s %3u threads  %.2f ms  %.1f GB/s written\n", what, nt, best, bytes / best / 1e6);
}
int main(int argc, char** argv)
{
    const size_t bytes = (size_t)1216 << 20, blocks = bytes / 4096, units = bytes / 16;   // 1.27 GB like the metric configuration's arrayData
    const double rawFrac = argc > 1 ? atof(argv[1]) : 0.033;
    uint8_t* dst = (uint8_t*)aligned_alloc(2 << 20, bytes); madvise(dst, bytes, MADV_HUGEPAGE); memset(dst, 1, bytes);
    std::vector<uint8_t> codes(units / 2); std::vector<uint32_t> ofs(blocks + 1); std::vector<uint8_t> raw;
    uint64_t s = 88172645463325252ull; auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    uint32_t nraw = 0; uint32_t run_state = 0; int run_left = 0;
    for (size_t u = 0; u < units; ++u) {
        if (u % 256 == 0) ofs[u / 256] = nraw;
        if (run_left == 0) { run_state = (uint32_t)(rnd() % 4); run_left = 1 + (int)(rnd() % 600); }
        --run_left;
        const bool isRaw = (rnd() % 100000) < (uint64_t)(rawFrac * 100000);
        const uint32_t code = isRaw ? 4u : run_state;
        codes[u / 2] = (uint8_t)((u & 1) ? (codes[u / 2] | (code << 4)) : code);
        if (isRaw) { ++nraw; for (int k = 0; k < 16; ++k) raw.push_back((uint8_t)rnd()); }
    }
    ofs[blocks] = nraw; raw.resize(raw.size() + 64);
    printf("hardware_concurrency %u; result %.2f GB, stream %.1f MB (%.1f %% raw units)\n", std::thread::hardware_concurrency(), bytes / 1e9, (codes.size() + raw.size() + ofs.size() * 4) / 1e6, 100.0 * nraw / units);
    FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"); if (f) { char b[64] = { 0 }; if (fgets(b, 63, f)) printf("cgroup cpu.max: %s", b); fclose(f); }
    for (unsigned nt : { 1u, 2u, 4u, 8u, 12u, 16u, 24u, 32u }) {
        fill("memset (glibc: non-temporal when large)", dst, bytes, nt);
        run<false>("decode, plain 16-byte stores", dst, codes.data(), ofs.data(), raw.data(), blocks, nt);
        run<true>("decode, non-temporal 16-byte stores", dst, codes.data(), ofs.data(), raw.data(), blocks, nt);
    }
    // a fresh destination every time (first touch: page faults inside the timed region), 16 threads
    for (int k = 0; k < 2; ++k) {
        uint8_t* fresh = (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (k) madvise(fresh, bytes, MADV_HUGEPAGE);
        const double t0 = now();
        std::vector<std::thread> th;
        for (unsigned p = 0; p < 16; ++p) th.emplace_back([=] { decode<true>(fresh, codes.data(), ofs.data(), raw.data(), blocks * p / 16, blocks * (p + 1) / 16); });
        for (auto& t : th) t.join();
        printf("decode into FRESH pages (%s), 16 threads: %.2f ms\n", k ? "MADV_HUGEPAGE" : "4K pages", now() - t0);
        munmap(fresh, bytes);
    }
    return 0;
}
