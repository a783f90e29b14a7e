// host_copy_rates.cpp -- how fast can host threads move OMM blocks out of a pinned staging buffer?  (profiles/tools; not part of the product)
// build: hipcc -O2 -o host_copy_rates host_copy_rates.cpp -lpthread ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void run(const char* what, uint8_t* dst, const uint8_t* src, size_t bytes, unsigned nt, size_t block)
{
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        const double t0 = now();
        std::vector<std::thread> th;
        for (unsigned p = 0; p < nt; ++p) th.emplace_back([=] { const size_t lo = bytes * p / nt / block * block, hi = bytes * (p + 1) / nt / block * block; for (size_t o = lo; o < hi; o += block) memcpy(dst + o, src + o, block); });
        for (auto& t : th) t.join();
        const double dt = now() - t0; if (dt < best) best = dt;
    }
    printf("%-34s %3u threads  block %7zu  %.2f ms  %.1f GB/s\n", what, nt, block, best, bytes / best / 1e6);
}
int main()
{
    const size_t bytes = (size_t)1280 << 20;
    uint8_t *pinned = nullptr, *pinned2 = nullptr;
    if (hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    if (hipHostMalloc((void**)&pinned2, bytes, hipHostMallocNonCoherent) != hipSuccess) { printf("hipHostMalloc(noncoherent) failed\n"); pinned2 = nullptr; }
    uint8_t* a = (uint8_t*)aligned_alloc(2 << 20, bytes); uint8_t* b = (uint8_t*)aligned_alloc(2 << 20, bytes);
    madvise(a, bytes, MADV_HUGEPAGE); madvise(b, bytes, MADV_HUGEPAGE);
    memset(a, 1, bytes); memset(b, 2, bytes); memset(pinned, 3, bytes); if (pinned2) memset(pinned2, 4, bytes);
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    for (unsigned nt : { 1u, 4u, 8u, 16u, 32u, 64u, 128u }) {
        run("pageable -> pageable", a, b, bytes, nt, 16384);
        run("pinned(default) -> pageable", a, pinned, bytes, nt, 16384);
        if (pinned2) run("pinned(noncoherent) -> pageable", a, pinned2, bytes, nt, 16384);
    }
    // device -> pinned and device -> pageable for reference
    uint8_t* d = nullptr; hipMalloc((void**)&d, bytes); hipMemset(d, 5, bytes); hipDeviceSynchronize();
    for (int k = 0; k < 2; ++k) {
        double t0 = now(); hipMemcpy(pinned, d, bytes, hipMemcpyDeviceToHost); printf("D2H -> pinned   %.2f ms\n", now() - t0);
        t0 = now(); hipMemcpy(a, d, bytes, hipMemcpyDeviceToHost); printf("D2H -> pageable %.2f ms\n", now() - t0);
    }
    return 0;
}
