// host_expand.cpp -- see host_expand.h
#include "host_expand.h"
#include <emmintrin.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <mutex>
#include <thread>
#include <stdlib.h>
#include <string.h>

namespace ommx {

// The CPUs this process may use AT ONCE: hardware threads, cut by the affinity mask and by the cgroup CPU quota of the process's OWN cgroup
// (/proc/self/cgroup names it; the root of the mounted hierarchy is only right inside a cgroup namespace).  Both can change while a baker lives -- a
// scheduler re-pins the process, an orchestrator edits the quota -- so the value is read again when the cached one is older than 100 ms (a bake asks
// three times; the files cost ~20 us).
static double cgroup_quota()
{
    char rel[512] = { 0 };
    if (FILE* f = fopen("/proc/self/cgroup", "r")) {   // v2: "0::/path"; v1: "N:cpu,cpuacct:/path"
        char line[768];
        while (fgets(line, sizeof line, f)) {
            char* c1 = strchr(line, ':'); char* c2 = c1 ? strchr(c1 + 1, ':') : nullptr;
            if (!c2) continue;
            const bool v2 = line[0] == '0' && c1 == line + 1 && c2 == c1 + 1, v1cpu = strstr(c1, "cpu,") == c1 + 1 || strstr(c1, ":cpu:") == c1 || strstr(c1, ",cpu:") != nullptr;
            if (!v2 && !v1cpu) continue;
            size_t n = strlen(c2 + 1); while (n && (c2[n] == '\n' || c2[n] == '/')) c2[n--] = 0;
            snprintf(rel, sizeof rel, "%s", c2 + 1);
            if (v2) break;
        }
        fclose(f);
    }
    double best = -1.0;
    auto take = [&](double q) { if (q > 0 && (best < 0 || q < best)) best = q; };
    // the quota of the own group and of every ancestor up to the mount point applies: walk up
    char path[768];
    for (int depth = 0; depth < 32; ++depth) {
        snprintf(path, sizeof path, "/sys/fs/cgroup%s/cpu.max", rel);
        if (FILE* f = fopen(path, "r")) {
            char q[64] = { 0 }; double per = 0;
            if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) take(atof(q) / per);
            fclose(f);
        } else {
            double q = -1, per = 0;
            snprintf(path, sizeof path, "/sys/fs/cgroup/cpu%s/cpu.cfs_quota_us", rel);
            if (FILE* fq = fopen(path, "r")) { if (fscanf(fq, "%lf", &q) != 1) q = -1; fclose(fq); }
            snprintf(path, sizeof path, "/sys/fs/cgroup/cpu%s/cpu.cfs_period_us", rel);
            if (FILE* fp = fopen(path, "r")) { if (fscanf(fp, "%lf", &per) != 1) per = 0; fclose(fp); }
            if (q > 0 && per > 0) take(q / per);
        }
        char* slash = strrchr(rel, '/');
        if (!slash) break;
        *slash = 0;   // (the last round runs with rel == "": the mount point itself)
    }
    return best;
}
unsigned effective_cpus()
{
    static std::mutex mu; static unsigned cached = 0; static std::chrono::steady_clock::time_point stamp;
    std::lock_guard<std::mutex> g(mu);
    const auto now = std::chrono::steady_clock::now();
    if (cached && now - stamp < std::chrono::milliseconds(100)) return cached;
    unsigned n = std::thread::hardware_concurrency(); if (n == 0) n = 1;
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && (unsigned)a < n) n = (unsigned)a; }
    const double quota = cgroup_quota();
    if (quota > 0 && quota < (double)n) n = (unsigned)(quota + 0.5);
    cached = n ? n : 1u; stamp = now;
    return cached;
}

WorkerPool::WorkerPool(unsigned workers)
{
    try {
        threads_.reserve(workers);
        for (unsigned k = 0; k < workers; ++k) threads_.emplace_back([this] { loop(); });
    } catch (...) {}   // (std::system_error: thread limit, cgroup pids -- whatever did start is kept)
}
WorkerPool::~WorkerPool()
{
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    wake_.notify_all();
    for (auto& t : threads_) t.join();
}
void WorkerPool::loop()
{
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(uint32_t)>* fn; uint32_t tasks;
        {
            std::unique_lock<std::mutex> lk(mu_);
            wake_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_; fn = fn_; tasks = tasks_;
        }
        for (uint32_t t; (t = next_.fetch_add(1, std::memory_order_relaxed)) < tasks; ) (*fn)(t);
        { std::lock_guard<std::mutex> g(mu_); if (--active_ == 0) done_.notify_all(); }
    }
}
void WorkerPool::run(uint32_t tasks, const std::function<void(uint32_t)>& fn)
{
    if (tasks == 0) return;
    const TurnGuard one(turn_);
    if (threads_.empty() || tasks == 1) { for (uint32_t t = 0; t < tasks; ++t) fn(t); return; }
    {
        std::lock_guard<std::mutex> g(mu_);
        fn_ = &fn; tasks_ = tasks; next_.store(0, std::memory_order_relaxed); active_ = (unsigned)threads_.size(); ++generation_;
    }
    wake_.notify_all();
    for (uint32_t t; (t = next_.fetch_add(1, std::memory_order_relaxed)) < tasks; ) fn(t);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return active_ == 0; });
    fn_ = nullptr;
}

bool WorkerPool::start(uint32_t tasks, std::function<void(uint32_t)> fn)
{
    if (threads_.empty() || tasks == 0) return false;
    turn_.take();   // (kept until wait(): one run at a time)
    background_ = std::move(fn); backgroundActive_ = true;
    {
        std::lock_guard<std::mutex> g(mu_);
        fn_ = &background_; tasks_ = tasks; next_.store(0, std::memory_order_relaxed); active_ = (unsigned)threads_.size(); ++generation_;
    }
    wake_.notify_all();
    return true;
}
void WorkerPool::wait()
{
    if (!backgroundActive_) return;
    {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return active_ == 0; });
        fn_ = nullptr;
    }
    backgroundActive_ = false; background_ = nullptr;
    turn_.give();
}

int WorkerPool::bind_near(const void* memory)
{
    const TurnGuard one(turn_);
    // the node of the page (move_pages with a null target list only reports): raw syscall, no libnuma in the image
    void* page = (void*)((uintptr_t)memory & ~(uintptr_t)4095); int node = -1;
#ifdef SYS_move_pages
    if (syscall(SYS_move_pages, 0, 1ul, &page, (const int*)nullptr, &node, 0) != 0) node = -1;
#endif
    cpu_set_t allowedNow; CPU_ZERO(&allowedNow);
    const bool haveMask = sched_getaffinity(0, sizeof allowedNow, &allowedNow) == 0;
    // (bound already -- unless the process's own mask has changed since: the workers may sit on CPUs it must no longer use)
    if (node < 0 || threads_.empty() || (node == boundNode_ && haveMask && CPU_EQUAL(&allowedNow, &boundMask_))) return node;
    char path[128]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    char list[4096] = { 0 };
    const bool got = fgets(list, sizeof list, f) != nullptr; fclose(f);
    if (!got) return -1;
    if (!haveMask) return -1;
    const cpu_set_t allowed = allowedNow;
    // the node's cores this process may use, one entry per physical core (the first hardware thread of its sibling list), in CPU order
    std::vector<int> cores;
    for (char* p = list; *p; ) {   // "64-127,192-255"
        char* end = nullptr; const long a = strtol(p, &end, 10); if (end == p) break;
        long b = a; p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            if (!CPU_ISSET((int)c, &allowed)) continue;
            long first = c;
            snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%ld/topology/thread_siblings_list", c);
            if (FILE* s = fopen(path, "r")) { if (fscanf(s, "%ld", &first) != 1) first = c; fclose(s); }
            if (first == c) cores.push_back((int)c);
        }
        if (*p == ',') ++p; else break;
    }
    if (cores.empty()) return -1;
    // Worker k gets its own window of the node's cores (k-th of W equal slices): the threads end up on different core complexes.  Bound to the node as a whole,
    // the scheduler packed them onto the first few cores and their sibling threads -- one complex's share of the memory bandwidth: 9 - 10 ms per 1.27 GB instead
    // of 3.6 (profiles/tools/host_numa_probe.cpp: 346 GB/s spread over the block's node, 130 GB/s packed, 40 - 95 GB/s from the other socket).
    const size_t W = threads_.size(), n = cores.size();
    for (size_t k = 0; k < W; ++k) {
        cpu_set_t want; CPU_ZERO(&want);
        size_t lo = k * n / W, hi = (k + 1) * n / W; if (hi <= lo) { lo = k % n; hi = lo + 1; }
        for (size_t c = lo; c < hi; ++c) CPU_SET(cores[c], &want);
        (void)pthread_setaffinity_np(threads_[k].native_handle(), sizeof want, &want);
    }
    boundNode_ = node; boundMask_ = allowed;
    return node;
}

HostCodecLayout host_codec_layout(uint64_t paddedBytes)
{
    HostCodecLayout c; c.units = paddedBytes / 16u; c.blocks = (c.units + 255u) / 256u; c.offOfs = 16u;
    c.offCodes = (c.offOfs + 4u * (c.blocks + 1u) + 15u) & ~15ull; c.offRaw = (c.offCodes + (c.units + 1u) / 2u + 15u) & ~15ull;
    return c;
}

namespace {
const uint32_t kPattern[4] = { 0u, 0x55555555u, 0xAAAAAAAAu, 0xFFFFFFFFu };
// a 4 KiB block of one repeated state (96 % of the blocks of a bake): non-temporal stores as wide as the CPU has them (the destination is 16-byte aligned;
// 4 KiB blocks of an aligned array are then 32- / 64-byte aligned iff the array is, which is checked per call)
__attribute__((target("avx2"))) void fill4k_avx2(uint8_t* d, uint32_t pattern)
{
    const __m256i v = _mm256_set1_epi32((int)pattern);
    for (int k = 0; k < 128; ++k) _mm256_stream_si256((__m256i*)(d + 32 * k), v);
}
void fill4k_sse2(uint8_t* d, uint32_t pattern)
{
    const __m128i v = _mm_set1_epi32((int)pattern);
    for (int k = 0; k < 256; ++k) _mm_stream_si128((__m128i*)(d + 16 * k), v);
}
typedef void (*Fill4k)(uint8_t*, uint32_t);
const Fill4k g_fill4k = __builtin_cpu_supports("avx2") ? fill4k_avx2 : fill4k_sse2;
template <bool NT> inline void put16(uint8_t* d, __m128i v) { if (NT) _mm_stream_si128((__m128i*)d, v); else _mm_storeu_si128((__m128i*)d, v); }
template <bool NT>
void expand(uint8_t* dst, uint64_t dstBytes, const uint8_t* stream, const HostCodecLayout& L, uint64_t b0, uint64_t b1, const ZeroedPieces* zeroed, uint64_t* skipped)
{
    uint64_t skippedHere = 0;
    const uint32_t* ofs = (const uint32_t*)(stream + L.offOfs);
    for (uint64_t b = b0; b < b1; ++b) {
        const uint64_t u0 = b * 256u, u1 = u0 + 256u < L.units ? u0 + 256u : L.units;
        const uint8_t* codes = stream + L.offCodes + u0 / 2u;
        const uint8_t* raw = stream + L.offRaw + 16ull * ofs[b];
        uint8_t* d = dst + u0 * 16u;
        const bool whole = u1 * 16u <= dstBytes && u1 - u0 == 256u;
        if (whole) {
            // a block of one repeated state (most of them): 128 equal code bytes 0x00 / 0x11 / 0x22 / 0x33
            uint64_t w0; memcpy(&w0, codes, 8);
            bool same = (w0 & 0xCCCCCCCCCCCCCCCCull) == 0 && ((w0 >> 4) & 0x0F0F0F0F0F0F0F0Full) == (w0 & 0x0F0F0F0F0F0F0F0Full) && w0 == (w0 & 0xFF) * 0x0101010101010101ull;
            for (int k = 1; same && k < 16; ++k) { uint64_t w; memcpy(&w, codes + 8 * k, 8); same = w == w0; }
            if (same) {
                // (a block of zeros in a piece of the array that was zeroed while the device was baking: nothing to write)
                if (zeroed && (w0 & 3u) == 0u) {
                    const size_t piece = (size_t)((u0 * 16u) >> 21);
                    if (piece < zeroed->pieces && zeroed->done[piece].load(std::memory_order_acquire)) { skippedHere += 4096u; continue; }
                }
                if (NT && ((uintptr_t)d & 31u) == 0u) { g_fill4k(d, kPattern[w0 & 3u]); continue; }
                const __m128i v = _mm_set1_epi32((int)kPattern[w0 & 3u]);
                for (int k = 0; k < 256; ++k) put16<NT>(d + 16 * k, v);
                continue;
            }
            // (in a zeroed piece, a 64-byte line of four zero units is left alone too: states come in long runs, so most lines of a mixed block are one state)
            const bool zeroedPiece = zeroed && (size_t)((u0 * 16u) >> 21) < zeroed->pieces && zeroed->done[(size_t)((u0 * 16u) >> 21)].load(std::memory_order_acquire)
                                     && ((uintptr_t)d & 63u) == 0u;
            for (int k = 0; k < 128; ++k) {
                if (zeroedPiece && (k & 1) == 0) { uint16_t four; memcpy(&four, codes + k, 2); if (four == 0u) { skippedHere += 64u; ++k; continue; } }
                const uint32_t two = codes[k], c0 = two & 15u, c1 = two >> 4;
                __m128i v0, v1;
                if (c0 < 4u) v0 = _mm_set1_epi32((int)kPattern[c0]); else { v0 = _mm_loadu_si128((const __m128i*)raw); raw += 16; }
                if (c1 < 4u) v1 = _mm_set1_epi32((int)kPattern[c1]); else { v1 = _mm_loadu_si128((const __m128i*)raw); raw += 16; }
                put16<NT>(d + 32 * k, v0); put16<NT>(d + 32 * k + 16, v1);
            }
            continue;
        }
        // the last block of the array: units that end beyond dstBytes are cut
        for (uint64_t u = u0; u < u1; ++u) {
            const uint32_t c = (codes[(u - u0) / 2u] >> (4u * (uint32_t)((u - u0) & 1u))) & 15u;
            uint8_t tmp[16];
            if (c < 4u) { for (int k = 0; k < 4; ++k) memcpy(tmp + 4 * k, &kPattern[c], 4); } else { memcpy(tmp, raw, 16); raw += 16; }
            const uint64_t at = u * 16u;
            if (at >= dstBytes) break;
            memcpy(dst + at, tmp, at + 16u <= dstBytes ? 16u : (size_t)(dstBytes - at));
        }
    }
    if (NT) _mm_sfence();
    if (skipped) *skipped += skippedHere;
}
} // namespace

void codec_expand_blocks(uint8_t* dst, uint64_t dstBytes, const uint8_t* stream, const HostCodecLayout& L, uint64_t b0, uint64_t b1, const ZeroedPieces* zeroed, uint64_t* skipped)
{
    if (((uintptr_t)dst & 15u) == 0) expand<true>(dst, dstBytes, stream, L, b0, b1, zeroed, skipped); else expand<false>(dst, dstBytes, stream, L, b0, b1, zeroed, skipped);
}

void fill_zero_nt(uint8_t* dst, size_t lo, size_t hi)
{
    size_t o = lo;
    if (((uintptr_t)dst & 31u) == 0u) for (; o + 4096u <= hi; o += 4096u) g_fill4k(dst + o, 0u);
    if (o < hi) memset(dst + o, 0, hi - o);
    _mm_sfence();
}

void codec_scatter_omms(const HostScatter& S, uint32_t j0, uint32_t j1)
{
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t item = S.order[j], n = S.sizes[j];
        uint8_t* dst = S.arrayData + S.dstOfs[j];
        if (!S.active[item]) {   // uniform item: constant pattern (as the device scatter computes it)
            uint32_t st = 0; for (uint32_t m = S.stateMask[item]; m > 1u; m >>= 1) ++st;
            uint32_t usedBits = (1u << (2u * S.level[item])) * (uint32_t)S.bits; if (usedBits > 8u) usedBits = 8u;
            uint32_t pat = 0;
            for (uint32_t b = 0; b < usedBits; b += (uint32_t)S.bits) pat |= st << b;
            memset(dst, (int)(pat & 0xFFu), n);
            continue;
        }
        const uint32_t r = S.owner[item];
        if (r >= S.world) continue;   // (cannot happen: every active item has an owner; checked by the caller's layout)
        const uint64_t c0 = S.cofs[j], c1 = c0 + n;
        if (S.raw[r]) { memcpy(dst, S.stream[r] + c0, n); continue; }
        if (((c0 | (uint64_t)n) & 4095u) == 0u) {   // whole codec blocks (every block of level >= 7 in 4-state): straight into the result
            codec_expand_blocks((uint8_t*)((uintptr_t)dst - (uintptr_t)c0), c1, S.stream[r], S.L, c0 / 4096u, c1 / 4096u);
            continue;
        }
        for (uint64_t B = c0 / 4096u; B * 4096u < c1; ++B) {   // small or unaligned blocks: one codec block at a time through a scratch page
            alignas(64) uint8_t tmp[4096];
            codec_expand_blocks((uint8_t*)((uintptr_t)tmp - (uintptr_t)(B * 4096u)), (B + 1u) * 4096u, S.stream[r], S.L, B, B + 1u);
            const uint64_t lo = B * 4096u > c0 ? B * 4096u : c0, hi = (B + 1u) * 4096u < c1 ? (B + 1u) * 4096u : c1;
            memcpy(dst + (lo - c0), tmp + (lo - B * 4096u), (size_t)(hi - lo));
        }
    }
}

} // namespace ommx
