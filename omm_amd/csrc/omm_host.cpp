// omm_host.cpp -- the C ABI (include/omm_mi355x.h) and the host orchestration of a bake.
//
// Host code is C++ like the reference's (libraries/omm-lib/src/bake.cpp, bake_cpu_impl.cpp); it only
// validates, builds the work-item list and drives the HIP kernels.  There is NO CPU classification
// path here: without a working HIP device every bake returns ommResult_FAILURE with a Fatal log line.
#include "../../include/omm_mi355x.h"
#include "../../include/omm_mi355x_ext.h"
#include "bake_types.h"
#include "bake_kernels.h"
#include "host_tail.h"
#include "host_expand.h"

#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <malloc.h>
#include <math.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <memory>
#include <new>
#include <exception>
#include <thread>
#include <sys/mman.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>
#include <functional>
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
    if (n && mem) { const size_t old = malloc_usable_size(mem); memcpy(n, mem, old < size ? old : size); free(mem); } // never reads past the old block
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

// ---- nothing may unwind through the C ABI (the SDK "never throws"): std::bad_alloc / length_error from the
//      host-side containers become ommResult_FAILURE with a log line ----
template <class F> ommResult guarded(const Logger* log, F&& body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc&) { if (log) log->msg(ommMessageSeverity_Fatal, "[Failure] - out of host memory"); }
    catch (const std::exception& e) { if (log) { char buf[256]; snprintf(buf, sizeof buf, "[Failure] - %s", e.what()); log->msg(ommMessageSeverity_Fatal, buf); } }
    catch (...) { if (log) log->msg(ommMessageSeverity_Fatal, "[Failure] - unexpected exception"); }
    return ommResult_FAILURE;
}

// ---- device arena: one grow-only HBM block per baker, reused across bakes ----
constexpr size_t kHostBlockBytes = 64u << 10;
// 64 KiB blocks of pinned host memory, the destination of a bake's small read-backs (counters, tail summary, histograms), which otherwise go through the
// runtime's staging buffer -- a wait and a host copy per read-back.  Process-wide free list: pinning costs a system call and a page-table update, and
// applications (and the test suite) create and destroy bakers by the hundred; blocks are never unpinned (a handful per concurrent bake).
struct HostBlockPool {
    std::mutex mu; std::vector<uint8_t*> idle;
    uint8_t* acquire() {
        { std::lock_guard<std::mutex> g(mu); if (!idle.empty()) { uint8_t* p = idle.back(); idle.pop_back(); return p; } }
        uint8_t* p = nullptr;
        if (hipHostMalloc((void**)&p, kHostBlockBytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void release(uint8_t* p) { if (p) { std::lock_guard<std::mutex> g(mu); idle.push_back(p); } }
    static HostBlockPool& get() { static HostBlockPool* pool = new HostBlockPool(); return *pool; }   // (never destroyed: no HIP calls at process exit)
};
struct DeviceArena {
    uint8_t* base = nullptr; size_t cap = 0, used = 0;
    uint8_t* hostBlock = nullptr; bool hostBlockTried = false;
    ~DeviceArena() { if (base) (void)hipFree(base); HostBlockPool::get().release(hostBlock); }
    // (null when pinning fails: the caller reads into pageable memory)
    uint8_t* host_block() {
        if (!hostBlockTried) { hostBlockTried = true; hostBlock = HostBlockPool::get().acquire(); }
        return hostBlock;
    }
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

// The device working set of one bake in flight: per-item tables + scratch, packed states + tile queue, and (sharded bakes) the exchange
// buffers.  A baker keeps a small pool of these sets: a bake takes one for its duration -- concurrent bakes on one baker get different
// sets, a sharded bake keeps its set from Begin to Destroy without holding any lock in between -- and steady-state bakes never hipMalloc.
// pinned host memory of a working set (grow-only like the device arenas): staging of the blocks that ommCpuBake streams out during classification
struct HostArena {
    uint8_t* base = nullptr; size_t cap = 0;
    ~HostArena() { if (base) (void)hipHostFree(base); }
    bool reserve(size_t bytes) {
        if (bytes <= cap) return true;
        if (base) { (void)hipHostFree(base); base = nullptr; cap = 0; }
        const size_t want = (bytes + (bytes >> 3) + ((size_t)2 << 20)) & ~(((size_t)2 << 20) - 1);   // 12 % head room: bakes of similar size do not re-pin
        if (hipHostMalloc((void**)&base, want, hipHostMallocDefault) != hipSuccess) { base = nullptr; (void)hipGetLastError(); return false; }
        cap = want; return true;
    }
};
// (the three HIP streams of a bake belong to the working set too: creating and destroying them costs ~0.3 ms per bake)
struct ArenaSet {
    DeviceArena tables, states, xchg; HostArena pinned;
    hipStream_t stream = nullptr, commStream = nullptr, placeStream = nullptr;
    ~ArenaSet() { for (hipStream_t s : { placeStream, commStream, stream }) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } }
};
struct ArenaPool {
    std::mutex mu; std::vector<std::unique_ptr<ArenaSet>> idle;
    std::unique_ptr<ArenaSet> acquire() {
        { std::lock_guard<std::mutex> g(mu); if (!idle.empty()) { std::unique_ptr<ArenaSet> a = std::move(idle.back()); idle.pop_back(); return a; } }
        return std::unique_ptr<ArenaSet>(new ArenaSet());
    }
    std::atomic<bool> retain{ true };
    // Idle sets kept: two whatever their size, and up to sixteen as long as the idle ones together stay below 1 GiB -- K caller threads that bake small
    // meshes on one baker (docs/integration_guide.md:434) would otherwise free and re-create K - 2 working sets (arenas: hipMalloc / hipFree, both
    // device-synchronising; three streams) on every round.
    static size_t bytes_of(const ArenaSet& a) { return a.tables.cap + a.states.cap + a.xchg.cap + a.pinned.cap; }
    void release(std::unique_ptr<ArenaSet> a) {
        std::unique_ptr<ArenaSet> drop;   // (freed outside the lock)
        {
            std::lock_guard<std::mutex> g(mu);
            size_t held = bytes_of(*a); for (const auto& i : idle) held += bytes_of(*i);
            if (retain.load() && (idle.size() < 2 || (idle.size() < 16 && held <= ((size_t)1 << 30)))) idle.push_back(std::move(a)); else drop = std::move(a);
        }
    }
    void trim() { std::vector<std::unique_ptr<ArenaSet>> drop; { std::lock_guard<std::mutex> g(mu); drop.swap(idle); } }   // (freed outside the lock)
};

// ---- device blocks of bake results, reused across bakes (hipMalloc / hipFree of a 1.3 GB block are synchronous and cost ~1 ms) ----
struct DevPool {
    struct Blk { void* p; size_t cap; bool used; bool exempt; };
    std::mutex mu; std::vector<Blk> blks; std::atomic<bool> retain{ true };
    ~DevPool() { for (auto& b : blks) (void)hipFree(b.p); }
    void* acquire(size_t bytes) {
        if (bytes == 0) bytes = 1;
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto& b : blks) if (!b.used && b.cap >= bytes && b.cap / 2 <= bytes + 4096) { b.used = true; return b.p; }
        }
        void* p = nullptr;
        const size_t cap = (bytes + 4095) & ~(size_t)4095;
        if (hipMalloc(&p, cap) != hipSuccess) {
            (void)hipGetLastError();
            trim(0);                                  // give the idle blocks back and retry once
            if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        std::lock_guard<std::mutex> g(mu);
        blks.push_back({ p, cap, true, false });
        return p;
    }
    void release(void* p) {
        if (!p) return;
        { std::lock_guard<std::mutex> g(mu); for (auto& b : blks) if (b.p == p) b.used = false; }
        trim(retain.load() ? 6 : 0, retain.load() ? (size_t)256 << 20 : 0);
    }
    // at most keepIdle idle blocks (two result sets: arrayData, descs, index) -- not counting up to 64 SMALL idle blocks of at most smallBytes together:
    // concurrent small bakes on one baker hand dozens of kilobyte-sized blocks back and forth, and hipFree synchronises the device
    void trim(size_t keepIdle, size_t smallBytes = 0) {
        std::vector<void*> drop;   // (hipFree outside the lock)
        {
            std::lock_guard<std::mutex> g(mu);
            size_t small = 0, smallCount = 0, idle = 0;
            for (auto& b : blks) if (!b.used) { if (b.cap <= ((size_t)4 << 20) && small + b.cap <= smallBytes && smallCount < 64) { small += b.cap; smallCount++; b.exempt = true; } else { b.exempt = false; idle++; } }
            for (size_t i = 0; i < blks.size() && idle > keepIdle; )
                if (!blks[i].used && !blks[i].exempt) { drop.push_back(blks[i].p); blks.erase(blks.begin() + (long)i); idle--; } else ++i;
        }
        for (void* p : drop) (void)hipFree(p);
    }
};

// ---- warm, pinned host memory for the (large) result array -------------------------------------------------
// A fresh 1.3 GB malloc costs more in page faults (70-100 ms) and munmap (95 ms) than the PCIe copy itself (24 ms at 57 GB/s), so
// with the DEFAULT allocator the arrayData block of a destroyed result is kept by its baker and handed to the next bake.  The blocks are
// page-locked (hipHostMalloc): device-to-host copies into them run on the SDMA engines, truly asynchronous and without occupying compute
// units -- a copy into pageable memory is a blit KERNEL that competes with the classification it is supposed to overlap (measured: the
// streamed bake's classification 37 ms instead of 28).  If pinning fails the block is ordinary memory, 2 MiB aligned (transparent huge
// pages) and pre-faulted by a few threads.  User-supplied allocators are always honoured as given.
struct HostPool {
    struct Blk { void* p; size_t cap; bool used; bool pinned; };
    std::mutex mu; std::vector<Blk> blks; std::atomic<bool> retain{ true };
    static constexpr size_t kMinBytes = 8u << 20, kHuge = 2u << 20;
    // Results between kSmallBytes and kMinBytes (a configs[1]-sized bake: 6.5 MB): a fresh malloc of that size is an mmap whose pages fault in under the
    // device-to-host copy -- 0.46 of the call's 1.16 ms -- and an munmap at ommCpuDestroyBakeResult.  They come from this pool too, but only kSmallInUse of them
    // at a time: an application that keeps thousands of small results alive must not find them all pinned (2 MB each at least).
    static constexpr size_t kSmallBytes = 256u << 10; static constexpr unsigned kSmallInUse = 8;
    bool wants(size_t bytes) {
        if (bytes >= kMinBytes) return true;
        if (bytes < kSmallBytes) return false;
        std::lock_guard<std::mutex> g(mu);
        unsigned n = 0; for (auto& b : blks) if (b.used && b.cap < kMinBytes) ++n;
        return n < kSmallInUse;
    }
    static void drop(const Blk& b) { if (b.pinned) (void)hipHostFree(b.p); else free(b.p); }
    ~HostPool() { for (auto& b : blks) drop(b); }
    // an IDLE block that fits, or null: never allocates (the zeroing ahead of a compressed result takes what the last bake left)
    void* acquire_idle(size_t bytes, size_t* cap, bool* pinned) {
        std::lock_guard<std::mutex> g(mu);
        for (auto& b : blks) if (!b.used && b.cap >= bytes && b.cap / 2 <= bytes + kHuge) { b.used = true; *cap = b.cap; *pinned = b.pinned; return b.p; }
        return nullptr;
    }
    void* acquire(size_t bytes, bool* pinned = nullptr) {
        if (pinned) *pinned = false;
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto& b : blks) if (!b.used && b.cap >= bytes && b.cap / 2 <= bytes + kHuge) { b.used = true; if (pinned) *pinned = b.pinned; return b.p; }
        }
        const size_t cap = (bytes + kHuge - 1) & ~(kHuge - 1);
        void* p = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) == hipSuccess && p) {   // (pinning touches every page: no pre-fault pass needed)
            std::lock_guard<std::mutex> g(mu);
            blks.push_back({ p, cap, true, true });
            if (pinned) *pinned = true;
            return p;
        }
        (void)hipGetLastError();
        p = aligned_alloc(kHuge, cap);
        if (!p) return nullptr;
        (void)madvise(p, cap, MADV_HUGEPAGE);
        unsigned nt = std::thread::hardware_concurrency(); nt = nt > 8 ? 8 : (nt ? nt : 1);
        if (cap < (64u << 20)) nt = 1;
        std::vector<std::thread> th;
        const size_t per = ((cap / nt) + kHuge - 1) & ~(kHuge - 1);
        auto touch = [p](size_t lo, size_t hi) { for (size_t o = lo; o < hi; o += 4096) ((volatile uint8_t*)p)[o] = 0; };
        for (unsigned k = 1; k < nt; ++k) { const size_t lo = per * k, hi = lo + per < cap ? lo + per : cap; if (lo < cap) th.emplace_back(touch, lo, hi); }
        touch(0, per < cap ? per : cap);
        for (auto& t : th) t.join();
        std::lock_guard<std::mutex> g(mu);
        blks.push_back({ p, cap, true, false });
        return p;
    }
    void release(void* p) {
        std::lock_guard<std::mutex> g(mu);
        // idle blocks kept: two large ones, and as many small ones as may be in use at once (<= 8 x 8 MB: concurrent callers of small bakes would otherwise
        // free and pin a block per bake); none with ommxBakerKnob_RetainMemory = 1
        size_t idle[2] = { 0, 0 };
        for (auto& b : blks) { if (b.p == p) b.used = false; if (!b.used) idle[b.cap < kMinBytes ? 1 : 0]++; }
        const size_t keep[2] = { retain.load() ? (size_t)2 : (size_t)0, retain.load() ? (size_t)kSmallInUse : (size_t)0 };
        for (size_t i = 0; i < blks.size(); ) {
            const int k = blks[i].cap < kMinBytes ? 1 : 0;
            if (!blks[i].used && idle[k] > keep[k] && (blks[i].p != p || keep[k] == 0)) { drop(blks[i]); blks.erase(blks.begin() + (long)i); idle[k]--; } else ++i;
        }
    }
    void trim() {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < blks.size(); ) if (!blks[i].used) { drop(blks[i]); blks.erase(blks.begin() + (long)i); } else ++i;
    }
};

// ---- device-to-host copies on the SDMA engines ------------------------------------------------------------------------------------
// hipMemcpyAsync moves device -> host data with a blit KERNEL (__amd_rocclr_copyBuffer, also into pinned memory): a kernel that sits on
// compute-unit slots waiting for PCIe, next to the persistent classification launch it is meant to overlap -- measured: the classification
// launches that ran beside such a copy took 2.5x as long.  The streamed result therefore goes through the HSA runtime that HIP itself
// sits on: hsa_amd_memory_async_copy runs on one of the chip's DMA engines and needs no compute unit.  Destination must be pinned
// (the baker's result pool); anything else keeps to hipMemcpyAsync.
struct SdmaCopier {
    hsa_agent_t gpu{ 0 }, cpu{ 0 }; hsa_signal_t sig{ 0 }; bool ok = false; int64_t pending = 0;
    struct Find { int wantOrdinal, seen; uint32_t wantBdf, wantDomain; bool haveBdf; hsa_agent_t gpu, cpu; bool gotGpu, gotCpu; };
    static hsa_status_t visit(hsa_agent_t a, void* u) {
        Find& f = *(Find*)u; hsa_device_type_t t;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        if (t == HSA_DEVICE_TYPE_CPU && !f.gotCpu) { f.cpu = a; f.gotCpu = true; }
        if (t == HSA_DEVICE_TYPE_GPU) {
            uint32_t bdf = 0; const bool hb = hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS;
            // (bus / device / function alone is not unique on hosts with several PCI domains: the domain has to agree too, where the runtime reports it)
            uint32_t dom = 0; const bool hd = hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom) == HSA_STATUS_SUCCESS;
            const bool match = f.haveBdf && hb ? (bdf & 0xFFFFu) == f.wantBdf && (!hd || dom == f.wantDomain) : f.seen == f.wantOrdinal;
            if (match && !f.gotGpu) { f.gpu = a; f.gotGpu = true; }
            f.seen++;
        }
        return HSA_STATUS_SUCCESS;
    }
    bool open(int hipDevice) {
        if (hsa_init() != HSA_STATUS_SUCCESS) return false;   // (reference counted: HIP holds the runtime open already)
        Find f; memset(&f, 0, sizeof f); f.wantOrdinal = hipDevice;
        char bus[32] = { 0 };
        if (hipDeviceGetPCIBusId(bus, sizeof bus, hipDevice) == hipSuccess) { unsigned dom = 0, b = 0, d = 0, fn = 0; if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &fn) == 4) { f.wantBdf = (b << 8) | (d << 3) | fn; f.wantDomain = dom; f.haveBdf = true; } }
        if (hsa_iterate_agents(visit, &f) != HSA_STATUS_SUCCESS || !f.gotGpu || !f.gotCpu) { (void)hsa_shut_down(); return false; }
        gpu = f.gpu; cpu = f.cpu;
        if (hsa_signal_create(0, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) { (void)hsa_shut_down(); return false; }
        ok = true; return true;
    }
    ~SdmaCopier() { if (ok) { (void)wait(); (void)hsa_signal_destroy(sig); (void)hsa_shut_down(); } }
    bool copy_to_host(void* dstHostPinned, const void* srcDevice, size_t bytes) {   // asynchronous; wait() before the data is read
        hsa_signal_add_relaxed(sig, 1); pending++;
        if (hsa_amd_memory_async_copy(dstHostPinned, cpu, srcDevice, gpu, bytes, 0, nullptr, sig) != HSA_STATUS_SUCCESS) { hsa_signal_subtract_relaxed(sig, 1); pending--; return false; }
        return true;
    }
    bool wait() {
        if (!ok || pending == 0) return true;
        const hsa_signal_value_t v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
        pending = 0;
        if (v != 0) { hsa_signal_store_relaxed(sig, 0); return false; }   // (negative: a copy failed)
        return true;
    }
};

// A baker works on ONE HIP device: the device that is current on the calling thread when its first texture is created.  HIP's current
// device is per thread (default 0), so every later entry point switches to the baker's device for the duration of the call and restores
// the caller's -- a worker thread of a multi-GPU host can bake without calling hipSetDevice itself, and a texture is never sampled from
// the wrong GPU.  One process driving several GPUs uses one baker per device.
struct DeviceScope {
    int prev = -1; bool changed = false;
    explicit DeviceScope(int dev) { if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess; }
    ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete; DeviceScope& operator=(const DeviceScope&) = delete;
};

struct Baker {
    Allocator mem; Logger log; ommBakerType type;
    std::atomic<int> device{ -1 };   // bound by the first ommCpuCreateTexture / deserialised texture
    int bind_device() {
        int d = device.load();
        if (d >= 0) return d;
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return -1;
        int expected = -1;
        return device.compare_exchange_strong(expected, cur) ? cur : expected;
    }
    std::shared_ptr<HostPool> hostPool = std::make_shared<HostPool>();
    std::shared_ptr<DevPool> devPool = std::make_shared<DevPool>();
    std::shared_ptr<ArenaPool> arenas = std::make_shared<ArenaPool>();   // device working sets, one per bake in flight
    // multi-device ommCpuBake (ommxBakerKnob_Devices): one shadow baker per further device (own pools and working sets; allocator, logger and knobs of this one)
    std::mutex peersMu; std::vector<std::unique_ptr<Baker>> peers;
    // helper threads of the compressed result (host_expand.h): started by the first bake that may use them (ommCpuBakeFlags_EnableInternalThreads)
    std::mutex workersMu; std::shared_ptr<WorkerPool> workers; unsigned workersWanted = 0;
    std::shared_ptr<WorkerPool> worker_pool(unsigned threads) {
        std::lock_guard<std::mutex> g(workersMu);
        const unsigned want = threads > 1 ? threads - 1 : 0;   // (the calling thread is one of them)
        if (!workers || workersWanted != want) { workers = std::make_shared<WorkerPool>(want); workersWanted = want; }   // (a bake in flight keeps the pool it took)
        return workers;
    }
    std::mutex timingsMu; ommxBakeTimings timings; bool haveTimings = false;
    std::atomic<uint64_t> knobs[ommxBakerKnob_MAX_NUM];   // ommxSetBakerKnob: 0 = default
    std::atomic<uint64_t> lastCompressedArrayBytes{ 0 };   // arrayDataSize of the last bake whose result took the compressed transfer (what the next one zeroes ahead)
    std::atomic<uint32_t> activeBakes{ 0 };               // ommCpuBake calls inside bake_impl right now (callers may bake concurrently on one baker)
    Baker() { for (auto& k : knobs) k.store(0); }
    uint64_t knob(ommxBakerKnob k) const { return knobs[k].load(std::memory_order_relaxed); }
};

// HIP events on the bake's own stream (torch / the caller never see this stream)
struct EventTimer {
    hipStream_t s; hipEvent_t ev[24]; int n = 0;
    explicit EventTimer(hipStream_t st) : s(st) { for (auto& e : ev) e = nullptr; }
    ~EventTimer() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); }
    int mark() { if (n >= 24) return -1; if (hipEventCreate(&ev[n]) != hipSuccess) return -1; (void)hipEventRecord(ev[n], s); return n++; }
    float ms(int a, int b) const { float v = 0.f; if (a < 0 || b < 0 || hipEventElapsedTime(&v, ev[a], ev[b]) != hipSuccess) return 0.f; return v; }
};
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct TexMip { int w = 0, h = 0; void* texels = nullptr; uint32_t* sat = nullptr; };
struct Texture {
    Allocator mem; const Logger* log = nullptr;
    ommCpuTextureFormat format = ommCpuTextureFormat_MAX_NUM; ommCpuTextureFlags flags = ommCpuTextureFlags_None; float alphaCutoff = -1.f;
    std::vector<TexMip> mips;
    int device = -1;   // the HIP device the texels live on (the baker's)
    // multi-device ommCpuBake (ommxBakerKnob_Devices): copies of the texels and summed-area tables on the other devices, made by the first bake that needs them
    std::mutex replicaMu; std::vector<std::unique_ptr<Texture>> replicas;
    ~Texture() { for (auto& m : mips) { if (m.texels) (void)hipFree(m.texels); if (m.sat) (void)hipFree(m.sat); } }
};

struct BakeResult {
    Allocator mem;
    void* arrayData = nullptr; ommCpuOpacityMicromapDesc* descs = nullptr;
    ommCpuOpacityMicromapUsageCount* arrayHist = nullptr; ommCpuOpacityMicromapUsageCount* indexHist = nullptr;
    int32_t* index = nullptr;
    float* triArea = nullptr;       // UV-space area per input triangle (bake_cpu_impl.cpp:1904-1915): ommDebugGetStats2's side channel
    std::shared_ptr<HostPool> pool; // set when arrayData came from the baker's warm pool (results may outlive their baker)
    ommCpuBakeResultDesc desc;
    BakeResult() { memset(&desc, 0, sizeof desc); }
    ~BakeResult() { if (pool) pool->release(arrayData); else mem.release(arrayData); mem.release(descs); mem.release(arrayHist); mem.release(indexHist); mem.release(index); mem.release(triArea); }
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
// what a bake's first transfer carries: the zeroed counters block (its 256-byte arena slot) and, in the slot behind it, the digest table -- one copy instead
// of a copy and a fill (two fill launches: 200 bytes are not a multiple of 16)
struct BakeHead { uint8_t counters[256]; uint64_t uniform[kNumLevels * 4]; BakeHead() { memset(counters, 0, sizeof counters); memcpy(uniform, uniform_digests().v, sizeof uniform); } };
static_assert(sizeof(SetupCounters) <= 256, "the counters block must fit its arena slot");
const BakeHead& bake_head() { static const BakeHead h; return h; }

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

uint32_t device_cu_count() // of the current device; sizes the persistent classification grid
{
    static std::mutex mu; static int cached[64]; static bool have[64];
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> g(mu);
    if (!have[dev]) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; cached[dev] = n; have[dev] = true; }
    return (uint32_t)cached[dev];
}

// Device-resident result of a bake (what ommxBakeDevice hands out, and what ommCpuBake copies to the host).
struct DeviceResult {
    uint8_t* arrayData = nullptr; uint64_t arrayDataSize = 0;
    ommCpuOpacityMicromapDesc* descs = nullptr; uint32_t numDescs = 0;
    void* index = nullptr; uint32_t numTris = 0; ommIndexFormat indexFormat = ommIndexFormat_UINT_32;
    uint32_t hist[2 * kNumLevels]; int bits = 2;
    const float* triAreaScratch = nullptr; // per-triangle UV areas in the bake's arena: valid until the session ends (ommCpuBake copies them out)
    std::shared_ptr<DevPool> pool;   // the baker's (results may outlive their baker)
    // ommCpuBake's compressed transfer: asked right before the gather, when the array's size is known and every OMM is a multiple of 16 bytes; may hand out a byte per
    // 16-byte unit and a zeroed word per 256-unit block (+ 1) for the gather to fill with the exchange codec's codes and raw counts (launch_gather_omms)
    std::function<bool(uint64_t arrayDataSize, uint8_t** unitCodes, uint32_t** blockRawCounts)> gatherCodes;
    DeviceResult() { memset(hist, 0, sizeof hist); }
    DeviceResult(const DeviceResult&) = delete;
    DeviceResult& operator=(const DeviceResult&) = delete;
    void* dev_alloc(size_t bytes) { return pool ? pool->acquire(bytes) : nullptr; }
    ~DeviceResult() { if (pool) { pool->release(arrayData); pool->release(descs); pool->release(index); } }
};

struct DeviceInputs {           // raw triangle data, device resident
    const void* texCoords; const void* indices; const uint8_t* perTriLevels;
};

// Multi-GPU: state kept between the phases of a sharded bake (ommxSharded*).  The rank classifies its share of the active
// work items; per-item metadata and surviving blocks are exchanged by the caller's collectives; the tail is replicated.
struct ShardCtx {
    uint32_t rank = 0, world = 1;
    ShardBounds bounds;
    TailInputs ti; TailOutputs to; TailCounts counts;
    SetupCounters hc;
    uint8_t *dStates = nullptr, *dActive = nullptr, *dLevel = nullptr, *dScratch = nullptr; uint64_t* dStateOfs = nullptr; uint32_t *dMask = nullptr, *dActiveIds = nullptr;
    int32_t* dIndex = nullptr; uint32_t *dArrayHist = nullptr, *dIndexHist = nullptr;
    size_t scratchBytes = 0; uint32_t flags = 0, T = 0; int bits = 2; bool asyncBegin = false;
    bool mergeStates = false;   // host-tail bakes: the packed states are zeroed first so that a SUM all-reduce over them is a merge
    int ev[5] = { -1, -1, -1, -1, -1 };   // HIP event marks of Begin: setup | triage | classify | digest
    // exchange buffers: carved from the session's arenas (tables: dMeta .. dTotals; xchg: contribution + gather staging), nothing to free
    uint32_t* dMeta = nullptr; uint8_t* dOwner = nullptr; uint64_t *dCofs = nullptr, *dTotals = nullptr; uint8_t *dContrib = nullptr, *dGathered = nullptr;
    uint8_t *dComp = nullptr, *dGatherComp = nullptr, *dCodecScratch = nullptr; uint32_t* dCompSize = nullptr; uint64_t compCap = 0; size_t codecScratchBytes = 0;   // block exchange codec (RCCL path): own stream, all ranks' streams, own size word, count / scan scratch
    uint64_t totals[kMaxRanks]; uint64_t strideBytes = 0;
    const float* dTriArea = nullptr;   // per-triangle UV areas in the session's arena (ommDebugGetStats2's side channel)
};

// opt-in lossy reducers (near-duplicate merge, maxArrayDataSize): classification on the device, serial tail on the host
struct HostTailRequest { std::vector<HostItem> items; };

// ommCpuBake: the result has to end up in host memory, and the 1.3 GB of arrayData of the metric configuration need 23 ms of PCIe time -- nearly
// as long as the classification itself.  The final ORDER of the blocks is known up front (descending level / spatial key / item index over the items
// that are emitted), so the tile queue of the levels >= 6 is cut into `chunks` consecutive ranges of that order, which one persistent launch drains in
// turn; when a range is complete (device-side count), its blocks are packed behind those of the earlier ranges -- a contiguous piece of the final
// arrayData -- on a second stream, and ONE SDMA copy moves that piece to its final place in the caller's array while the launch classifies the
// following ranges (tail_kernels.hip: "Streamed result").  The placement is verified against the ordinary tail at the end; on a mismatch (a duplicate
// block whose first occurrence was classified later) the bake falls back to the ordinary gather + copy.
struct StreamOut {
    // in
    uint32_t chunksWanted = 0;        // 0 = decide from the size of the bake
    bool forced = false;              // (knob set: stream whatever the size)
    bool compressedAvailable = false; // the caller can send the finished array as a codec stream instead (helper threads): stream only when that is the faster of the two
    ArenaSet* set = nullptr; hipStream_t copyStream = nullptr, placeStream = nullptr;
    void* allocUser = nullptr; uint8_t* (*alloc)(void* user, uint64_t upperBoundBytes, bool* pinned) = nullptr;   // the host arrayData, before the first block leaves
    int device = 0;
    // out
    bool used = false, fellBack = false; uint32_t chunks = 0; uint64_t streamedBytes = 0;
    double classifyEndMs = 0, lastByteMs = 0;
    SdmaCopier sdma;   // copies in flight when bake_core returns: the caller queues its small read-backs behind the bake, THEN waits for these (finish())
    bool finish() { bool ok = true; if (copyStream) ok = hipStreamSynchronize(copyStream) == hipSuccess; ok = sdma.wait() && ok; lastByteMs = now_ms(); return ok; }
};
struct StreamCtx {   // what the hook behind a classification launch needs
    hipStream_t stream, place; hipEvent_t* fences; const uint32_t* activeIds; uint32_t numActive; StreamSegment proto; void* scratch; size_t scratchBytes;
    uint64_t* digests; unsigned long long* cursor; uint8_t* stage; uint64_t* placed; uint32_t* ctl; unsigned long long* hostCursor; hipEvent_t* events; uint32_t numEvents, recorded; bool ok;
    const uint32_t* queueCtl; bool paired;                 // paired: two queue sections per range (bake_kernels.h: ClassifyChunks)
    const uint32_t* earlyList; uint32_t earlyCapacity; const uint8_t* itemLevel;   // the early class ordered by the range it is classified with (ctl: start / count per range)
    double waitSeconds;   // how long the placement stream waits for a range before it gives the stream up (scaled with the estimated classification time)
};
// in front of the persistent launch: the placement stream starts behind the tile triage (the section tails are final from here on)
void stream_mark_hook(void* user)
{
    StreamCtx& c = *(StreamCtx*)user;
    c.ok = c.ok && hipEventRecord(c.fences[0], c.stream) == hipSuccess && hipStreamWaitEvent(c.place, c.fences[0], 0) == hipSuccess;
}
void stream_hook(void* user, uint32_t chunk, const ClassifySegment* segs, uint32_t numSegs, bool last)
{
    // The placement runs on a second, high-priority stream next to the ONE persistent classification launch: a one-lane kernel holds that stream until
    // the range's section of the tile queue is complete (device-side count, no launch boundary, no host round trip), then the digest / placement kernels
    // of the range run in the workgroup slot the classification leaves free on every CU.
    StreamCtx& c = *(StreamCtx*)user;
    if (chunk >= c.numEvents) { c.ok = false; return; }
    if (last) c.ok = c.ok && hipEventRecord(c.fences[1], c.stream) == hipSuccess && hipStreamWaitEvent(c.place, c.fences[1], 0) == hipSuccess;   // (the lower levels: behind everything)
    else launch_stream_wait_sections(c.queueCtl, c.paired ? 2u * chunk : chunk, c.paired ? 2u : 1u, c.ctl, c.waitSeconds, c.place);
    // CalcDigest (bake_cpu_impl.cpp:1038-1040).  A range of the levels >= 6: its own items (the early ones among them have their digest from this range
    // or an earlier one) and, in the same launch, the early items classified WITH this range -- the families that start in it, wherever their members lie --,
    // which then enter the table ahead of the range's placement
    const bool lists = !last && !c.proto.disableDedup;
    StreamSegment e = c.proto; e.ids = c.earlyList; e.count = c.earlyList ? c.earlyCapacity : 0u; e.level = 6; e.range = chunk;
    e.liveStart = c.ctl + kStreamCtlEarlyStart + chunk; e.liveCount = c.ctl + kStreamCtlEarlyCount + chunk; e.itemLevel = c.itemLevel;
    for (uint32_t k = 0; lists && k < (numSegs ? numSegs : 1u); ++k) {
        DigestLists D; memset(&D, 0, sizeof D);
        if (k < numSegs) { D.ids = c.activeIds + segs[k].first; D.count = segs[k].count; D.level = segs[k].level; D.only = c.proto.early; D.want = 0; }
        if (k == 0) { D.listB = e.ids; D.capacityB = e.count; D.liveStart = e.liveStart; D.liveCount = e.liveCount; D.itemLevel = e.itemLevel; }
        launch_digest_lists(c.proto.states, c.proto.stateOfs, D, (uint32_t)c.proto.bits, c.digests, c.place);
    }
    if (lists && e.count) launch_stream_insert_list(e, c.numActive, c.scratch, c.scratchBytes, c.place);
    for (uint32_t k = 0; k < numSegs && c.ok; ++k) {
        StreamSegment g = c.proto; g.ids = c.activeIds + segs[k].first; g.count = segs[k].count; g.level = segs[k].level; g.range = chunk;
        if (last && !g.disableDedup) launch_digest(g.states, g.stateOfs, g.ids, g.count, g.level, (uint32_t)g.bits, c.digests, c.place);   // (the levels below 6)
        c.ok = run_stream_segment(g, c.numActive, c.scratch, c.scratchBytes, c.cursor, c.stage, c.placed, c.ctl, c.place) == hipSuccess;
    }
    launch_stream_publish(c.cursor, c.hostCursor + chunk, c.place);
    c.ok = c.ok && hipEventRecord(c.events[chunk], c.place) == hipSuccess;
    if (c.ok) c.recorded = chunk + 1u;
}
struct MarkCtx { EventTimer* et; int mark, markGeneric; };
void mark_hook(void* user) { MarkCtx& c = *(MarkCtx*)user; c.mark = c.et->mark(); }
void mark_generic_hook(void* user) { MarkCtx& c = *(MarkCtx*)user; c.markGeneric = c.et->mark(); }

// One byte per micro-triangle of every work item on the host, for the serial reducers (host_tail.cpp): near-duplicate merge and the
// maxArrayDataSize budget work on the reference's own data layout (bake_cpu_impl.cpp:401-411).  Synchronises the stream.
ommResult gather_host_items(const Logger& L, hipStream_t stream, uint32_t U, uint32_t T, const SetupCounters& hc, const float* dUv, const uint8_t* dLevel,
                            const uint8_t* dActive, const uint32_t* dMask, const uint64_t* dStateOfs, const uint8_t* dStates, const int32_t* dTriToItem, int bits,
                            std::vector<HostItem>& items)
{
    {   // refuse cleanly when that cannot fit in host memory instead of dying in std::bad_alloc half way
        const uint64_t bytes = 2 * hc.stateBytes * (bits == 1 ? 2u : 1u);   // the packed states of the non-uniform items, once as read back and once per item
        const uint64_t phys = (uint64_t)sysconf(_SC_PHYS_PAGES) * (uint64_t)sysconf(_SC_PAGE_SIZE);
        if (bytes > phys / 2) return L.failure("[Failure] - near-duplicate merging / maxArrayDataSize need the packed states of every non-uniform work item on the host: not enough host memory for this bake");
    }
    std::vector<float> hUv((size_t)U * 6); std::vector<uint8_t> hLevel(U), hActive(U), hStates((size_t)hc.stateBytes);
    std::vector<uint32_t> hMask(U); std::vector<uint64_t> hOfs(U); std::vector<int32_t> hTri(T);
    bool ok = true;
    if (U) {
        ok = HIP_OK(hipMemcpyAsync(hUv.data(), dUv, (size_t)U * 24, hipMemcpyDeviceToHost, stream)) && HIP_OK(hipMemcpyAsync(hLevel.data(), dLevel, U, hipMemcpyDeviceToHost, stream))
          && HIP_OK(hipMemcpyAsync(hActive.data(), dActive, U, hipMemcpyDeviceToHost, stream)) && HIP_OK(hipMemcpyAsync(hMask.data(), dMask, (size_t)U * 4, hipMemcpyDeviceToHost, stream))
          && HIP_OK(hipMemcpyAsync(hOfs.data(), dStateOfs, (size_t)U * 8, hipMemcpyDeviceToHost, stream));
    }
    if (ok && hc.stateBytes) ok = HIP_OK(hipMemcpyAsync(hStates.data(), dStates, (size_t)hc.stateBytes, hipMemcpyDeviceToHost, stream));
    if (ok && T) ok = HIP_OK(hipMemcpyAsync(hTri.data(), dTriToItem, (size_t)T * 4, hipMemcpyDeviceToHost, stream));
    ok = ok && HIP_OK(hipStreamSynchronize(stream));
    if (!ok) return L.failure("[Failure] - device to host transfer of the micro-triangle states failed");
    items.resize(U);
    for (uint32_t i = 0; i < U; ++i) {
        HostItem& it = items[i];
        it.level = hLevel[i]; it.format = bits; memcpy(it.uv, &hUv[(size_t)i * 6], 24);
        const size_t n = (size_t)1 << (2 * it.level);
        // uniform items stay (state, level); the others keep the device's 2-bit packing (1-bit states of a 2-state bake are widened to it)
        if (!hActive[i]) { uint32_t st = 0; while (!((hMask[i] >> st) & 1u) && st < 3) ++st; it.uniform = (int)st; }
        else {
            const uint8_t* p = hStates.data() + hOfs[i];
            it.uniform = -1; it.packed.assign(n >= 4 ? n / 4 : 1, 0);
            if (bits == 2) memcpy(it.packed.data(), p, n >= 4 ? n / 4 : 1);
            else for (size_t u = 0; u < n; ++u) it.packed[u >> 2] = (uint8_t)(it.packed[u >> 2] | (((p[u >> 3] >> (u & 7)) & 1u) << ((u & 3) << 1)));
        }
    }
    for (uint32_t t = 0; t < T; ++t) if (hTri[t] >= 0) items[(size_t)hTri[t]].prims.push_back(t);
    return ommResult_SUCCESS;
}

// the serial tail itself (reference: bake_cpu_impl.cpp:1068-1688 on the classified states) + index narrowing in place (:1872-1902)
ommResult run_host_tail_for(const ommCpuBakeInputDesc& d, uint32_t T, std::vector<HostItem>& items, HostTailResult& hres, ommIndexFormat& ifmt)
{
    const uint32_t fl = (uint32_t)d.bakeFlags;
    HostTailDesc td; td.format = (int)d.format; td.disableSpecial = (fl & (1u << 1)) != 0; td.disableDedup = (fl & (1u << 3)) != 0;
    td.nearDup = (fl & (1u << 4)) != 0; td.nearDupBrute = (fl & (1u << 10)) != 0; td.rejectionThreshold = d.rejectionThreshold;
    td.nearDupFactor = d.nearDuplicateDeduplicationFactor; td.maxArrayDataSize = d.maxArrayDataSize; td.numTris = T; td.unresolved = (int32_t)d.unresolvedTriState;
    if (run_host_tail(td, items, hres)) return ommResult_FAILURE;
    hres.index.resize(T ? T : 1);
    ifmt = ommIndexFormat_UINT_32;
    const bool allow8 = (fl & (1u << 6)) != 0, force32 = (fl & (1u << 2)) != 0;
    if (allow8 && T <= 127 && !force32) { int8_t* p8 = (int8_t*)hres.index.data(); for (uint32_t i = 0; i < T; ++i) { const int32_t v = hres.index[i]; p8[i] = (int8_t)v; } ifmt = ommIndexFormat_UINT_8; }
    else if (T <= 32767 && !force32) { int16_t* p16 = (int16_t*)hres.index.data(); for (uint32_t i = 0; i < T; ++i) { const int32_t v = hres.index[i]; p16[i] = (int16_t)v; } ifmt = ommIndexFormat_UINT_16; }
    return ommResult_SUCCESS;
}

// The bake proper: bake_cpu_impl.cpp:1923-1985 re-organised for the device.  `din` points at device copies of the
// caller's texCoords / indexBuffer / subdivisionLevels; `hostDesc` (may be null for device-resident callers) is the
// same desc with host pointers, needed only by the serial fallbacks.
ommResult bake_core(Baker& baker, const ommCpuBakeInputDesc& d, const DeviceInputs& din, const ommCpuBakeInputDesc* hostDesc,
                    DeviceArena* arena, DeviceArena* statesArena, hipStream_t stream, EventTimer& et, DeviceResult& R, ommxBakeTimings& tm,
                    ShardCtx* sh = nullptr, HostTailRequest* ht = nullptr, StreamOut* so = nullptr)
{
    const Logger& L = baker.log;
    const uint32_t flags = (uint32_t)d.bakeFlags;
    if (!R.pool) R.pool = baker.devPool;
    const Texture& tex = *untag<Texture>(d.texture);
    const uint32_t T = d.indexCount / 3u;
    const int bits = (int)d.format;
    // DisableFineClassification with the 2-state format (internal flag bit 9): unresolved micro-triangles keep UnknownOpaque (3), which has no 1-bit form -- the
    // reference digests the unpacked states and ORs `3 << (i & 7)` into the packed bytes (bake_cpu_impl.cpp:1811).  The states are therefore kept in the
    // 2-bit packing up to the final gather, which applies that rule (launch_gather_omms: storeBits != bits).
    const int storeBits = (bits == 1 && (flags & (1u << 9))) ? 2 : bits;
    const uint32_t maxItems = T ? T : 1;

    // ---- device layout (worst case: every triangle is its own work item) ----
    const size_t setupBytes = setup_scratch_bytes(T), tailBytes = tail_scratch_bytes(maxItems, T), streamScratch = so ? stream_scratch_bytes(maxItems) : 0;
    const size_t scratchBytes = std::max(std::max(setupBytes, tailBytes), streamScratch);
    const size_t i32 = pad256((size_t)maxItems * 4), i64 = pad256((size_t)maxItems * 8);
    const size_t shardBytes = sh ? pad256((size_t)maxItems * 16) + pad256(maxItems) + i64 + pad256(sizeof(uint64_t) * kMaxRanks) : 0;
    // streamed result: placed offset per item, cursor + control words; preview: collapsed UVs, 16-byte state slots, offsets, masks, early flags
    const size_t streamBytes = so ? i64 + 512 + pad256((size_t)maxItems * 24) + pad256((size_t)maxItems * kPreviewSlotBytes) + i64 + i32 * 3 + pad256(maxItems) + pad256(sizeof(unsigned long long) * kFineSlots * kFineStride) : 0;
    const size_t need = pad256((size_t)maxItems * 24) + 3 * pad256(maxItems) + i64 * 2 + i32 * 13 + pad256(sizeof(SetupCounters)) + 4096 + pad256(sizeof(unsigned long long) * kFineSlots * kFineStride) + pad256(scratchBytes) + shardBytes + streamBytes;
    if (!arena->reserve(need)) return L.failure("[Failure] - out of device memory for the bake working set");
    float* dUv = arena->take<float>((size_t)maxItems * 6);
    uint8_t* dLevel = arena->take<uint8_t>(maxItems); uint8_t* dDegen = arena->take<uint8_t>(maxItems); uint8_t* dActive = arena->take<uint8_t>(maxItems);
    uint64_t* dStateOfs = arena->take<uint64_t>(maxItems); uint64_t* dDigests = arena->take<uint64_t>(maxItems);
    uint32_t* dItemIds = arena->take<uint32_t>(maxItems); uint32_t* dActiveIds = arena->take<uint32_t>(maxItems);
    int32_t* dTriToItem = arena->take<int32_t>(maxItems); int32_t* dIndex = arena->take<int32_t>(maxItems);
    uint32_t* dMask = arena->take<uint32_t>(maxItems); uint32_t* dKnown = arena->take<uint32_t>(maxItems);
    int32_t* dSpecial = arena->take<int32_t>(maxItems); uint32_t* dRep = arena->take<uint32_t>(maxItems);
    uint32_t* dOrder = arena->take<uint32_t>(maxItems); uint32_t* dDstOfs = arena->take<uint32_t>(maxItems);
    uint32_t* dSizes = arena->take<uint32_t>(maxItems); int32_t* dItemValue = arena->take<int32_t>(maxItems);
    float* dTriArea = arena->take<float>(maxItems); R.triAreaScratch = dTriArea;
    SetupCounters* dCounters = arena->take<SetupCounters>(1);
    uint64_t* dUniformDigest = arena->take<uint64_t>(kNumLevels * 4);
    uint32_t* dArrayHist = arena->take<uint32_t>(kNumLevels); uint32_t* dIndexHist = arena->take<uint32_t>(kNumLevels); uint32_t* dErr = arena->take<uint32_t>(1);
    unsigned long long* dFine = arena->take<unsigned long long>(kFineSlots * kFineStride); // striped statistic counter (bake_types.h)
    uint8_t* dScratch = arena->take<uint8_t>(scratchBytes);
    if (sh) { sh->dMeta = arena->take<uint32_t>((size_t)maxItems * 4); sh->dOwner = arena->take<uint8_t>(maxItems); sh->dCofs = arena->take<uint64_t>(maxItems); sh->dTotals = arena->take<uint64_t>(kMaxRanks); }
    uint64_t* dPlaced = nullptr; unsigned long long* dCursor = nullptr; uint32_t* dStreamCtl = nullptr;
    float* dUv2 = nullptr; uint8_t *dStates2 = nullptr, *dEarly = nullptr; uint32_t *dEarlyList = nullptr, *dEarlyLead = nullptr; uint64_t* dOfs2 = nullptr; uint32_t* dMask2 = nullptr; unsigned long long* dFine2 = nullptr;
    if (so) {
        dPlaced = arena->take<uint64_t>(maxItems); dCursor = arena->take<unsigned long long>(1); dStreamCtl = arena->take<uint32_t>(kStreamCtlWords);
        dUv2 = arena->take<float>((size_t)maxItems * 6); dStates2 = arena->take<uint8_t>((size_t)maxItems * kPreviewSlotBytes); dOfs2 = arena->take<uint64_t>(maxItems);
        dMask2 = arena->take<uint32_t>(maxItems); dEarly = arena->take<uint8_t>(maxItems); dEarlyList = arena->take<uint32_t>(maxItems); dEarlyLead = arena->take<uint32_t>(maxItems); dFine2 = arena->take<unsigned long long>(kFineSlots * kFineStride);
    }
    if (arena->used > arena->cap) return L.failure("[Failure] - internal error: the working-set layout exceeds its reservation");

    // ---- SetupWorkItems (bake_cpu_impl.cpp:589-660) on the device ----
    const double coreT0 = now_ms();
    const int e0 = et.mark();
    SetupParams S; memset(&S, 0, sizeof S);
    S.texCoords = din.texCoords; S.indices = din.indices; S.perTriLevels = din.perTriLevels;
    S.stride = d.texCoordStrideInBytes ? d.texCoordStrideInBytes : (d.texCoordFormat == ommTexCoordFormat_UV32_FLOAT ? 8u : 4u);
    S.uvFormat = d.texCoordFormat; S.indexFormat = d.indexFormat; S.numTris = T;
    S.globalLevel = d.maxSubdivisionLevel; S.dynScale = d.dynamicSubdivisionScale; S.edgeHeuristic = (flags & (1u << 11)) != 0;   // (EnableEdgeHeuristic, bake_cpu_impl.cpp:48,547)
    S.texW = tex.mips[0].w; S.texH = tex.mips[0].h; S.disableDedup = (flags & (1u << 3)) != 0;
    S.wantWorkload = 1;   // (also the input of the two shape decisions below: streamed result, deferred generic pass)
    S.degenerateInvalid = (flags & (1u << 8)) != 0;
    const bool checkWorkload = ((flags & (1u << 5)) != 0) || d.maxWorkloadSize != 0xFFFFFFFFFFFFFFFFull;
    S.format = (int)d.format;   // (per-triangle formats other than the global one are refused above: one format per bake)
    bool ok = (const uint8_t*)dUniformDigest == (const uint8_t*)dCounters + 256   // (adjacent arena slots: see BakeHead)
           && HIP_OK(hipMemcpyAsync(dCounters, &bake_head(), sizeof(BakeHead), hipMemcpyHostToDevice, stream));
    ok = ok && HIP_OK(run_setup_fetch(S, dScratch, scratchBytes, dCounters, dKnown, maxItems, (uint32_t*)dFine, 2u * kFineSlots * kFineStride, dTriArea, stream));
    if (!ok) return L.failure("[Failure] - no usable HIP device (the MI355X baker has no CPU fallback)");
    SetupCounters hc; memset(&hc, 0, sizeof hc);
    if (d.dynamicSubdivisionScale > 0.f && T) { // degenerate triangles under dynamic subdivision need glibc's log2f: host (bake_cpu_impl.cpp:511-528)
        if (!HIP_OK(hipMemcpyAsync(&hc, dCounters, sizeof hc, hipMemcpyDeviceToHost, stream)) || !HIP_OK(hipStreamSynchronize(stream)))
            return L.failure("[Failure] - device work-item setup failed");
        if (hc.numPending) {
            std::vector<uint32_t> pend(hc.numPending); std::vector<float> puv((size_t)hc.numPending * 6); std::vector<uint8_t> plv(hc.numPending);
            struct PoolBlock { DevPool* pool; void* p; ~PoolBlock() { if (p) pool->release(p); } } tmp{ baker.devPool.get(), baker.devPool->acquire((size_t)hc.numPending * 24 + 256) };
            if (!tmp.p) return L.failure("[Failure] - device work-item setup failed");
            if (!HIP_OK(copy_pending_to_host(dScratch, scratchBytes, T, hc.numPending, pend.data(), puv.data(), tmp.p, stream))) return L.failure("[Failure] - device work-item setup failed");
            {   // (tiny triangles under dynamic subdivision are degenerate by the tens of thousands -- configs[4]: 1.5 ms of log2f on one thread; four share it)
                ommCpuBakeInputDesc tmpDesc = d; tmpDesc.subdivisionLevels = nullptr;   // per-triangle overrides were already honoured on the device
                auto part = [&](uint32_t k0, uint32_t k1) {
                    for (uint32_t k = k0; k < k1; ++k) { HostTri t; memcpy(t.p, &puv[(size_t)k * 6], 24); plv[k] = (uint8_t)level_for_primitive(tmpDesc, flags, 0, t, S.texW, S.texH); }
                };
                // helper threads only with the caller's permission (ommCpuBakeFlags_EnableInternalThreads, omm.h:303: what the reference spends on its OpenMP
                // loops); a thread that cannot be started (std::system_error: thread limit, cgroup pids) leaves its share to this thread -- nothing escapes the C ABI
                const uint32_t n = hc.numPending, ways = (n >= 8192u && (flags & (uint32_t)ommCpuBakeFlags_EnableInternalThreads)) ? 4u : 1u;
                std::vector<std::thread> helpers;
                uint32_t started = 0;
                try {
                    helpers.reserve(ways);
                    for (uint32_t w = 1; w < ways; ++w) { helpers.emplace_back(part, (uint32_t)((uint64_t)n * w / ways), (uint32_t)((uint64_t)n * (w + 1u) / ways)); started = w; }
                } catch (...) {}
                part(0u, (uint32_t)((uint64_t)n / ways));
                for (auto& h : helpers) h.join();
                for (uint32_t w = started + 1u; w < ways; ++w) part((uint32_t)((uint64_t)n * w / ways), (uint32_t)((uint64_t)n * (w + 1u) / ways));   // (shares whose thread did not start)
            }
            if (!HIP_OK(run_setup_fix_pending(S, dScratch, scratchBytes, pend.data(), plv.data(), hc.numPending, tmp.p, stream))) return L.failure("[Failure] - device work-item setup failed");
        }
    }
    if (!HIP_OK(run_setup_items(S, dScratch, scratchBytes, dCounters, dUv, dLevel, dDegen, dTriToItem, dItemIds, stream)))
        return L.failure("[Failure] - device work-item setup failed");
    const int e1 = et.mark();

    // ---- classification parameters ----
    ClassifyParams P; memset(&P, 0, sizeof P);
    P.mipCount = (int)tex.mips.size();
    for (int m = 0; m < P.mipCount; ++m) {
        DevMip& dm = P.mips[m]; const TexMip& tmip = tex.mips[m];
        dm.texels = tmip.texels; dm.sat = tmip.sat; dm.w = tmip.w; dm.h = tmip.h;
        dm.log2w = (int)ctz32((uint32_t)tmip.w); dm.log2h = (int)ctz32((uint32_t)tmip.h);
        dm.pow2 = is_pow2(tmip.w) && is_pow2(tmip.h);
        dm.fw = (float)tmip.w; dm.fh = (float)tmip.h; dm.rw = 1.f / (float)tmip.w; dm.rh = 1.f / (float)tmip.h;
    }
    P.pow2Dispatch = P.mips[0].pow2;
    P.texIsFp32 = tex.format == ommCpuTextureFormat_FP32;
    P.addrMode = d.runtimeSamplerDesc.addressingMode;
    P.filterLinear = d.runtimeSamplerDesc.filter == ommTextureFilterMode_Linear;
    P.format = storeBits; P.promotion = d.unknownStatePromotion; P.stateGT = d.alphaCutoffGreater; P.stateLE = d.alphaCutoffLessEqual;
    P.useCoarse = tex.mips[0].sat != nullptr && P.mipCount == 1 && P.filterLinear;
    P.cutoff = d.alphaCutoff; P.borderAlpha = d.runtimeSamplerDesc.borderAlpha;
    P.wantKnownCount = d.rejectionThreshold > 0.f;
    P.noFine = (flags & (1u << 9)) != 0;   // DisableFineClassification (bake_cpu_impl.cpp:45,822-823)
    P.altKernel = (flags & (1u << 8)) ? ((flags & (1u << 7)) ? 2 : 1) : 0;   // DisableLevelLineIntersection (+ EnableAABBTesting), bake_cpu_impl.cpp:44-45,915-966

    // ---- level-0 hierarchical query per item + compaction of the items that need per-micro-triangle work; ONE sync ----
    launch_triage(P, dUv, dLevel, dDegen, dCounters, maxItems, dMask, dActive, dScratch, stream);
    uint8_t* const hostBlock = arena->host_block();   // layout: [0, 256) counters, [256, 512) tail summary, [1 KiB, ..) the final read-backs
    SetupCounters* const hcDst = hostBlock ? (SetupCounters*)hostBlock : &hc;
    if (!HIP_OK(run_prep(dItemIds, dActive, dLevel, storeBits, maxItems, dCounters, dActiveIds, dStateOfs, dScratch, scratchBytes, stream)) ||
        !HIP_OK(hipMemcpyAsync(hcDst, dCounters, sizeof hc, hipMemcpyDeviceToHost, stream)) || !HIP_OK(hipStreamSynchronize(stream)))
        return L.failure("[Failure] - device work-list compaction failed");
    if (hostBlock) memcpy(&hc, hcDst, sizeof hc);

    uint32_t U = hc.numItems;
    if ((flags & (1u << 5)) && hc.numDisabled != 0) { // bake_cpu_impl.cpp:652-657
        char buf[256];
        snprintf(buf, sizeof buf, "[Info] - The workload consists of %d unclassifiable triangles, these will be classified as unresolvedTriState = %s.", hc.numDisabled, special_name(d.unresolvedTriState));
        L.msg(ommMessageSeverity_Info, buf);
    }
    // ---- ValidateWorkloadSize (bake_cpu_impl.cpp:662-713) ----
    if (checkWorkload) {
        if (d.maxWorkloadSize != 0xFFFFFFFFFFFFFFFFull && hc.workload > d.maxWorkloadSize) return ommResult_WORKLOAD_TOO_BIG;
        if ((flags & (1u << 5)) && hc.workload > (1ull << 27)) {
            char buf[256];
            snprintf(buf, sizeof buf, "[Perf Warning] - The workload consists of %lld work items (number of texels to classify), which corresponds to roughly %lld 1024x1024 textures."
                     " This is unusually large and may result in long bake times.", (long long)hc.workload, (long long)(hc.workload >> 20));
            L.msg(ommMessageSeverity_PerfWarning, buf);
        }
    }
    if ((flags & (1u << 7)) && !(flags & (1u << 8)))   // bake_cpu_impl.cpp:718-719 (ResampleCoarse is the first to look, behind the workload validation)
        return L.invalid("[Invalid Arg] - EnableAABBTesting can't be used without also setting DisableLevelLineIntersection");
    // rank ranges of the per-level active lists (single GPU: every range is the whole level group)
    ShardBounds bounds; memset(&bounds, 0, sizeof bounds);
    bounds.rank = sh ? sh->rank : 0; bounds.world = sh ? sh->world : 1;
    uint32_t lvlFirst[kNumLevels], lvlCount[kNumLevels];
    for (int l = 0; l < kNumLevels; ++l) {
        const uint64_t a = hc.activeStart[l], cnt = hc.activeStart[l + 1] - hc.activeStart[l];
        for (uint32_t r = 0; r <= bounds.world; ++r) bounds.b[l][r] = (uint32_t)(a + cnt * r / bounds.world);
        lvlFirst[l] = bounds.b[l][bounds.rank]; lvlCount[l] = bounds.b[l][bounds.rank + 1] - bounds.b[l][bounds.rank];
    }
    // packed states of the active items + the queue of open tiles (48-byte records, bake_kernels.hip) + its 4 control words
    const size_t stateBytes = pad256(hc.stateBytes ? (size_t)hc.stateBytes : 256), ctlBytes = pad256(sizeof(uint32_t) * kClassifyCtlWords);
    double microAll = 0; for (int l = 0; l < kNumLevels; ++l) microAll += (double)hc.levelCount[l] * (double)(1ull << (2 * l));
    // ---- streamed result (ommCpuBake)?  Worth it when the packed states are large enough for the copy to matter ----
    uint32_t streamChunks = 0; const uint32_t numActiveAll = hc.activeStart[kNumLevels];
    uint8_t* hostArray = nullptr; unsigned long long* hCursor = nullptr; bool hostPinned = false;
    if (so && numActiveAll && !(flags & (1u << 1)) && !P.altKernel && storeBits == bits) {   // (with special indices disabled every uniform item is a block too: the plain path handles that)
        uint32_t k = so->chunksWanted;
        if (!so->forced) {
            // >= 64 MiB of packed states: one range per 32 MiB, at most 32 (round 3, classification-bound, at 1.27 GB: 8 / 16 / 24 / 32 ranges = 38.4 / 36.7 / 36.2 / 36.2 ms)
            k = hc.stateBytes >= (64ull << 20) ? (uint32_t)(hc.stateBytes >> 25) : 0u; if (k > kMaxStreamRanges) k = kMaxStreamRanges;   // (round 4, copy-bound: 12 / 16 / 24 / 32 ranges = 32.4 / 31.8 / 31.0 / 30.9 ms)
            // ... and only when the copy is worth hiding.  Streaming costs the classification about a quarter of its time (5 instead of 6 workgroups per CU,
            // the placement kernels next to it) and saves at most the copy (~57 GB/s over PCIe).  The classification time is estimated from the two
            // quantities that drive it: micro-triangles (sub-texel ones, mostly culled: 2.5e-10 ms each -- 27 ms for 6.5e10, 128 ms for 6.0e11 measured) and
            // texels under the triangles' boxes (micro-triangles of several texels walk them: 4e-9 ms each -- 55 ms for 1.4e10 measured on asset-sized
            // cards, where streaming the 0.28 GB result made the bake 10 ms SLOWER).
            const double classifyMs = 2.5e-10 * microAll + 4e-9 * (double)hc.workload, copyMs = (double)hc.stateBytes / 57e6;
            if (copyMs <= 0.25 * classifyMs) k = 0;
            // ... and, when the compressed transfer is available too (round 5: the codec stream over the link, expanded by the helper threads AFTER the bake), only
            // when streaming wins.  Both are priced from the same two quantities, with this round's rates (1.35e-10 ms per micro-triangle: 8.8 ms for 6.5e10, 88 for
            // 6.0e11) and 65 % of the packed states as the result (what is left after promotion and dedup at the metric configuration):
            //     streamed    6 + max(1.15 classification + 2.7 (preview), result / 57 GB/s)       c2: 28.3 (29.4 measured)   configs[4]: 110 (was 118 at 95 ms)
            //     compressed  3.3 + classification + result / 150 GB/s + codec (0.63 ms per GB)       c2: 18.4 (17 - 20)         configs[4]: 117 (124)
            // -- a long classification hides the whole copy, a short one is better off with every workgroup on the chip and the expansion behind it.
            if (k && so->compressedAvailable) {
                const double cMs = 1.35e-10 * microAll + 4e-9 * (double)hc.workload, resultBytes = 0.65 * (double)hc.stateBytes;
                const double streamedMs = 6.0 + std::max(1.15 * cMs + 2.7, resultBytes / 57e6), compressedMs = 3.3 + cMs + resultBytes / 150e6 + resultBytes * 0.63e-9;
                if (compressedMs <= streamedMs) k = 0;
            }
        }
        if (k > kMaxStreamRanges) k = kMaxStreamRanges;
        if (k && so->set->pinned.reserve(4096) && (hostArray = so->alloc(so->allocUser, hc.stateBytes, &hostPinned)) != nullptr) { streamChunks = k; hCursor = (unsigned long long*)so->set->pinned.base; }
    }
    const size_t queueBytes = pad256((size_t)classify_queue_records(lvlCount, streamChunks > 1) * kTileRecordBytes + 16);
    // ---- deferred generic pass?  (bake_kernels.hip: classify_generic)  For bakes whose cost is in the texels under their micro-triangles, not in their number:
    // the same two terms as above.  Not for streamed bakes (a range must be complete when its queue sections are); a sharded rank defers the walks of
    // its share like a single GPU does (asset-sized cards, 8 ranks: 15.0 -> 12 ms per step), except when the ranks merge their states by summation.
    uint64_t genericCapacity = 0;
    {
        const uint64_t mode = baker.knob(ommxBakerKnob_GenericPass);
        const bool wanted = mode == 2 || (mode == 0 && 4e-9 * (double)hc.workload > 2.5e-10 * microAll);
        if (wanted && !streamChunks && !(sh && sh->mergeStates) && !ht && numActiveAll) {
            genericCapacity = hc.stateBytes * 8ull / (uint64_t)storeBits;            // every micro-triangle of every active item ...
            if (genericCapacity > (256ull << 20)) genericCapacity = 256ull << 20;   // ... at most 2 GB of entries (a tile that finds no room walks its micro-triangles itself)
        }
    }
    const size_t genericBytes = genericCapacity ? pad256((size_t)genericCapacity * 8) + 256 : 0;
    if (!statesArena->reserve(stateBytes + queueBytes + ctlBytes + (streamChunks ? stateBytes : 0) + genericBytes)) return L.failure("[Failure] - out of device memory for the packed micro-triangle states");
    uint8_t* dStates = statesArena->base;
    void* dTileQueue = statesArena->base + stateBytes; uint32_t* dQueueCtl = (uint32_t*)(statesArena->base + stateBytes + queueBytes);
    uint8_t* dStage = streamChunks ? statesArena->base + stateBytes + queueBytes + ctlBytes : nullptr;
    uint8_t* dGeneric = genericCapacity ? statesArena->base + stateBytes + queueBytes + ctlBytes + (streamChunks ? stateBytes : 0) : nullptr;   // count word (256 B), then the entries
    const int e1b = et.mark();

    // ---- ResampleCoarse + ResampleFine (bake_cpu_impl.cpp:715-1029) on the active items ----
    ItemArrays A; A.uv = dUv; A.degenerate = dDegen; A.stateOfs = dStateOfs; A.states = dStates; A.stateMask = dMask; A.knownCount = dKnown; A.fineCount = dFine;
    // (dFine was zeroed by the first set-up launch, with the known counts)
    if (sh && sh->mergeStates && !HIP_OK(hipMemsetAsync(dStates, 0, stateBytes, stream))) return L.failure("[Failure] - device memset failed");
    if (sh && sh->world > 1) { // even out the per-rank cost: interleave every level's active list (tail_kernels.hip: shard_interleave)
        // the permuted copy goes through the (idle) setup / tail scratch block: no allocation, no synchronisation, stream ordered
        uint32_t* tmp = (uint32_t*)dScratch;
        bool okp = (size_t)hc.activeStart[kNumLevels] * 4 <= scratchBytes;
        for (int l = 0; l < kNumLevels && okp; ++l) {
            const uint32_t a = hc.activeStart[l], cnt = hc.activeStart[l + 1] - hc.activeStart[l];
            if (cnt < 3) continue;
            uint32_t stride = (uint32_t)((double)cnt * 0.6180339887498949); if (stride < 1) stride = 1;
            auto gcd = [](uint32_t x, uint32_t y) { while (y) { const uint32_t t = x % y; x = y; y = t; } return x; };
            while (gcd(stride, cnt) != 1) ++stride;   // (cnt - 1 is always coprime: terminates)
            launch_shard_interleave(dActiveIds + a, tmp + a, cnt, stride % cnt, stream);
            okp = HIP_OK(hipMemcpyAsync(dActiveIds + a, tmp + a, (size_t)cnt * 4, hipMemcpyDeviceToDevice, stream));
        }
        if (!okp) return L.failure("[Failure] - shard permutation failed");
    }
    // streamed: the active lists in the order of the final result, events behind the classification launches, the cursor published to a pinned word after each
    struct EventList { hipEvent_t ev[2 * kMaxClassifyChunks + 3]; uint32_t n = 0; ~EventList() { for (uint32_t k = 0; k < n; ++k) (void)hipEventDestroy(ev[k]); } } chunkEvents;   // placement done [K + 1] | fences [K + 2]
    StreamCtx sc; ClassifyChunks cc; memset(&cc, 0, sizeof cc); cc.count = 1;
    MarkCtx mk; mk.et = &et; mk.mark = -1; mk.markGeneric = -1;
    const bool noDedup = (flags & (1u << 3)) != 0;
    int pv0 = -1, pv1 = -1;   // HIP event marks around the preview of a streamed bake
    if (streamChunks) {
        bool oks = HIP_OK(hipMemsetAsync(dPlaced, 0xFF, (size_t)maxItems * 8, stream)) && HIP_OK(hipMemsetAsync(dCursor, 0, 8, stream)) && HIP_OK(hipMemsetAsync(dStreamCtl, 0, sizeof(uint32_t) * kStreamCtlWords, stream));
        oks = oks && HIP_OK(run_stream_begin(dActiveIds, numActiveAll, dUv, dLevel, dScratch, scratchBytes, stream));
        for (uint32_t k = 0; oks && k < 2u * streamChunks + 3u; ++k) { oks = HIP_OK(hipEventCreateWithFlags(&chunkEvents.ev[k], hipEventDisableTiming)); chunkEvents.n += oks ? 1u : 0u; }
        if (!oks) return L.failure("[Failure] - could not set up the streamed result");
        sc.stream = stream; sc.place = so->placeStream; sc.fences = chunkEvents.ev + streamChunks + 1u; sc.activeIds = dActiveIds; sc.numActive = numActiveAll; sc.scratch = dScratch; sc.scratchBytes = scratchBytes; sc.digests = dDigests;
        sc.cursor = dCursor; sc.stage = dStage; sc.placed = dPlaced; sc.ctl = dStreamCtl; sc.hostCursor = hCursor; sc.events = chunkEvents.ev; sc.numEvents = streamChunks + 1u; sc.recorded = 0; sc.ok = true;
        memset(&sc.proto, 0, sizeof sc.proto);
        sc.proto.stateMask = dMask; sc.proto.knownCount = dKnown; sc.proto.digests = dDigests; sc.proto.states = dStates; sc.proto.stateOfs = dStateOfs;
        sc.proto.rejectionThreshold = d.rejectionThreshold; sc.proto.bits = bits; sc.proto.disableDedup = noDedup ? 1 : 0;
        cc.count = streamChunks; cc.after = stream_hook; cc.mark = stream_mark_hook; cc.user = &sc; cc.early = nullptr; sc.queueCtl = dQueueCtl;
        sc.paired = streamChunks > 1; sc.earlyList = nullptr; sc.earlyCapacity = 0; sc.itemLevel = dLevel;
        sc.waitSeconds = 4.0 + 100.0 * 1e-3 * (2.5e-10 * microAll + 4e-9 * (double)hc.workload);   // (100 x the estimate of the whole classification, never below 4 s)
        // preview (tail_kernels.hip): level-5 classification of the items of level >= 6 into buffers of its own; items that share their preview are classified early
        const uint32_t first6 = hc.activeStart[6], count6 = numActiveAll - hc.activeStart[6];
        if (count6 && streamChunks > 1) {
            pv0 = et.mark();
            ClassifyParams P2 = P; P2.format = 2; P2.promotion = 1; P2.wantKnownCount = 0; P2.noFine = 0;
            ItemArrays A2 = A; A2.uv = dUv2; A2.stateOfs = dOfs2; A2.states = dStates2; A2.stateMask = dMask2; A2.fineCount = dFine2;   // (its level-line statistic goes nowhere)
            bool okp = HIP_OK(hipMemsetAsync(dEarly, 0, maxItems, stream)) && HIP_OK(hipMemsetAsync(dMask2, 0, (size_t)maxItems * 4, stream));
            launch_stream_preview_prepare(dActiveIds + first6, count6, dUv, P.mips[0].fw, P.mips[0].fh, dUv2, dOfs2, dEarly, stream);
            {   // the preview items as ONE level-5 class of the ordinary classification: a 1024-tile each, tile triage (most previews are settled by one SAT
                // query), LDS window and the single-texel pass for the rest -- through the bake's own tile queue, which is free until the real launch
                static_assert(kPreviewLevel == 5, "the preview is the 1024-tile class");
                uint32_t first2[kNumLevels], count2[kNumLevels];
                for (int l = 0; l < kNumLevels; ++l) { first2[l] = 0; count2[l] = 0; }
                first2[kPreviewLevel] = first6; count2[kPreviewLevel] = count6;
                okp = okp && HIP_OK(launch_classify(P2, A2, dActiveIds, first2, count2, dTileQueue, dQueueCtl, device_cu_count(), stream, nullptr));
            }
            ClassifyPlan plan; classify_plan(lvlFirst, lvlCount, streamChunks, &plan);   // (the ranges launch_classify will cut)
            okp = okp && HIP_OK(run_stream_preview_flags(dActiveIds + first6, count6, first6, numActiveAll, dStates2, dLevel, dEarly, dStreamCtl, dScratch, scratchBytes, dEarlyLead, dEarlyList, plan, stream));
            if (!okp) return L.failure("[Failure] - could not set up the streamed result");
            cc.early = dEarly; cc.earlyLead = dEarlyLead; cc.earlyStage = dStage; sc.proto.early = dEarly; sc.earlyList = dEarlyList; sc.earlyCapacity = count6;
            pv1 = et.mark();
        }
    } else { cc.mark = mark_hook; cc.user = &mk; cc.early = nullptr; }   // (HIP event in front of the persistent launch of the levels >= 6)
    if (dGeneric) {
        if (!HIP_OK(hipMemsetAsync(dGeneric, 0, 256, stream))) return L.failure("[Failure] - device memset failed");
        cc.generic.count = (unsigned long long*)dGeneric; cc.generic.entries = (uint2*)(dGeneric + 256); cc.generic.capacity = (uint32_t)genericCapacity;
        cc.markGeneric = mark_generic_hook;   // (cc.user is the MarkCtx: a deferred pass and a streamed result exclude each other)
    }
    if (!HIP_OK(launch_classify(P, A, dActiveIds, lvlFirst, lvlCount, dTileQueue, dQueueCtl, device_cu_count(), stream, &cc))) return L.failure("[Failure] - kernel launch failed");
    if (streamChunks && !sc.ok) return L.failure("[Failure] - kernel launch failed");
    const int e2 = et.mark();
    if (ht) { // bring the per-micro-triangle states to the host for the serial tail (host_tail.cpp)
        const ommResult gr = gather_host_items(L, stream, U, T, hc, dUv, dLevel, dActive, dMask, dStateOfs, dStates, dTriToItem, bits, ht->items);
        tm.setupMs = et.ms(e0, e1); tm.triageMs = et.ms(e1, e1b); tm.classifyMs = et.ms(e1b, e2); tm.uniqueItems = U;
        return gr;
    }
    // ---- CalcDigest (bake_cpu_impl.cpp:1038-1040): active items here, uniform ones from the table in the tail ----
    if (!noDedup && !streamChunks)   // (a streamed bake computed them range by range)
        launch_digest_levels(dStates, dStateOfs, dActiveIds, lvlFirst, lvlCount, (uint32_t)storeBits, dDigests, stream);
    if (!HIP_OK(hipGetLastError())) return L.failure("[Failure] - kernel launch failed");
    const int e3 = et.mark();
    // ---- streamed result: everything is enqueued; follow the classification launches and send what each one placed ----
    uint64_t sent = 0;
    SdmaCopier localSdma;   // (ommCpuBake: StreamOut's; its destructor waits for copies in flight: error paths included)
    SdmaCopier& sdma = so ? so->sdma : localSdma;
    if (streamChunks) {
        const bool useSdma = hostPinned && sdma.open(so->device);
        for (uint32_t k = 0; k < sc.recorded; ++k) {
            if (!HIP_OK(hipEventSynchronize(chunkEvents.ev[k]))) return L.failure("[Failure] - the classification failed");
            if (k + 1u == sc.recorded) so->classifyEndMs = now_ms();
            if (k < 32u) tm.streamRangeReadyMs[k] = (float)(now_ms() - coreT0);
            const uint64_t cur = *(volatile unsigned long long*)(hCursor + k);
            if (cur > sent) {
                // (a copy the DMA engine refuses goes through the HIP runtime instead of failing the bake)
                const bool okc = cur <= hc.stateBytes && ((useSdma && sdma.copy_to_host(hostArray + sent, dStage + sent, (size_t)(cur - sent)))
                                                          || HIP_OK(hipMemcpyAsync(hostArray + sent, dStage + sent, (size_t)(cur - sent), hipMemcpyDeviceToHost, so->copyStream)));
                if (!okc) return L.failure("[Failure] - device to host transfer of the bake result failed");
                if (sent == 0) tm.streamFirstCopyMs = (float)(now_ms() - coreT0);
                tm.streamLastCopyMs = (float)(now_ms() - coreT0);
                sent = cur;
            }
        }
    }
    // ---- promote / dedup / sort / offsets on the device ----
    TailInputs ti; memset(&ti, 0, sizeof ti);
    ti.numItems = U; ti.numTris = T; ti.uv = dUv; ti.level = dLevel; ti.stateMask = dMask; ti.knownCount = dKnown; ti.digests = dDigests;
    ti.uniformDigest = dUniformDigest; ti.triToItem = dTriToItem; ti.format = bits;
    // every work item outside the active lists is uniform, and uniform items of one level and state share a digest: at most 13 x 4 of those
    ti.maxDistinctDigests = hc.activeStart[kNumLevels] + 64u;
    ti.disableSpecial = (flags & (1u << 1)) != 0; ti.disableDedup = (flags & (1u << 3)) != 0;
    ti.rejectionThreshold = d.rejectionThreshold; ti.unresolved = (int32_t)d.unresolvedTriState; ti.errorFlag = dErr;
    TailOutputs to; memset(&to, 0, sizeof to);
    to.special = dSpecial; to.rep = dRep; to.order = dOrder; to.dstOfs = dDstOfs; to.sizes = dSizes; to.itemValue = dItemValue;
    to.indexBuffer = dIndex; to.arrayHist = dArrayHist; to.indexHist = dIndexHist;
    if (sh) { // sharded bake: hand the per-item metadata of this rank's share to the caller and stop here (ommxShardedBegin)
        const uint32_t numActive = hc.activeStart[kNumLevels];
        sh->bounds = bounds; sh->ti = ti; sh->to = to; sh->hc = hc; sh->dStates = dStates; sh->dActive = dActive; sh->dLevel = dLevel; sh->dScratch = dScratch;
        sh->dStateOfs = dStateOfs; sh->dMask = dMask; sh->dActiveIds = dActiveIds; sh->dIndex = dIndex; sh->dArrayHist = dArrayHist; sh->dIndexHist = dIndexHist; sh->dTriArea = dTriArea;
        sh->scratchBytes = scratchBytes; sh->flags = flags; sh->T = T; sh->bits = bits;
        launch_shard_pack_meta(bounds, dActiveIds, numActive, dMask, dKnown, dDigests, sh->dMeta, stream);
        if (!sh->asyncBegin && !HIP_OK(hipStreamSynchronize(stream))) return L.failure("[Failure] - sharded classification failed");   // (the one-call RCCL path stays on the stream)
        sh->ev[0] = e0; sh->ev[1] = e1; sh->ev[2] = e1b; sh->ev[3] = e2; sh->ev[4] = e3;   // (read in Finish, when the events are complete)
        tm.uniqueItems = U; tm.activeItems = numActive; tm.stateBytes = hc.stateBytes;
        for (int l = 0; l < kNumLevels; ++l) tm.microTriangles += (uint64_t)hc.levelCount[l] << (2 * l);
        return ommResult_SUCCESS;
    }
    // the index format depends on the triangle count alone: the tail's index kernel writes the narrowed buffer next to the int32 one (bake_cpu_impl.cpp:1872-1902)
    const bool allow8 = (flags & (1u << 6)) != 0, force32 = (flags & (1u << 2)) != 0;
    int idxBytes = 4; R.indexFormat = ommIndexFormat_UINT_32;
    if (allow8 && T <= 127 && !force32) { idxBytes = 1; R.indexFormat = ommIndexFormat_UINT_8; }
    else if (T <= 32767 && !force32) { idxBytes = 2; R.indexFormat = ommIndexFormat_UINT_16; }
    R.index = R.dev_alloc((size_t)(T ? T : 1) * 4); // the reference narrows in place inside an int32 vector (:1882-1900)
    if (!R.index) return L.failure("[Failure] - could not allocate the device result");
    to.narrowIndex = R.index; to.narrowBytes = idxBytes;
    TailCounts counts;
    if (!HIP_OK(run_tail(ti, to, dScratch, scratchBytes, &counts, stream, hostBlock ? hostBlock + 256 : nullptr))) return L.failure("[Failure] - device tail failed");
    if (counts.arrayDataSize > 0xFFFFFFFFull) return ommResult_FAILURE; // bake_cpu_impl.cpp:1774-1775
    const int e4 = et.mark();

    // ---- Serialize (bake_cpu_impl.cpp:1756-1920): gather the surviving blocks into arrayData order, descriptors, index narrowing ----
    const uint32_t E = counts.numOmms;
    R.bits = bits; R.numDescs = E; R.arrayDataSize = E ? counts.arrayDataSize : 0; R.numTris = T;
    ok = true;
    bool streamed = false;
    if (streamChunks) {
        // the blocks are (or will shortly be) at their final offsets in the caller's array, provided the speculative placement equals the exact layout the
        // tail has just produced: compare, and fall back to the ordinary gather + copy otherwise
        uint32_t hctl[4] = { 0u, 0u, 0u, 0u };
        launch_stream_verify(dOrder, dDstOfs, E, dPlaced, dStreamCtl, stream);
        if (!HIP_OK(hipMemcpyAsync(hctl, dStreamCtl, sizeof hctl, hipMemcpyDeviceToHost, stream)) || !HIP_OK(hipStreamSynchronize(stream)))
            return L.failure("[Failure] - could not verify the streamed result");
        streamed = hctl[2] == 0u && sent == R.arrayDataSize;
        if (!streamed) {
            (void)so->finish();   // (the caller is about to overwrite the host array with the ordinary copy: nothing of the discarded stream may still be landing in it)
            char buf[256];
            snprintf(buf, sizeof buf, "[Perf Warning] - the streamed result was discarded (%u blocks placed / %u expected, %llu bytes / %llu expected, duplicate owned by a later range: %u): "
                     "falling back to one copy after the bake", hctl[0], E, (unsigned long long)sent, (unsigned long long)R.arrayDataSize, hctl[1]);
            L.msg(ommMessageSeverity_PerfWarning, buf);
        }
        so->chunks = streamChunks; so->streamedBytes = sent; so->used = streamed; so->fellBack = !streamed;
        tm.streamChunks = streamed ? streamChunks : 0u; tm.streamedBytes = sent; tm.streamEarlyItems = hctl[3];
    }
    if (E) {
        R.descs = (ommCpuOpacityMicromapDesc*)R.dev_alloc(sizeof(ommCpuOpacityMicromapDesc) * (size_t)E);
        if (!streamed) R.arrayData = (uint8_t*)R.dev_alloc((size_t)counts.arrayDataSize);
        ok = R.descs != nullptr && (streamed || R.arrayData != nullptr);
        if (ok) {
            if (!streamed) {
                uint8_t* unitCodes = nullptr; uint32_t* blockRawCounts = nullptr;
                if (R.gatherCodes && storeBits == bits && counts.smallOmms == 0 && !R.gatherCodes(counts.arrayDataSize, &unitCodes, &blockRawCounts)) { unitCodes = nullptr; blockRawCounts = nullptr; }
                launch_gather_omms(dStates, dStateOfs, dActive, dMask, dLevel, bits, storeBits, dOrder, dDstOfs, dSizes, E, R.arrayData, stream, unitCodes, blockRawCounts, R.descs);
            } else launch_write_descs(dOrder, dDstOfs, dLevel, bits, E, R.descs, stream);   // (the gather writes the descriptors of its OMMs itself)
        }
    }
    // the two histograms, the consistency word and the striped statistic counters were taken from the arena back to back: ONE read-back
    unsigned long long fineCount = 0;
    std::vector<unsigned long long> fineSlots((size_t)kFineSlots * kFineStride, 0ull);
    const size_t spanBytes = (size_t)((const uint8_t*)(dFine + fineSlots.size()) - (const uint8_t*)dArrayHist);
    // (into the arena's pinned block when there is one: three copies queued, ONE wait -- into pageable memory each copy is a wait of its own)
    uint32_t hostCtlLocal[kClassifyCtlWords];
#ifdef OMMX_GD_STATS
    unsigned long long genericLocal[20] = { 0 };
#else
    unsigned long long genericLocal[3] = { 0, 0, 0 };
#endif   // reservations (incl. null padding), the pass's cursor, micro-triangles it classified
    const size_t spanAt = 1024, ctlAt = spanAt + pad256(spanBytes), genAt = ctlAt + sizeof hostCtlLocal;
    const bool pinnedBack = hostBlock && genAt + sizeof genericLocal <= kHostBlockBytes;
    std::vector<uint8_t> spanLocal(pinnedBack ? 0 : spanBytes);
    uint8_t* const span = pinnedBack ? hostBlock + spanAt : spanLocal.data();
    uint32_t* const hostCtl = pinnedBack ? (uint32_t*)(hostBlock + ctlAt) : hostCtlLocal;
    unsigned long long* const genericWords = pinnedBack ? (unsigned long long*)(hostBlock + genAt) : genericLocal;
    memset(hostCtl, 0, sizeof hostCtlLocal); memset(genericWords, 0, sizeof genericLocal);
    ok = ok && HIP_OK(hipMemcpyAsync(span, dArrayHist, spanBytes, hipMemcpyDeviceToHost, stream));
    uint32_t queueTails[2] = { 0u, 0u };   // open tiles of the two tile sizes (statistics)
    if (hc.activeStart[kNumLevels]) ok = ok && HIP_OK(hipMemcpyAsync(hostCtl, dQueueCtl, sizeof hostCtlLocal, hipMemcpyDeviceToHost, stream));
    if (dGeneric) ok = ok && HIP_OK(hipMemcpyAsync(genericWords, dGeneric, sizeof genericLocal, hipMemcpyDeviceToHost, stream));
    const int e5 = et.mark();
    ok = ok && HIP_OK(hipStreamSynchronize(stream));
    // (a streamed result may still be on its way to the host: the caller queues its small read-backs first and then waits, StreamOut::finish)
    if (!ok) return L.failure("[Failure] - could not materialise the bake result on the device");
    memcpy(R.hist, span, sizeof(uint32_t) * kNumLevels);
    memcpy(R.hist + kNumLevels, span + ((const uint8_t*)dIndexHist - (const uint8_t*)dArrayHist), sizeof(uint32_t) * kNumLevels);
    memcpy(fineSlots.data(), span + ((const uint8_t*)dFine - (const uint8_t*)dArrayHist), sizeof(unsigned long long) * fineSlots.size());

    tm.uploadMs = 0.f; tm.hostSetupMs = 0.f; tm.setupMs = et.ms(e0, e1); tm.triageMs = et.ms(e1, e1b); tm.classifyMs = et.ms(e1b, e2); tm.digestMs = et.ms(e2, e3);
    tm.streamPreviewMs = pv0 >= 0 ? et.ms(pv0, pv1) : 0.f;
    tm.tailMs = et.ms(e3, e4); tm.gatherMs = et.ms(e4, e5); tm.persistentMs = mk.mark >= 0 ? et.ms(mk.mark, mk.markGeneric >= 0 ? mk.markGeneric : e2) : 0.f;
    tm.genericMs = mk.markGeneric >= 0 ? et.ms(mk.markGeneric, e2) : 0.f; tm.genericMicroTriangles = genericWords[2];
#ifdef OMMX_GD_STATS
    if (dGeneric) { fprintf(stderr, "GDSTATS entries %llu:", genericWords[2]); for (int i = 0; i < 12; ++i) fprintf(stderr, " %llu", genericWords[8 + i]); fprintf(stderr, "\n"); }
#endif
    for (int k = 0; k < kFineSlots; ++k) fineCount += fineSlots[(size_t)k * kFineStride];
    queueTails[1] = hostCtl[kCtl1024 + kSecTails]; for (uint32_t k = 0; k < kMaxClassifyChunks; ++k) queueTails[0] += hostCtl[kSecTails + k];   // (1024-tile queue; sections of the 4096-tile queue)
    tm.openTiles = queueTails[0] + queueTails[1]; tm.openTileMicroTriangles = (uint64_t)queueTails[0] * 4096u + (uint64_t)queueTails[1] * 1024u;
    tm.fineMicroTriangles = fineCount; tm.uniqueItems = U; tm.activeItems = hc.activeStart[kNumLevels]; tm.stateBytes = hc.stateBytes; tm.microTriangles = 0;
    for (int l = 0; l < kNumLevels; ++l) tm.microTriangles += (uint64_t)hc.levelCount[l] << (2 * l);
    {   // classify_tiles launches: one per level below 5, one for level 5, ONE for all levels >= 6 (bake_kernels.hip)
        bool big = false;
        for (int l = 0; l < kNumLevels; ++l) { const bool any = hc.activeStart[l + 1] != hc.activeStart[l]; if (l < 6) tm.classifyLaunches += any; else big = big || any; }
        tm.classifyLaunches += big;
    }
    return ommResult_SUCCESS;
}

inline bool wants_host_tail(const ommCpuBakeInputDesc& d) { return ((uint32_t)d.bakeFlags & ((1u << 4) | (1u << 10))) != 0 || d.maxArrayDataSize != 0xFFFFFFFFu; }

ommResult scope_fences(const Baker& baker, const ommCpuBakeInputDesc& d, bool formatsOnHost, bool hostTailOk = true, bool plainEntry = true)
{
    const Logger& L = baker.log;
    const uint32_t flags = (uint32_t)d.bakeFlags;
    if (wants_host_tail(d) && !hostTailOk) // the serial reducers run on the host over the merged states: not in the caller-driven four-phase protocol
        { L.msg(ommMessageSeverity_Fatal, "[Not Implemented] - near-duplicate merging / maxArrayDataSize budgets are available through ommCpuBake, ommxBakeDevice and ommxShardedBakeRccl, not through ommxShardedBegin/Tail/Finish"); return ommResult_NOT_IMPLEMENTED; }
    // internal flags (bake_cpu_impl.cpp:43-48): EnableAABBTesting (7) / DisableLevelLineIntersection (8) select the reference's ConservativeBilinearKernel
    // (ClassifyParams::altKernel), DisableFineClassification (9), the brute-force near-duplicate search (10) and EnableEdgeHeuristic (11) are honoured
    // without the fine pass unresolved micro-triangles keep the state UnknownOpaque (3), which the reference ORs into ONE bit of a 2-state block together with
    // its neighbour's (bake_cpu_impl.cpp:1811) while digesting the unpacked value: bake_core keeps such a bake in the 2-bit packing up to the final gather --
    // on the two single-device entry points; not where blocks travel between ranks or to the host tail in their packed form
    if ((flags & (1u << 9)) != 0 && d.format == ommFormat_OC1_2_State && (!plainEntry || wants_host_tail(d)))
        { L.msg(ommMessageSeverity_Fatal, "[Not Implemented] - internal bake flag DisableFineClassification (bit 9) with OC1_2_State is supported by ommCpuBake / ommxBakeDevice without near-duplicate merging or a size budget"); return ommResult_NOT_IMPLEMENTED; }
    if (d.formats) { // the reference sizes its arrays from the global format only (bake_cpu_impl.cpp:1763-1772): mixed formats corrupt its heap
        if (!formatsOnHost) return L.failure("[Failure] - per-triangle formats are not supported on the device-resident entry point");
        for (uint32_t i = 0; i < d.indexCount / 3u; ++i)
            if (d.formats[i] != ommFormat_INVALID && d.formats[i] != d.format)
                return L.failure("[Failure] - per-triangle formats that differ from the global format are not supported");
    }
    return ommResult_SUCCESS;
}

struct DeviceBakeResult {
    Allocator mem; DeviceResult R;
    ommCpuOpacityMicromapUsageCount arrayHist[2 * kNumLevels], indexHist[2 * kNumLevels];
    ommCpuBakeResultDesc desc;
};

// result of the serial host tail -> device-resident result (ommxBakeDevice / ommxShardedBakeRccl with the opt-in lossy reducers)
ommResult upload_host_tail(Baker& b, const HostTailResult& hres, ommIndexFormat ifmt, uint32_t T, int bits, hipStream_t stream, DeviceBakeResult* res)
{
    DeviceResult& R = res->R;
    R.pool = b.devPool; R.bits = bits; R.numTris = T; R.indexFormat = ifmt;
    const uint32_t E = (uint32_t)hres.descs.size();
    R.numDescs = E; R.arrayDataSize = E ? hres.arrayData.size() : 0;
    bool ok = true;
    if (E) {
        R.arrayData = (uint8_t*)R.dev_alloc(hres.arrayData.size()); R.descs = (ommCpuOpacityMicromapDesc*)R.dev_alloc(sizeof(ommCpuOpacityMicromapDesc) * (size_t)E);
        ok = R.arrayData && R.descs && HIP_OK(hipMemcpyAsync(R.arrayData, hres.arrayData.data(), hres.arrayData.size(), hipMemcpyHostToDevice, stream))
          && HIP_OK(hipMemcpyAsync(R.descs, hres.descs.data(), sizeof(ommCpuOpacityMicromapDesc) * (size_t)E, hipMemcpyHostToDevice, stream));
    }
    R.index = R.dev_alloc((size_t)(T ? T : 1) * 4);
    ok = ok && R.index && (!T || HIP_OK(hipMemcpyAsync(R.index, hres.index.data(), (size_t)T * 4, hipMemcpyHostToDevice, stream)));
    ok = ok && HIP_OK(hipStreamSynchronize(stream));
    if (!ok) return b.log.failure("[Failure] - could not materialise the bake result on the device");
    const size_t nAH = hres.arrayHist.size() < 2 * (size_t)kNumLevels ? hres.arrayHist.size() : 2 * (size_t)kNumLevels, nIH = hres.indexHist.size() < 2 * (size_t)kNumLevels ? hres.indexHist.size() : 2 * (size_t)kNumLevels;
    memcpy(res->arrayHist, hres.arrayHist.data(), sizeof(ommCpuOpacityMicromapUsageCount) * nAH);
    memcpy(res->indexHist, hres.indexHist.data(), sizeof(ommCpuOpacityMicromapUsageCount) * nIH);
    res->desc.arrayData = E ? R.arrayData : nullptr; res->desc.arrayDataSize = (uint32_t)R.arrayDataSize;
    res->desc.descArray = E ? R.descs : nullptr; res->desc.descArrayCount = E;
    res->desc.descArrayHistogram = res->arrayHist; res->desc.descArrayHistogramCount = (uint32_t)nAH;
    res->desc.indexBuffer = R.index; res->desc.indexCount = T; res->desc.indexFormat = ifmt;
    res->desc.indexHistogram = res->indexHist; res->desc.indexHistogramCount = (uint32_t)nIH;
    return ommResult_SUCCESS;
}

struct BakeSession { // device working set + streams of one bake in flight
    std::shared_ptr<ArenaPool> pool; std::unique_ptr<ArenaSet> set; DeviceArena* arena; DeviceArena* states; hipStream_t stream = nullptr, commStream = nullptr, placeStream = nullptr;
    explicit BakeSession(Baker& b) : pool(b.arenas), set(pool->acquire()) { arena = &set->tables; states = &set->states; }
    ~BakeSession() {
        // the set goes back to the pool only when nothing on the device can still touch it (the streams stay with the set)
        if (placeStream) (void)hipStreamSynchronize(placeStream);
        if (commStream) (void)hipStreamSynchronize(commStream);
        if (stream) (void)hipStreamSynchronize(stream);
        pool->release(std::move(set));
    }
    bool open() { if (!set->stream && hipStreamCreateWithFlags(&set->stream, hipStreamNonBlocking) != hipSuccess) return false; stream = set->stream; return true; }
    bool open_comm() { if (!set->commStream && hipStreamCreateWithFlags(&set->commStream, hipStreamNonBlocking) != hipSuccess) return false; commStream = set->commStream; return true; }
    // high priority: its (small) kernels are dispatched ahead of the next persistent classification launch instead of behind it
    bool open_place() {
        if (!set->placeStream) {
            int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (hipStreamCreateWithPriority(&set->placeStream, hipStreamNonBlocking, hi) != hipSuccess) return false;
        }
        placeStream = set->placeStream; return true;
    }
};

// largest vertex index of the caller's index buffer (3 M entries at the metric configuration: vectorised where the CPU has AVX2, 0.2 instead of 1.3 ms)
template <class T> inline uint32_t max_of(const T* p, size_t n) { uint32_t m = 0; for (size_t i = 0; i < n; ++i) m = p[i] > m ? p[i] : m; return m; }
__attribute__((target("avx2"))) uint32_t max_of_u32_avx2(const uint32_t* p, size_t n) { uint32_t m = 0; for (size_t i = 0; i < n; ++i) m = p[i] > m ? p[i] : m; return m; }
__attribute__((target("avx2"))) uint32_t max_of_u16_avx2(const uint16_t* p, size_t n) { uint32_t m = 0; for (size_t i = 0; i < n; ++i) m = p[i] > m ? p[i] : m; return m; }
uint32_t max_index(const void* idx, ommIndexFormat fmt, size_t n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (fmt == ommIndexFormat_UINT_8) return max_of((const uint8_t*)idx, n);
    if (fmt == ommIndexFormat_UINT_16) return avx2 ? max_of_u16_avx2((const uint16_t*)idx, n) : max_of((const uint16_t*)idx, n);
    return avx2 ? max_of_u32_avx2((const uint32_t*)idx, n) : max_of((const uint32_t*)idx, n);
}

ommResult bake_impl_multi(Baker& baker, const ommCpuBakeInputDesc& d, ommCpuBakeResult* out, uint32_t devices);   // (below, next to the sharded bake it is made of)
constexpr uint64_t kCompressedMinBytes = 32ull << 20;   // smaller arrays cross the link as they are (0.6 ms at 57 GB/s)
// ommCpuBake: host arrays in, host arrays out
ommResult bake_impl(Baker& baker, const ommCpuBakeInputDesc& d, ommCpuBakeResult* out)
{
    const Logger& L = baker.log;
    const double t0 = now_ms();
    struct Active { std::atomic<uint32_t>& n; explicit Active(std::atomic<uint32_t>& c) : n(c) { n.fetch_add(1); } ~Active() { n.fetch_sub(1); } } activeGuard(baker.activeBakes);
    const ommResult fr = scope_fences(baker, d, true);
    if (fr != ommResult_SUCCESS) return fr;
    const uint32_t T = d.indexCount / 3u;
    // several devices behind the one call (ommxBakerKnob_Devices): the work items are shared out, every device hands its own blocks to the host
    // (not where blocks cannot travel in their packed form -- the lossy reducers, bit 9 with the 2-state format -- and not with per-triangle formats: one device)
    if (const uint64_t nd = baker.knob(ommxBakerKnob_Devices))
        if (nd >= 2 && !wants_host_tail(d) && !d.formats && !(((uint32_t)d.bakeFlags & (1u << 9)) != 0 && d.format == ommFormat_OC1_2_State)) return bake_impl_multi(baker, d, out, (uint32_t)nd);
    const DeviceScope onBakersDevice(baker.bind_device());
    BakeSession ses(baker);
    if (!ses.open()) return L.failure("[Failure] - no usable HIP device (the MI355X baker has no CPU fallback)");
    hipStream_t stream = ses.stream;

    // ---- upload the caller's triangle data (the C ABI gives no vertex count: it is max(index)+1, as serialize_impl.cpp:60-79) ----
    const size_t idxSize = d.indexFormat == ommIndexFormat_UINT_8 ? 1 : (d.indexFormat == ommIndexFormat_UINT_16 ? 2 : 4);
    const uint32_t maxIndex = max_index(d.indexBuffer, d.indexFormat, 3ull * T);
    const uint32_t stride = d.texCoordStrideInBytes ? d.texCoordStrideInBytes : (d.texCoordFormat == ommTexCoordFormat_UV32_FLOAT ? 8u : 4u);
    const size_t elem = d.texCoordFormat == ommTexCoordFormat_UV32_FLOAT ? 8 : 4;
    const size_t uvBytes = T ? (size_t)stride * maxIndex + elem : 0, idxBytes = idxSize * 3ull * T, lvlBytes = d.subdivisionLevels ? T : 0;
    uint8_t* dRaw = (uint8_t*)baker.devPool->acquire(pad256(uvBytes) + pad256(idxBytes) + pad256(lvlBytes) + 256);
    if (!dRaw) return L.failure("[Failure] - out of device memory for the triangle data");
    struct RawGuard { DevPool* pool; uint8_t* p; ~RawGuard() { pool->release(p); } } rawGuard{ baker.devPool.get(), dRaw };
    EventTimer et(stream);
    const int u0 = et.mark();
    DeviceInputs din; din.texCoords = dRaw; din.indices = dRaw + pad256(uvBytes); din.perTriLevels = lvlBytes ? dRaw + pad256(uvBytes) + pad256(idxBytes) : nullptr;
    bool ok = true;
    if (uvBytes) ok = ok && HIP_OK(hipMemcpyAsync(dRaw, d.texCoords, uvBytes, hipMemcpyHostToDevice, stream));
    if (idxBytes) ok = ok && HIP_OK(hipMemcpyAsync(dRaw + pad256(uvBytes), d.indexBuffer, idxBytes, hipMemcpyHostToDevice, stream));
    if (lvlBytes) ok = ok && HIP_OK(hipMemcpyAsync(dRaw + pad256(uvBytes) + pad256(idxBytes), d.subdivisionLevels, lvlBytes, hipMemcpyHostToDevice, stream));
    if (!ok) return L.failure("[Failure] - host to device transfer failed");
    const int u1 = et.mark();

    DeviceResult R; ommxBakeTimings tm; memset(&tm, 0, sizeof tm);
    // declared after R and the upload guard, so destroyed first: pooled blocks are only handed back once the stream is idle (error paths too)
    struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } syncOnExit{ stream };
    if (wants_host_tail(d)) {
        // near-duplicate merge / size budget: device classification, then the reference's serial tail on the host (host_tail.cpp)
        HostTailRequest ht;
        const ommResult hr = bake_core(baker, d, din, &d, ses.arena, ses.states, stream, et, R, tm, nullptr, &ht);
        if (hr != ommResult_SUCCESS) return hr;
        HostTailResult hres; ommIndexFormat ifmt = ommIndexFormat_UINT_32;
        if (run_host_tail_for(d, T, ht.items, hres, ifmt) != ommResult_SUCCESS) return ommResult_FAILURE;
        BakeResult* res = baker.mem.make<BakeResult>();
        if (!res) return ommResult_FAILURE;
        res->mem = baker.mem;
        const uint32_t E = (uint32_t)hres.descs.size();
        if (E) {
            res->arrayData = baker.mem.allocate(hres.arrayData.size(), 64); res->descs = (ommCpuOpacityMicromapDesc*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapDesc) * (size_t)E, 16);
        }
        res->index = (int32_t*)baker.mem.allocate(sizeof(int32_t) * (size_t)(T ? T : 1), 16);
        res->triArea = (float*)baker.mem.allocate(sizeof(float) * (size_t)(T ? T : 1), 16);
        res->arrayHist = (ommCpuOpacityMicromapUsageCount*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels, 16);
        res->indexHist = (ommCpuOpacityMicromapUsageCount*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels, 16);
        if ((E && (!res->arrayData || !res->descs)) || !res->index || !res->triArea || !res->arrayHist || !res->indexHist)
            { baker.mem.destroy(res); return L.failure("[Failure] - the memory allocator returned null for the bake result"); }
        if (T && (!HIP_OK(hipMemcpyAsync(res->triArea, R.triAreaScratch, sizeof(float) * (size_t)T, hipMemcpyDeviceToHost, stream)) || !HIP_OK(hipStreamSynchronize(stream))))
            { baker.mem.destroy(res); return L.failure("[Failure] - device to host transfer of the bake result failed"); }
        if (E) { memcpy(res->arrayData, hres.arrayData.data(), hres.arrayData.size()); memcpy(res->descs, hres.descs.data(), sizeof(ommCpuOpacityMicromapDesc) * (size_t)E); }
        memcpy(res->index, hres.index.data(), sizeof(int32_t) * (size_t)T);
        memcpy(res->arrayHist, hres.arrayHist.data(), sizeof(ommCpuOpacityMicromapUsageCount) * hres.arrayHist.size());
        memcpy(res->indexHist, hres.indexHist.data(), sizeof(ommCpuOpacityMicromapUsageCount) * hres.indexHist.size());
        res->desc.arrayData = E ? res->arrayData : nullptr; res->desc.arrayDataSize = E ? (uint32_t)hres.arrayData.size() : 0;
        res->desc.descArray = E ? res->descs : nullptr; res->desc.descArrayCount = E;
        res->desc.descArrayHistogram = res->arrayHist; res->desc.descArrayHistogramCount = (uint32_t)hres.arrayHist.size();
        res->desc.indexBuffer = res->index; res->desc.indexCount = T; res->desc.indexFormat = ifmt;
        res->desc.indexHistogram = res->indexHist; res->desc.indexHistogramCount = (uint32_t)hres.indexHist.size();
        tm.totalMs = (float)(now_ms() - t0);
        { std::lock_guard<std::mutex> g(baker.timingsMu); baker.timings = tm; baker.haveTimings = true; }
        *out = (ommCpuBakeResult)res;
        return ommResult_SUCCESS;
    }
    // the finished blocks travel to their final place in the result array while the classification is still running (StreamOut): the array is allocated
    // when the first block is about to leave, from an upper bound of its size (every active work item a block)
    BakeResult* res = baker.mem.make<BakeResult>();
    if (!res) return ommResult_FAILURE;
    res->mem = baker.mem;
    // (error paths: nothing may still be copying into the result array when it is freed)
    struct ResGuard { Baker& b; BakeResult*& r; BakeSession& s; ~ResGuard() { if (r) { if (s.commStream) (void)hipStreamSynchronize(s.commStream); (void)hipStreamSynchronize(s.stream); b.mem.destroy(r); } } } resGuard{ baker, res, ses };
    // Zeroing ahead (round 6, compressed transfer).  The expansion of the result is bound by what twelve threads can store (1.27 GB in 3.6 ms), and a quarter of
    // the metric configuration's 4-KiB codec blocks repeat state 0 = zeros.  While the device bakes the host is idle: the baker's helper threads zero an IDLE block
    // of the result pool (what the previous bake left; never a fresh allocation, never a user allocator's memory) in pieces of 2 MiB, up to the size of the last
    // compressed result; if the block then fits this bake's result it becomes the result, and the expansion leaves zero blocks of complete pieces alone.
    // Whatever happens -- another transfer, a larger result, an error -- the block is either handed out after the threads have stopped, or goes back to the pool.
    struct Prefill {
        std::shared_ptr<WorkerPool> pool; std::shared_ptr<HostPool> host; uint8_t* block = nullptr; size_t cap = 0, bytes = 0, pieces = 0; bool pinned = false, started = false;
        std::unique_ptr<std::atomic<uint8_t>[]> done; std::atomic<bool> cancel{ false };
        void stop() { if (started) { cancel.store(true); pool->wait(); started = false; } }   // (pieces not begun stay unmarked: the expansion writes them like any other)
        uint64_t zeroed() const { uint64_t n = 0; for (size_t j = 0; j < pieces; ++j) if (done[j].load()) n += (j + 1 < pieces ? (size_t)2 << 20 : bytes - (j << 21)); return n; }
        ~Prefill() { stop(); if (block) host->release(block); }
    } prefill;
    struct ArrayAlloc {
        Baker* b; BakeResult* res; uint64_t cap; bool pinned; Prefill* pre; bool fromPrefill;
        static uint8_t* get(void* u, uint64_t bytes, bool* pinned) {
            ArrayAlloc& a = *(ArrayAlloc*)u;
            if (pinned) *pinned = false;
            if (a.res->arrayData && a.cap >= bytes) { if (pinned) *pinned = a.pinned; return (uint8_t*)a.res->arrayData; }
            if (!a.res->arrayData && a.pre && a.pre->block) {   // the block that was zeroed ahead, if it holds this result (its threads have stopped before anybody writes into it)
                a.pre->stop();
                if (a.pre->cap >= bytes) {
                    a.res->arrayData = a.pre->block; a.res->pool = a.pre->host; a.pinned = a.pre->pinned; a.cap = bytes; a.fromPrefill = true; a.pre->block = nullptr;
                    if (pinned) *pinned = a.pinned;
                    return (uint8_t*)a.res->arrayData;
                }
                a.pre->host->release(a.pre->block); a.pre->block = nullptr;
            }
            if (a.res->arrayData) { if (a.res->pool) { a.res->pool->release(a.res->arrayData); a.res->pool.reset(); } else a.b->mem.release(a.res->arrayData); a.res->arrayData = nullptr; a.cap = 0; }
            // (small results are pooled -- pinned -- only while this is the baker's one bake in flight: sixteen callers copying into pinned blocks at once take
            //  turns on the copy engines -- measured on configs[1]: 2 278 -> 1 190 bakes/s at 16 threads --, while one caller gains 0.28 ms per bake)
            if (a.b->mem.alloc == default_alloc && ((size_t)bytes >= HostPool::kMinBytes || (a.b->activeBakes.load() <= 1u && a.b->hostPool->wants((size_t)bytes)))) {
                a.res->arrayData = a.b->hostPool->acquire((size_t)bytes, &a.pinned);
                if (a.res->arrayData) a.res->pool = a.b->hostPool;
            }
            if (!a.res->arrayData) { a.res->arrayData = a.b->mem.allocate((size_t)bytes, 64); a.pinned = false; }
            a.cap = a.res->arrayData ? bytes : 0;
            if (pinned) *pinned = a.pinned;
            return (uint8_t*)a.res->arrayData;
        }
    } arrayAlloc{ &baker, res, 0, false, &prefill, false };
    StreamOut so; so.set = ses.set.get(); so.allocUser = &arrayAlloc; so.alloc = &ArrayAlloc::get;
    if (const uint64_t k = baker.knob(ommxBakerKnob_StreamChunks)) { so.chunksWanted = (uint32_t)k; so.forced = true; }
    // How a large arrayData reaches the caller (ommxBakerKnob_ResultTransfer).  COMPRESSED (round 5): the bake finishes on the device, the array crosses PCIe as
    // a codec stream (tail_kernels.hip: 6 % of its bytes at the metric configuration) and host threads expand it into the caller's array -- 190 - 250 GB/s with
    // 8 - 16 threads on the GPU box (profiles/r05_host_fill_rates.txt) against the 57 GB/s of the link.  It needs the caller's permission to use threads
    // (ommCpuBakeFlags_EnableInternalThreads, omm.h:303) and CPUs to run them on (below six, the STREAMED form -- blocks placed and copied by the DMA engine
    // while the classification runs -- is the faster one: one thread expands 30 GB/s).
    const uint64_t transferKnob = baker.knob(ommxBakerKnob_ResultTransfer);
    // threads: three quarters of the CPUs the process may use, at most 12 -- the probe's rate is flat from 12 threads on, and a process that runs as many
    // busy threads as its cgroup quota allows is throttled for the rest of the scheduler period as soon as anything else (the HIP runtime's threads, the
    // caller's) runs beside them: measured 6 - 19 ms per expansion with 16 threads on 16 CPUs of quota (ommxBakerKnob_ExpandThreads overrides)
    unsigned expandThreads = effective_cpus() * 3u / 4u; expandThreads = expandThreads > 12u ? 12u : (expandThreads < 1u ? 1u : expandThreads);
    if (const uint64_t k = baker.knob(ommxBakerKnob_ExpandThreads)) expandThreads = (unsigned)k;
    const bool wantCompressed = !so.forced && (transferKnob == ommxResultTransfer_Compressed ||
                                               (transferKnob == ommxResultTransfer_Auto && ((uint32_t)d.bakeFlags & (uint32_t)ommCpuBakeFlags_EnableInternalThreads) != 0 && effective_cpus() >= 6u));
    // (automatic choice with threads: both transfers are on offer, bake_core prices them once it knows the size of the bake -- a classification of 100 ms hides the
    //  whole copy of a streamed result, configs[4]; a short one is better off with the compressed transfer behind it, the metric configuration)
    so.compressedAvailable = wantCompressed && transferKnob == ommxResultTransfer_Auto;
    const bool canStream = (!wantCompressed || so.compressedAvailable) && transferKnob != ommxResultTransfer_Plain && ses.open_comm() && ses.open_place();
    so.copyStream = ses.commStream; so.placeStream = ses.placeStream; so.device = baker.bind_device();
    struct CodecOut { uint8_t* dBlock = nullptr; DevPool* pool = nullptr; uint8_t* dComp = nullptr; uint32_t* dSize = nullptr; HostCodecLayout L{}; uint64_t padded = 0, cap = 0; bool on = false;
                      uint8_t* unitCodes = nullptr; uint32_t* blockRawCounts = nullptr; size_t scratchBytes = 0;   // (set when the gather produces the codes)
                      ~CodecOut() { if (dBlock) pool->release(dBlock); } } co;
    // one device block for the codec: size word (256 B) | scan scratch | [unit codes | block counts] | the stream
    auto codec_block = [&](uint64_t arrayDataSize, bool withCodes) -> bool {
        co.padded = (arrayDataSize + 255u) & ~255ull; co.L = host_codec_layout(co.padded);
        co.cap = co.L.offRaw + co.padded / 2u + 16u;   // a stream that does not shrink below half takes the plain copy
        co.scratchBytes = pad256(shard_codec_scratch_bytes(co.padded));
        const size_t codeBytes = withCodes ? pad256((size_t)(co.padded / 16u)) : 0, countBytes = withCodes ? pad256(((size_t)co.L.blocks + 1) * 4) : 0;
        co.pool = baker.devPool.get(); co.dBlock = (uint8_t*)baker.devPool->acquire(256 + co.scratchBytes + codeBytes + countBytes + (size_t)co.cap);
        if (!co.dBlock || co.L.blocks >= 0x7FFFFFFFull) return false;
        co.dSize = (uint32_t*)co.dBlock; co.dComp = co.dBlock + 256 + co.scratchBytes + codeBytes + countBytes;
        if (withCodes) { co.unitCodes = co.dBlock + 256 + co.scratchBytes; co.blockRawCounts = (uint32_t*)(co.unitCodes + codeBytes); }
        return true;
    };
    if (wantCompressed) R.gatherCodes = [&](uint64_t arrayDataSize, uint8_t** unitCodes, uint32_t** blockRawCounts) -> bool {
        if (arrayDataSize < kCompressedMinBytes || (arrayDataSize & 15u) != 0 || !codec_block(arrayDataSize, true)) return false;
        // (the units of the padding behind the array count as zeros; the counts start at zero)
        const uint64_t units = arrayDataSize / 16u, paddedUnits = co.padded / 16u;
        bool okm = HIP_OK(hipMemsetAsync(co.blockRawCounts, 0, ((size_t)co.L.blocks + 1) * 4, stream));
        if (okm && paddedUnits > units) okm = HIP_OK(hipMemsetAsync(co.unitCodes + units, 0, (size_t)(paddedUnits - units), stream));
        if (!okm) { co.unitCodes = nullptr; co.blockRawCounts = nullptr; return false; }
        *unitCodes = co.unitCodes; *blockRawCounts = co.blockRawCounts;
        return true;
    };
    // zeroing ahead: only with the default allocator, the compressed transfer on offer, helper threads, and an idle pool block of the last compressed result's size
    if (wantCompressed && baker.mem.alloc == default_alloc && baker.knob(ommxBakerKnob_RetainMemory) == 0 && baker.knob(ommxBakerKnob_ZeroAhead) == 0) {
        const uint64_t last = baker.lastCompressedArrayBytes.load();
        if (last >= kCompressedMinBytes && (prefill.block = (uint8_t*)baker.hostPool->acquire_idle((size_t)last, &prefill.cap, &prefill.pinned)) != nullptr) {
            prefill.host = baker.hostPool; prefill.pool = baker.worker_pool(expandThreads);
            prefill.bytes = (size_t)last < prefill.cap ? (size_t)last : prefill.cap; prefill.pieces = (prefill.bytes + ((size_t)2 << 20) - 1) >> 21;
            prefill.done.reset(new (std::nothrow) std::atomic<uint8_t>[prefill.pieces]);
            if (prefill.done && ((uintptr_t)prefill.block & 4095u) == 0u) {
                for (size_t j = 0; j < prefill.pieces; ++j) prefill.done[j].store(0);
                if (baker.knob(ommxBakerKnob_HelperAffinity) == 0) (void)prefill.pool->bind_near(prefill.block);
                Prefill* pf = &prefill;
                prefill.started = prefill.pool->start((uint32_t)prefill.pieces, [pf](uint32_t j) {
                    if (pf->cancel.load(std::memory_order_relaxed)) return;
                    const size_t lo = (size_t)j << 21, hi = lo + ((size_t)2 << 20) < pf->bytes ? lo + ((size_t)2 << 20) : pf->bytes;
                    fill_zero_nt(pf->block, lo, hi);
                    pf->done[j].store(1, std::memory_order_release);
                });
            }
        }
    }
    const ommResult br = bake_core(baker, d, din, &d, ses.arena, ses.states, stream, et, R, tm, nullptr, nullptr, canStream ? &so : nullptr);
    R.gatherCodes = nullptr;
    if (br != ommResult_SUCCESS) return br;

    // ---- copy the (rest of the) result out through the user's allocator ----
    const int d0 = et.mark();
    const uint32_t E = R.numDescs;
    // ---- compressed result: codec stream of the finished array (device), one copy of the stream, expansion by the baker's helper threads ----
    std::vector<uint8_t> codecHead;   // header + offsets of the codec stream (1 MB per GB of arrayData)
    const double c0 = now_ms();
    if (E && wantCompressed && !so.used && R.arrayDataSize >= kCompressedMinBytes) {
        // (the device array was taken from the result pool, whose blocks are multiples of 4096 bytes: the codec may read the padding, the host never writes it)
        if (co.unitCodes)   // the gather has left the codes of the units and the raw counts of the blocks: the array is read for its raw units only
            co.on = HIP_OK(run_shard_compress_coded(R.arrayData, co.padded, co.unitCodes, co.blockRawCounts, co.dComp, co.cap, co.dSize, co.dBlock + 256, co.scratchBytes, stream));
        else if (!co.dBlock && codec_block(R.arrayDataSize, false))
            co.on = HIP_OK(run_shard_compress(R.arrayData, co.padded, co.dComp, co.cap, co.dSize, co.dBlock + 256, co.scratchBytes, stream));
        (void)hipGetLastError();
    }
    if (E) {
        ok = ArrayAlloc::get(&arrayAlloc, R.arrayDataSize, nullptr) != nullptr;   // (a streamed bake has it already; every other path asks for the exact size)
        res->descs = (ommCpuOpacityMicromapDesc*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapDesc) * (size_t)E, 16);
        ok = ok && res->descs;
        if (co.on) {   // the stream's header (its length) and the per-block offsets of the raw units: the first piece of the stream, needed before the others can be cut
            codecHead.resize((size_t)co.L.offCodes);
            ok = ok && HIP_OK(hipMemcpyAsync(codecHead.data(), co.dComp, (size_t)co.L.offCodes, hipMemcpyDeviceToHost, stream));
        } else if (!so.used) ok = ok && HIP_OK(hipMemcpyAsync(res->arrayData, R.arrayData, (size_t)R.arrayDataSize, hipMemcpyDeviceToHost, stream));
    }
    res->index = (int32_t*)baker.mem.allocate(sizeof(int32_t) * (size_t)(T ? T : 1), 16);
    res->triArea = (float*)baker.mem.allocate(sizeof(float) * (size_t)(T ? T : 1), 16);
    ok = ok && res->index != nullptr && res->triArea != nullptr;
    const size_t outIdx = R.indexFormat == ommIndexFormat_UINT_8 ? 1 : (R.indexFormat == ommIndexFormat_UINT_16 ? 2 : 4);
    // descriptors, index buffer and triangle areas (8.6 MB at the metric configuration, into the caller's pageable arrays: each copy holds its thread until it is done).
    // With a compressed result they cross the link on the second stream WHILE the helper threads expand the array (one of the expansion's tasks issues them);
    // otherwise here, in front of the wait.
    auto small_copies = [&](hipStream_t s) -> bool {
        bool k = true;
        if (E) k = HIP_OK(hipMemcpyAsync(res->descs, R.descs, sizeof(ommCpuOpacityMicromapDesc) * (size_t)E, hipMemcpyDeviceToHost, s));
        if (k && T) k = HIP_OK(hipMemcpyAsync(res->index, R.index, outIdx * T, hipMemcpyDeviceToHost, s));
        if (k && T) k = HIP_OK(hipMemcpyAsync(res->triArea, R.triAreaScratch, sizeof(float) * (size_t)T, hipMemcpyDeviceToHost, s));
        return k;
    };
    hipEvent_t evResult = nullptr;   // the device result is complete (everything in front of it on the bake's stream)
#ifndef OMMX_DEFER_SMALL
#define OMMX_DEFER_SMALL 1
#endif
    bool deferSmall = OMMX_DEFER_SMALL && ok && co.on && ses.open_comm() && HIP_OK(hipEventCreateWithFlags(&evResult, hipEventDisableTiming));
    if (deferSmall) deferSmall = HIP_OK(hipEventRecord(evResult, stream)) && HIP_OK(hipStreamWaitEvent(ses.commStream, evResult, 0));
    struct EvGuard { hipEvent_t& e; ~EvGuard() { if (e) (void)hipEventDestroy(e); } } evGuard{ evResult };
    if (ok && !deferSmall) ok = small_copies(stream);
    const int d1 = et.mark();
    ok = ok && HIP_OK(hipStreamSynchronize(stream));
    if (so.chunks) { ok = so.finish() && ok; if (so.used) tm.streamTailMs = (float)(so.lastByteMs - so.classifyEndMs); }   // (the last streamed bytes arrive while the small arrays above are read back)
    if (ok && co.on) {
        const double c1 = now_ms();
        uint64_t streamBytes = 0; memcpy(&streamBytes, codecHead.data(), 8);   // (shard_codec_finish: the length the stream has, or would have had)
        const bool fits = streamBytes <= co.cap && streamBytes >= co.L.offRaw && ses.set->pinned.reserve((size_t)streamBytes + 4096);   // (pinned staging of the stream's size)
        uint8_t* hStream = ses.set->pinned.base;
        if (fits) memcpy(hStream, codecHead.data(), codecHead.size());
        if (!fits) {   // noise-like states (the stream would not be below half of the array): the array itself crosses the link
            ok = HIP_OK(hipMemcpyAsync(res->arrayData, R.arrayData, (size_t)R.arrayDataSize, hipMemcpyDeviceToHost, stream)) && (!deferSmall || small_copies(stream)) && HIP_OK(hipStreamSynchronize(stream));
            tm.resultTransfer = ommxResultTransfer_Plain;
        } else {
            // The rest of the stream lands in the working set's pinned block slice by slice -- the codes of a range of codec blocks and the raw units those blocks
            // point at (both contiguous: the offsets are a prefix sum) -- and the helper threads expand slice k while slice k + 1 is on the link.
            constexpr uint32_t kSlices = 12;
            const uint32_t* ofs = (const uint32_t*)(hStream + co.L.offOfs);
            hipEvent_t evs[kSlices]; uint32_t nev = 0;
            for (; nev < kSlices; ++nev) if (!HIP_OK(hipEventCreateWithFlags(&evs[nev], hipEventDisableTiming))) break;
            ok = nev == kSlices;
            uint64_t blockCut[kSlices + 1];
            for (uint32_t k = 0; k <= kSlices; ++k) blockCut[k] = co.L.blocks * k / kSlices;
            for (uint32_t k = 0; ok && k < kSlices; ++k) {
                const uint64_t b0 = blockCut[k], b1 = blockCut[k + 1];
                const uint64_t c0b = co.L.offCodes + b0 * 128u, c1b = b1 == co.L.blocks ? co.L.offRaw : co.L.offCodes + b1 * 128u;
                const uint64_t r0b = co.L.offRaw + 16ull * ofs[b0], r1b = co.L.offRaw + 16ull * ofs[b1];
                ok = r0b <= r1b && r1b <= streamBytes;   // (a corrupt offset table must not turn into a wild copy)
                if (ok && c1b > c0b) ok = HIP_OK(hipMemcpyAsync(hStream + c0b, co.dComp + c0b, (size_t)(c1b - c0b), hipMemcpyDeviceToHost, stream));
                if (ok && r1b > r0b) ok = HIP_OK(hipMemcpyAsync(hStream + r0b, co.dComp + r0b, (size_t)(r1b - r0b), hipMemcpyDeviceToHost, stream));
                ok = ok && HIP_OK(hipEventRecord(evs[k], stream));
            }
            if (ok) {
                const std::shared_ptr<WorkerPool> poolRef = baker.worker_pool(expandThreads); WorkerPool& pool = *poolRef;
                // the helper threads next to the memory they fill: on the two-socket host of the GPU box a process whose threads happened to run on the other socket
                // expanded at half the rate (five process pairs on one lease: 17.0 - 25.0 ms per call without, 17.0 - 19.8 with the binding)
                if (baker.knob(ommxBakerKnob_HelperAffinity) == 0) (void)pool.bind_near(res->arrayData);
                uint8_t* dst = (uint8_t*)res->arrayData; const uint64_t dstBytes = R.arrayDataSize; const HostCodecLayout L = co.L;
                // ONE run over all tasks (2 MiB of the array each, in slice order): a task whose slice is not on the host yet polls the slices' events in order --
                // whichever thread gets there first moves the `arrived` mark -- so the threads never meet at a barrier between slices (12 runs, one per slice,
                // cost ~0.5 ms of joins and idle tails)
                constexpr uint64_t kTaskBlocks = 512;
                const uint64_t numTasks = (co.L.blocks + kTaskBlocks - 1) / kTaskBlocks;
                std::atomic<uint32_t> arrived{ 0 }; std::atomic<bool> failed{ false };
                // (the result array is the block that was zeroed ahead: blocks of zeros in its complete pieces are left alone)
                std::atomic<uint64_t> skippedBytes{ 0 };
                const ZeroedPieces zeroedPieces{ prefill.done.get(), prefill.pieces };
                const ZeroedPieces* const zeroedPtr = arrayAlloc.fromPrefill && prefill.done ? &zeroedPieces : nullptr;
                ok = HIP_OK(hipEventSynchronize(evs[0]));
                if (ok) arrived.store(1);
                const int dev = baker.bind_device();
                const uint32_t extra = deferSmall ? 1u : 0u;   // task 0: the small arrays, on the second stream
                if (ok) pool.run((uint32_t)numTasks + extra, [&](uint32_t t0) {
                    if (extra && t0 == 0u) {
                        const DeviceScope onDev(dev);
                        if (!small_copies(ses.commStream) || !HIP_OK(hipStreamSynchronize(ses.commStream))) failed.store(true);
                        return;
                    }
                    const uint32_t t = t0 - extra;
                    const uint64_t s0 = (uint64_t)t * kTaskBlocks, s1 = s0 + kTaskBlocks < L.blocks ? s0 + kTaskBlocks : L.blocks;
                    uint32_t need = 0; while (need + 1u < kSlices && blockCut[need + 1u] < s1) ++need;   // the last slice this task reads from
                    // (a few polls back to back -- a slice is ~0.1 ms of PCIe time --, then short sleeps: up to twelve threads spinning on hipEventQuery burn the
                    //  process's CPU quota for nothing, and a throttled process expands slower)
                    uint32_t polls = 0;
                    for (uint32_t a; (a = arrived.load(std::memory_order_acquire)) <= need && !failed.load(std::memory_order_relaxed); ) {
                        const hipError_t q = hipEventQuery(evs[a]);
                        if (q == hipSuccess) { uint32_t expect = a; arrived.compare_exchange_strong(expect, a + 1u, std::memory_order_acq_rel); polls = 0; }
                        else if (q != hipErrorNotReady) { (void)hipGetLastError(); failed.store(true); }
                        else { (void)hipGetLastError(); if (++polls < 32u) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20)); }
                    }
                    if (!failed.load(std::memory_order_relaxed)) {
                        uint64_t skippedHere = 0;
                        codec_expand_blocks(dst, dstBytes, hStream, L, s0, s1, zeroedPtr, &skippedHere);
                        if (skippedHere) skippedBytes.fetch_add(skippedHere, std::memory_order_relaxed);
                    }
                });
                ok = ok && !failed.load();
                tm.expandThreads = pool.workers() + 1u;
                tm.expandSkippedBytes = skippedBytes.load(); tm.prefilledBytes = zeroedPtr ? prefill.zeroed() : 0;
            }
            if (!ok) (void)hipStreamSynchronize(stream);   // (nothing may still be landing in the pinned block when the session hands it back)
            for (uint32_t k = 0; k < nev; ++k) (void)hipEventDestroy(evs[k]);
            tm.resultTransfer = ommxResultTransfer_Compressed; tm.compressedBytes = streamBytes;
            if (ok && fits) baker.lastCompressedArrayBytes.store(R.arrayDataSize);   // (what the next bake of this baker zeroes ahead)
        }
        tm.compressMs = (float)(c1 - c0); tm.expandMs = (float)(now_ms() - c1);
    } else if (so.used) tm.resultTransfer = ommxResultTransfer_Streamed;
    else tm.resultTransfer = ommxResultTransfer_Plain;
    if (!ok) { return L.failure("[Failure] - device to host transfer of the bake result failed"); }

    // histograms: format {2-state, 4-state} x level ascending, non-zero entries only (:1833-1850); one global format here
    res->arrayHist = (ommCpuOpacityMicromapUsageCount*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels, 16);
    res->indexHist = (ommCpuOpacityMicromapUsageCount*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels, 16);
    if (!res->arrayHist || !res->indexHist) { return L.failure("[Failure] - the memory allocator returned null for the bake result"); }
    uint32_t nAH = 0, nIH = 0;
    for (uint32_t l = 0; l < (uint32_t)kNumLevels; ++l) {
        if (R.hist[l]) { res->arrayHist[nAH].count = R.hist[l]; res->arrayHist[nAH].subdivisionLevel = (uint16_t)l; res->arrayHist[nAH].format = (uint16_t)R.bits; nAH++; }
        if (R.hist[kNumLevels + l]) { res->indexHist[nIH].count = R.hist[kNumLevels + l]; res->indexHist[nIH].subdivisionLevel = (uint16_t)l; res->indexHist[nIH].format = (uint16_t)R.bits; nIH++; }
    }
    res->desc.arrayData = E ? res->arrayData : nullptr; res->desc.arrayDataSize = E ? (uint32_t)R.arrayDataSize : 0;
    res->desc.descArray = E ? res->descs : nullptr; res->desc.descArrayCount = E;
    res->desc.descArrayHistogram = res->arrayHist; res->desc.descArrayHistogramCount = nAH;
    res->desc.indexBuffer = res->index; res->desc.indexCount = T; res->desc.indexFormat = R.indexFormat;
    res->desc.indexHistogram = res->indexHist; res->desc.indexHistogramCount = nIH;
    tm.uploadMs = et.ms(u0, u1); tm.downloadMs = et.ms(d0, d1); tm.totalMs = (float)(now_ms() - t0);
    { std::lock_guard<std::mutex> g(baker.timingsMu); baker.timings = tm; baker.haveTimings = true; }
    *out = (ommCpuBakeResult)res;
    res = nullptr;   // (handed over: the guard lets go)
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
    // ommBakerType_GPU bakers are creatable and destroyable like in the SDK (support/tests/test_basic.cpp:46-51); every ommGpu* entry
    // point answers NOT_IMPLEMENTED and the ommCpu* ones reject them ("Baker was not created as the right type")
    return guarded(nullptr, [&] {
        Baker* b = mem.make<Baker>();
        if (!b) return ommResult_FAILURE;
        b->mem = mem; b->log.iface = desc->messageInterface; b->type = desc->type;
        *outBaker = (ommBaker)((uintptr_t)b | (desc->type == ommBakerType_CPU ? kCpuBaker : kGpuBaker));
        return ommResult_SUCCESS;
    });
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

namespace {
ommResult create_texture_impl(Baker* b, const ommCpuTextureDesc* desc, ommCpuTexture* outTexture)
{
    const Logger& L = b->log;
    const DeviceScope onBakersDevice(b->bind_device());
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
    t->mem = b->mem; t->log = &b->log; t->format = desc->format; t->flags = desc->flags; t->alphaCutoff = desc->alphaCutoff; t->device = b->bind_device();
    const bool linear = ((uint32_t)desc->flags & (uint32_t)ommCpuTextureFlags_DisableZOrder) != 0;
    const size_t px = desc->format == ommCpuTextureFormat_FP32 ? 4 : 1;
    const bool enableSAT = desc->alphaCutoff >= 0; // texture_impl.cpp:91
    bool ok = true;
    std::vector<uint8_t> staging; std::vector<void*> satScratch;
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
            if (ok) {   // (null stream; the pooled scratch block goes back after the device synchronisation below)
                uint32_t* scratch = (uint32_t*)b->devPool->acquire(sat_scratch_bytes(m.w, m.h));
                ok = scratch != nullptr;
                if (ok) { satScratch.push_back(scratch); launch_sat_build(m.texels, desc->format == ommCpuTextureFormat_FP32, m.sat, scratch, m.w, m.h, desc->alphaCutoff, nullptr); ok = HIP_OK(hipGetLastError()); }
            }
        }
        t->mips.push_back(m);
    }
    if (ok) ok = HIP_OK(hipDeviceSynchronize()); else (void)hipDeviceSynchronize();
    for (void* p : satScratch) b->devPool->release(p);
    if (!ok) { (void)hipGetLastError(); b->mem.destroy(t); return L.failure("[Failure] - could not create the texture on the HIP device (no CPU fallback)"); }
    *outTexture = (ommCpuTexture)((uintptr_t)t | kTexture);
    return ommResult_SUCCESS;
}
} // namespace

OMM_MI355X_API ommResult ommCpuCreateTexture(ommBaker baker, const ommCpuTextureDesc* desc, ommCpuTexture* outTexture)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    if (desc == 0) return b->log.invalid("texture desc was not set");
    if (tag_of(baker) != kCpuBaker) return b->log.invalid("Baker was not created as the right type");
    return guarded(&b->log, [&] { return create_texture_impl(b, desc, outTexture); });
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
    return guarded(&b->log, [&] { return bake_impl(*b, *desc, outBakeResult); });
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

namespace {
// CollectStats (debug_impl.cpp:512-641), in the reference's evaluation order: per-descriptor micro-triangle counts are uint32, the
// per-reference products are taken in 32 bits before they are added to the 64-bit totals (`uint * uint32_t`, :630-633), the areas of
// the triangles that share a descriptor are summed in triangle order and the descriptors are visited in ascending index order
// (std::map, :523,621).  `area` == nullptr (ommDebugGetStats) leaves knownAreaMetric at 0.
ommResult collect_stats(const ommCpuBakeResultDesc* res, const float* area, ommDebugStats* out)
{
    if (res == nullptr || out == nullptr) return ommResult_INVALID_ARGUMENT;
    ommDebugStats st; memset(&st, 0, sizeof st);
    const uint32_t T = res->indexCount;
    float totalArea = 0.f, knownArea = 0.f;
    if (area) for (uint32_t i = 0; i < T; ++i) totalArea += area[i];
    std::vector<uint32_t> refs((size_t)res->descArrayCount + 1, 0);
    std::vector<float> refArea(area ? (size_t)res->descArrayCount + 1 : 0, 0.f);
    for (uint32_t i = 0; i < T; ++i) {
        int32_t v;
        if (res->indexFormat == ommIndexFormat_UINT_8) v = ((const int8_t*)res->indexBuffer)[i];
        else if (res->indexFormat == ommIndexFormat_UINT_16) v = ((const int16_t*)res->indexBuffer)[i];
        else v = ((const int32_t*)res->indexBuffer)[i];
        if (v == ommSpecialIndex_FullyTransparent) { st.totalFullyTransparent++; knownArea += area ? area[i] : 0; }
        else if (v == ommSpecialIndex_FullyOpaque) { st.totalFullyOpaque++; knownArea += area ? area[i] : 0; }
        else if (v == ommSpecialIndex_FullyUnknownTransparent) st.totalFullyUnknownTransparent++;
        else if (v == ommSpecialIndex_FullyUnknownOpaque) st.totalFullyUnknownOpaque++;
        else if (v >= 0 && (uint32_t)v < res->descArrayCount) {
            if (area) { if (refs[(size_t)v] == 0) refArea[(size_t)v] = area[i]; else refArea[(size_t)v] += area[i]; }
            refs[(size_t)v]++;
        }
    }
    for (uint32_t i = 0; i < res->descArrayCount; ++i) {
        if (!refs[i]) continue;
        const ommCpuOpacityMicromapDesc& dd = res->descArray[i];
        const uint8_t* data = (const uint8_t*)res->arrayData + dd.offset;
        const uint32_t nM = 1u << (dd.subdivisionLevel << 1);
        const uint32_t is2 = dd.format == ommFormat_OC1_2_State;
        uint32_t c[4] = { 0, 0, 0, 0 };
        for (uint32_t u = 0; u < nM; ++u) {
            const uint8_t v = data[u >> (2 + is2)];
            c[is2 ? ((v >> (u & 7)) & 1u) : ((v >> ((u << 1) & 7)) & 3u)]++;
        }
        if (area) {
            const uint32_t totalKnown = c[0] + c[1], totalUnknown = c[2] + c[3];
            const float known = (float)totalKnown / (float)(totalKnown + totalUnknown);
            knownArea += known * refArea[i];
        }
        st.totalTransparent += (uint32_t)(refs[i] * c[0]); st.totalOpaque += (uint32_t)(refs[i] * c[1]);
        st.totalUnknownTransparent += (uint32_t)(refs[i] * c[2]); st.totalUnknownOpaque += (uint32_t)(refs[i] * c[3]);
    }
    st.knownAreaMetric = area ? knownArea / totalArea : 0;
    *out = st;
    return ommResult_SUCCESS;
}
const Logger* baker_log(ommBaker baker) { return baker ? &untag<Baker>(baker)->log : nullptr; }
} // namespace

// include/omm.h:1201, bake.cpp:338-357
OMM_MI355X_API ommResult ommDebugGetStats(ommBaker baker, const ommCpuBakeResultDesc* res, ommDebugStats* out)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    if (tag_of(baker) != kCpuBaker && tag_of(baker) != kGpuBaker) return ommResult_INVALID_ARGUMENT;
    return guarded(baker_log(baker), [&] { return collect_stats(res, nullptr, out); });
}

// include/omm.h:1202, bake.cpp:359-386: statistics of a result OBJECT, with the per-triangle areas it carries
OMM_MI355X_API ommResult ommDebugGetStats2(ommBaker baker, ommCpuBakeResult res, ommDebugStats* out)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    if (res == 0) return ommResult_INVALID_ARGUMENT;   // (the reference dereferences a null result here)
    if (tag_of(baker) != kCpuBaker && tag_of(baker) != kGpuBaker) return ommResult_INVALID_ARGUMENT;
    const BakeResult* r = (const BakeResult*)res;
    return guarded(baker_log(baker), [&] { return collect_stats(&r->desc, r->triArea, out); });
}

// include/omm.h:1204, bake.cpp:388-408, debug_impl.cpp:654-670
OMM_MI355X_API ommResult ommDebugSaveBinaryToDisk(ommBaker baker, const ommCpuBlobDesc* data, const char* path)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    if (path == 0) return ommResult_INVALID_ARGUMENT;
    if (tag_of(baker) != kCpuBaker && tag_of(baker) != kGpuBaker) return ommResult_INVALID_ARGUMENT;
    const Logger& L = untag<Baker>(baker)->log;
    FILE* f = data ? fopen(path, "wb") : nullptr;
    bool ok = f != nullptr;
    if (ok && data->size) ok = fwrite(data->data, 1, (size_t)data->size, f) == (size_t)data->size;
    if (f) ok = (fclose(f) == 0) && ok;
    if (!ok) { char buf[512]; snprintf(buf, sizeof buf, "Unable to save file %s", path); L.msg(ommMessageSeverity_Error, buf); return ommResult_INVALID_ARGUMENT; } // log.ErrorArgf
    return ommResult_SUCCESS;
}

// ---- link-compatible stubs of the SDK's GPU baker and PNG dump (include/omm.h:1127-1141,1199; bake.cpp:262-336) ----
// Argument checks follow the reference; past them the answer is NOT_IMPLEMENTED (+ a log line where a baker is at hand).
namespace {
ommResult gpu_baker_not_built(const Logger* L, const char* fn)
{
    if (L) { char buf[256]; snprintf(buf, sizeof buf, "[Not Implemented] - %s: the SDK's D3D12/Vulkan GPU baker is not part of the MI355X build (use ommCpuBake, which runs on the GPU here)", fn); L->msg(ommMessageSeverity_Fatal, buf); }
    return ommResult_NOT_IMPLEMENTED;
}
}
OMM_MI355X_API ommResult ommGpuGetStaticResourceData(ommGpuResourceType, uint8_t*, size_t* outByteSize)
{
    if (outByteSize == nullptr) return ommResult_INVALID_ARGUMENT;
    return gpu_baker_not_built(nullptr, "ommGpuGetStaticResourceData");
}
OMM_MI355X_API ommResult ommGpuCreatePipeline(ommBaker baker, const ommGpuPipelineConfigDesc* config, ommGpuPipeline* outPipeline)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    const Logger& L = untag<Baker>(baker)->log;
    if (config == 0) return L.invalid("[Invalid Arg] - pipeline config desc must be provided");
    if (tag_of(baker) != kGpuBaker) return L.invalid("[Invalid Arg] - invalid baker type");
    if (outPipeline) *outPipeline = nullptr;
    return gpu_baker_not_built(&L, "ommGpuCreatePipeline");
}
OMM_MI355X_API ommResult ommGpuDestroyPipeline(ommBaker baker, ommGpuPipeline pipeline)
{
    if (pipeline == 0 || baker == 0 || tag_of(baker) != kGpuBaker) return ommResult_INVALID_ARGUMENT;
    return gpu_baker_not_built(&untag<Baker>(baker)->log, "ommGpuDestroyPipeline");
}
OMM_MI355X_API ommResult ommGpuGetPipelineDesc(ommGpuPipeline pipeline, const ommGpuPipelineInfoDesc**)
{
    if (pipeline == 0) return ommResult_INVALID_ARGUMENT;
    return gpu_baker_not_built(nullptr, "ommGpuGetPipelineDesc");   // (no pipeline handle can exist: CreatePipeline never succeeds)
}
OMM_MI355X_API ommResult ommGpuGetPreDispatchInfo(ommGpuPipeline pipeline, const ommGpuDispatchConfigDesc* config, ommGpuPreDispatchInfo*)
{
    if (pipeline == 0 || config == 0) return ommResult_INVALID_ARGUMENT;
    return gpu_baker_not_built(nullptr, "ommGpuGetPreDispatchInfo");
}
OMM_MI355X_API ommResult ommGpuDispatch(ommGpuPipeline pipeline, const ommGpuDispatchConfigDesc* config, const ommGpuDispatchChain**)
{
    if (pipeline == 0 || config == 0) return ommResult_INVALID_ARGUMENT;
    return gpu_baker_not_built(nullptr, "ommGpuDispatch");
}
OMM_MI355X_API ommResult ommDebugSaveAsImages(ommBaker baker, const ommCpuBakeInputDesc* bakeInputDesc, const ommCpuBakeResultDesc*, const ommDebugSaveImagesDesc* desc)
{
    if (baker == 0 || bakeInputDesc == 0 || desc == 0) return ommResult_INVALID_ARGUMENT;
    if (tag_of(baker) != kCpuBaker && tag_of(baker) != kGpuBaker) return ommResult_INVALID_ARGUMENT;
    const Logger& L = untag<Baker>(baker)->log;
    L.msg(ommMessageSeverity_Fatal, "[Not Implemented] - ommDebugSaveAsImages: the PNG overlay dump (stb) is not part of the MI355X build");
    return ommResult_NOT_IMPLEMENTED;
}

OMM_MI355X_API ommResult ommxBakeDevice(ommBaker baker, const ommCpuBakeInputDesc* desc, ommxDeviceBakeResult* outResult)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    if (desc == 0 || outResult == 0) return b->log.invalid("input desc was not set");
    if (tag_of(baker) != kCpuBaker) return b->log.invalid("Baker was not created as the right type");
    if (desc->texture == 0) return b->log.invalid("[Invalid Argument] - ommCpuBakeInputDesc has no texture set");
    if (tag_of(desc->texture) == kTexture &&
        ((unsigned)desc->runtimeSamplerDesc.addressingMode >= (unsigned)ommTextureAddressMode_MAX_NUM ||
         (unsigned)desc->runtimeSamplerDesc.filter >= (unsigned)ommTextureFilterMode_MAX_NUM))
        return ommResult_FAILURE;
    ommResult r = validate_desc(*b, *desc);
    if (r != ommResult_SUCCESS) return r;
    r = scope_fences(*b, *desc, false);
    if (r != ommResult_SUCCESS) return r;
    return guarded(&b->log, [&]() -> ommResult {
    const double t0 = now_ms();
    const DeviceScope onBakersDevice(b->bind_device());
    BakeSession ses(*b);
    if (!ses.open()) return b->log.failure("[Failure] - no usable HIP device (the MI355X baker has no CPU fallback)");
    DeviceBakeResult* res = b->mem.make<DeviceBakeResult>();
    if (!res) return ommResult_FAILURE;
    res->mem = b->mem; memset(&res->desc, 0, sizeof res->desc);
    EventTimer et(ses.stream);
    ommxBakeTimings tm; memset(&tm, 0, sizeof tm);
    DeviceInputs din; din.texCoords = desc->texCoords; din.indices = desc->indexBuffer; din.perTriLevels = desc->subdivisionLevels;
    if (wants_host_tail(*desc)) {
        // near-duplicate merge / size budget: classification on the device, the reference's serial tail on the host, result uploaded again
        HostTailRequest ht; DeviceResult scratchR;
        ommResult hr = bake_core(*b, *desc, din, nullptr, ses.arena, ses.states, ses.stream, et, scratchR, tm, nullptr, &ht);
        HostTailResult hres; ommIndexFormat ifmt = ommIndexFormat_UINT_32;
        if (hr == ommResult_SUCCESS) hr = run_host_tail_for(*desc, desc->indexCount / 3u, ht.items, hres, ifmt);
        if (hr == ommResult_SUCCESS) hr = upload_host_tail(*b, hres, ifmt, desc->indexCount / 3u, (int)desc->format, ses.stream, res);
        if (hr != ommResult_SUCCESS) { (void)hipStreamSynchronize(ses.stream); b->mem.destroy(res); return hr; }
        tm.totalMs = (float)(now_ms() - t0);
        { std::lock_guard<std::mutex> g(b->timingsMu); b->timings = tm; b->haveTimings = true; }
        *outResult = (ommxDeviceBakeResult)res;
        return ommResult_SUCCESS;
    }
    r = bake_core(*b, *desc, din, nullptr, ses.arena, ses.states, ses.stream, et, res->R, tm);
    if (r != ommResult_SUCCESS) { (void)hipStreamSynchronize(ses.stream); b->mem.destroy(res); return r; } // (pooled blocks go back only when the stream is idle)
    uint32_t nAH = 0, nIH = 0;
    for (uint32_t l = 0; l < (uint32_t)kNumLevels; ++l) {
        if (res->R.hist[l]) { res->arrayHist[nAH].count = res->R.hist[l]; res->arrayHist[nAH].subdivisionLevel = (uint16_t)l; res->arrayHist[nAH].format = (uint16_t)res->R.bits; nAH++; }
        if (res->R.hist[kNumLevels + l]) { res->indexHist[nIH].count = res->R.hist[kNumLevels + l]; res->indexHist[nIH].subdivisionLevel = (uint16_t)l; res->indexHist[nIH].format = (uint16_t)res->R.bits; nIH++; }
    }
    res->desc.arrayData = res->R.arrayData; res->desc.arrayDataSize = (uint32_t)res->R.arrayDataSize;
    res->desc.descArray = res->R.descs; res->desc.descArrayCount = res->R.numDescs;
    res->desc.descArrayHistogram = res->arrayHist; res->desc.descArrayHistogramCount = nAH;
    res->desc.indexBuffer = res->R.index; res->desc.indexCount = res->R.numTris; res->desc.indexFormat = res->R.indexFormat;
    res->desc.indexHistogram = res->indexHist; res->desc.indexHistogramCount = nIH;
    tm.totalMs = (float)(now_ms() - t0);
    { std::lock_guard<std::mutex> g(b->timingsMu); b->timings = tm; b->haveTimings = true; }
    *outResult = (ommxDeviceBakeResult)res;
    return ommResult_SUCCESS;
    });
}

// ---- multi-GPU sharded bake (include/omm_mi355x_ext.h) ----
namespace {
struct ShardedBake {
    Allocator mem; Baker* baker = nullptr; BakeSession ses; ShardCtx ctx; ommxBakeTimings tm; double t0 = 0;
    std::unique_ptr<EventTimer> et;
    explicit ShardedBake(Baker& b) : baker(&b), ses(b) { memset(&tm, 0, sizeof tm); }
};

// ---- RCCL, bound at first use ----
// The collectives of the sharded bake are RCCL calls made from this library on its own streams (ommxShardedBakeRccl).  librccl is
// resolved with dlopen at the first call, so the drop-in keeps loading on machines (and for callers) that never shard a bake; when the
// process already holds an RCCL (e.g. torch's) its SONAME librccl.so.1 resolves to that instance.
typedef void* rcclComm_t;
struct RcclUniqueId { char internal[128]; };                       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
enum { kRcclSum = 0, kRcclMax = 2, kRcclMin = 3, kRcclUint8 = 1, kRcclUint32 = 3 };   // ncclRedOp_t (sum, max, min) / ncclDataType_t (uint8, uint32) values of rccl.h
struct RcclApi {
    void* dso = nullptr; std::string error;
    int (*getUniqueId)(RcclUniqueId*) = nullptr;
    int (*commInitRank)(rcclComm_t*, int, RcclUniqueId, int) = nullptr;
    int (*commDestroy)(rcclComm_t) = nullptr;
    int (*commCount)(rcclComm_t, int*) = nullptr;
    int (*commUserRank)(rcclComm_t, int*) = nullptr;
    int (*allReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*allGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    const char* (*getErrorString)(int) = nullptr;
    bool ok() const { return dso != nullptr && error.empty(); }
};
const RcclApi& rccl()
{
    static const RcclApi api = [] {
        RcclApi a;
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { a.dso = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (a.dso) break; }
        if (!a.dso) { a.error = "librccl.so.1 could not be loaded"; return a; }
        auto sym = [&](const char* n) { void* p = dlsym(a.dso, n); if (!p && a.error.empty()) a.error = std::string("librccl lacks ") + n; return p; };
        a.getUniqueId = (int (*)(RcclUniqueId*))sym("ncclGetUniqueId");
        a.commInitRank = (int (*)(rcclComm_t*, int, RcclUniqueId, int))sym("ncclCommInitRank");
        a.commDestroy = (int (*)(rcclComm_t))sym("ncclCommDestroy");
        a.commCount = (int (*)(rcclComm_t, int*))sym("ncclCommCount");
        a.commUserRank = (int (*)(rcclComm_t, int*))sym("ncclCommUserRank");
        a.allReduce = (int (*)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t))sym("ncclAllReduce");
        a.allGather = (int (*)(const void*, void*, size_t, int, rcclComm_t, hipStream_t))sym("ncclAllGather");
        a.getErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        return a;
    }();
    return api;
}
struct RcclComm {
    rcclComm_t comm = nullptr; bool owned = false; int rank = 0, world = 1;
    uint32_t* dStatus = nullptr; hipStream_t statusStream = nullptr;   // status agreement (rccl_agree): two device words and a stream for ranks that have no bake stream
    int statusDevice = -1;                                             // ... and the device they live on (rccl_status_on_device)
    bool custom = false; ommxCollectives user{};                       // ommxCommFromCollectives: the caller's transport instead of RCCL
    // the two collectives of the sharded bake (uint32 all-reduce, byte all-gather), stream-ordered; 0 = success
    int all_reduce(const void* send, void* recv, size_t count, int rcclOp, hipStream_t stream) const
    {
        if (!custom) return rccl().allReduce(send, recv, count, kRcclUint32, rcclOp, comm, stream);
        return user.allReduceU32(user.user, send, recv, count, rcclOp == kRcclSum ? ommxReduceOp_Sum : (rcclOp == kRcclMax ? ommxReduceOp_Max : ommxReduceOp_Min), (void*)stream);
    }
    int all_gather(const void* send, void* recv, size_t bytesPerRank, hipStream_t stream) const
    {
        if (!custom) return rccl().allGather(send, recv, bytesPerRank, kRcclUint8, comm, stream);
        return user.allGatherBytes(user.user, send, recv, bytesPerRank, (void*)stream);
    }
    const char* error_string(int code) const { return custom ? "the caller's collective reported a failure" : (rccl().getErrorString ? rccl().getErrorString(code) : "RCCL error"); }
};
// The two status words and the stream of idle ranks are set up when the communicator is made (on the device that is current then), so that a rank
// that has run out of device memory later can still take part in every agreement; rccl_agree() only allocates if that did not succeed.
void rccl_prepare_status(RcclComm* c)
{
    if (!HIP_OK(hipGetDevice(&c->statusDevice))) { c->statusDevice = -1; (void)hipGetLastError(); }
    if (!c->dStatus && !HIP_OK(hipMalloc((void**)&c->dStatus, 2 * sizeof(uint32_t)))) { c->dStatus = nullptr; (void)hipGetLastError(); }
    if (!c->statusStream && !HIP_OK(hipStreamCreateWithFlags(&c->statusStream, hipStreamNonBlocking))) { c->statusStream = nullptr; (void)hipGetLastError(); }
}
// The communicator may have been made before the caller selected the rank's device (ommxCommFromCollectives / ommxRcclCommWrap in front of
// torch.cuda.set_device): status words and stream of another GPU than the bake's would mix devices inside one collective.  Called under the baker's
// DeviceScope: anything that lives elsewhere is released and made again here.
void rccl_status_on_device(RcclComm* c, int device)
{
    if (c->statusDevice == device && c->dStatus && c->statusStream) return;
    if (c->statusDevice != device) {
        if (c->statusStream) { (void)hipStreamSynchronize(c->statusStream); (void)hipStreamDestroy(c->statusStream); c->statusStream = nullptr; }
        if (c->dStatus) { (void)hipFree(c->dStatus); c->dStatus = nullptr; }
        (void)hipGetLastError();
    }
    rccl_prepare_status(c);   // (on the current device = `device`)
}
// A rank-local failure (out of memory, mostly) must not leave the other ranks waiting in the next collective: before every data collective each rank
// contributes its status to a one-element MIN all-reduce and all of them go on, or none.  `stream`: the bake's stream (idle ranks: the communicator's own).
bool rccl_agree(RcclComm* rc, hipStream_t stream, bool mineOk, const Logger& L, const char* stage)
{
    bool ok = true;
    if (!rc->dStatus) ok = HIP_OK(hipMalloc((void**)&rc->dStatus, 2 * sizeof(uint32_t)));
    if (ok && !stream) { if (!rc->statusStream) ok = HIP_OK(hipStreamCreateWithFlags(&rc->statusStream, hipStreamNonBlocking)); stream = rc->statusStream; }
    if (!ok) { (void)hipGetLastError(); L.failure("[Failure] - sharded bake: no device memory for the status word (the other ranks may be waiting in a collective)"); return false; }
    const uint32_t mine = mineOk ? 1u : 0u; uint32_t all = 0;
    ok = HIP_OK(hipMemcpyAsync(rc->dStatus, &mine, sizeof mine, hipMemcpyHostToDevice, stream)) && HIP_OK(hipStreamSynchronize(stream));
    ok = ok && rc->all_reduce(rc->dStatus, rc->dStatus + 1, 1, kRcclMin, stream) == 0;
    ok = ok && HIP_OK(hipMemcpyAsync(&all, rc->dStatus + 1, sizeof all, hipMemcpyDeviceToHost, stream)) && HIP_OK(hipStreamSynchronize(stream));
    if (!ok) { L.failure("[Failure] - sharded bake: the status all-reduce failed"); return false; }
    if (mineOk && all == 0u) { char buf[200]; snprintf(buf, sizeof buf, "[Failure] - sharded bake: another rank failed (%s); this rank stops with it", stage); L.failure(buf); }
    return mineOk && all != 0u;
}
// The same agreement carrying a number: every rank contributes the device word *dWord (0xFFFFFFFF if it failed) to a MAX all-reduce; *outMax = the
// largest one.  false = a rank failed (this one or another).
bool rccl_agree_max(RcclComm* rc, hipStream_t stream, bool mineOk, uint32_t* dWord, uint32_t* outMax, const Logger& L, const char* stage)
{
    bool ok = true;
    if (!rc->dStatus) ok = HIP_OK(hipMalloc((void**)&rc->dStatus, 2 * sizeof(uint32_t)));
    if (!ok) { (void)hipGetLastError(); L.failure("[Failure] - sharded bake: no device memory for the status word (the other ranks may be waiting in a collective)"); return false; }
    const uint32_t failed = 0xFFFFFFFFu; uint32_t all = failed;
    if (!mineOk) ok = HIP_OK(hipMemcpyAsync(dWord, &failed, sizeof failed, hipMemcpyHostToDevice, stream)) && HIP_OK(hipStreamSynchronize(stream));
    ok = ok && rc->all_reduce(dWord, rc->dStatus + 1, 1, kRcclMax, stream) == 0;
    ok = ok && HIP_OK(hipMemcpyAsync(&all, rc->dStatus + 1, sizeof all, hipMemcpyDeviceToHost, stream)) && HIP_OK(hipStreamSynchronize(stream));
    if (!ok) { L.failure("[Failure] - sharded bake: the status all-reduce failed"); return false; }
    if (mineOk && all == failed) { char buf[200]; snprintf(buf, sizeof buf, "[Failure] - sharded bake: another rank failed (%s); this rank stops with it", stage); L.failure(buf); }
    *outMax = all;
    return mineOk && all != failed;
}

// The all-gather of the block contributions moves in chunks of <= 64 MiB per rank (at most 8 chunks, multiples of 256 bytes), so that the
// scatter of one chunk overlaps the transfer of the next.  ommxBakerKnob_ShardChunkBytes overrides the chunk size (tests use tiny chunks to
// force blocks across chunk boundaries).
uint64_t shard_chunk_bytes(const Baker& b, uint64_t strideBytes)
{
    uint64_t want = 64ull << 20;
    if (const uint64_t v = b.knob(ommxBakerKnob_ShardChunkBytes)) want = v;
    uint64_t chunks = (strideBytes + want - 1) / want; if (chunks > 8) chunks = 8; if (chunks < 1) chunks = 1;
    return ((((strideBytes + chunks - 1) / chunks) + 255) & ~255ull);
}

ommResult sharded_checks(ommBaker baker, const ommCpuBakeInputDesc* desc, Baker** outB, bool hostTailOk)
{
    if (baker == 0) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    if (desc == 0) return b->log.invalid("input desc was not set");
    if (tag_of(baker) != kCpuBaker) return b->log.invalid("Baker was not created as the right type");
    if (desc->texture == 0) return b->log.invalid("[Invalid Argument] - ommCpuBakeInputDesc has no texture set");
    if (tag_of(desc->texture) == kTexture &&
        ((unsigned)desc->runtimeSamplerDesc.addressingMode >= (unsigned)ommTextureAddressMode_MAX_NUM ||
         (unsigned)desc->runtimeSamplerDesc.filter >= (unsigned)ommTextureFilterMode_MAX_NUM))
        return ommResult_FAILURE;
    ommResult r = validate_desc(*b, *desc);
    if (r != ommResult_SUCCESS) return r;
    r = scope_fences(*b, *desc, false, hostTailOk, false);
    if (r != ommResult_SUCCESS) return r;
    *outB = b;
    return ommResult_SUCCESS;
}

// phase 1: replicated setup + triage, classification and digests of this rank's share, metadata words packed for the all-reduce.
// async: nothing is synchronised at the end (the RCCL all-reduce follows on the same stream).
ommResult sharded_begin(Baker* b, const ommCpuBakeInputDesc* desc, uint32_t rank, uint32_t worldSize, bool async, ShardedBake** out, bool mergeStates = false)
{
    ShardedBake* sb = b->mem.make<ShardedBake>(*b);
    if (!sb) return ommResult_FAILURE;
    sb->mem = b->mem; sb->t0 = now_ms();
    if (!sb->ses.open()) { b->mem.destroy(sb); return b->log.failure("[Failure] - no usable HIP device (the MI355X baker has no CPU fallback)"); }
    sb->ctx.rank = rank; sb->ctx.world = worldSize; sb->ctx.asyncBegin = async; sb->ctx.mergeStates = mergeStates;
    sb->et.reset(new EventTimer(sb->ses.stream));
    DeviceResult unused;
    DeviceInputs din; din.texCoords = desc->texCoords; din.indices = desc->indexBuffer; din.perTriLevels = desc->subdivisionLevels;
    const ommResult r = bake_core(*b, *desc, din, nullptr, sb->ses.arena, sb->ses.states, sb->ses.stream, *sb->et, unused, sb->tm, &sb->ctx);
    if (r != ommResult_SUCCESS) { b->mem.destroy(sb); return r; }
    *out = sb;
    return ommResult_SUCCESS;
}

// phase 2 (after the metadata all-reduce): replicated tail, per-rank layout, this rank's surviving blocks packed into its contribution.
// Ends with the contribution complete on the stream (not synchronised); the two host read-backs inside (OMM count, per-rank totals) remain.
ommResult sharded_tail(ShardedBake* sb)
{
    ShardCtx& c = sb->ctx; const Logger& L = sb->baker->log; hipStream_t stream = sb->ses.stream;
    const uint32_t numActive = c.hc.activeStart[kNumLevels];
    if (!HIP_OK(hipMemsetAsync(c.dOwner, 0xFF, c.ti.numItems ? c.ti.numItems : 1, stream))) return L.failure("[Failure] - device memset failed");
    launch_shard_unpack_meta(c.bounds, c.dActiveIds, numActive, c.dMeta, c.dMask, (uint32_t*)c.ti.knownCount, c.ti.digests, c.dOwner, stream);
    if (!HIP_OK(run_tail(c.ti, c.to, c.dScratch, c.scratchBytes, &c.counts, stream))) return L.failure("[Failure] - device tail failed");
    if (c.counts.arrayDataSize > 0xFFFFFFFFull) return ommResult_FAILURE; // bake_cpu_impl.cpp:1774-1775
    if (!HIP_OK(run_shard_layout(c.to.order, c.to.sizes, c.dActive, c.dOwner, c.counts.numOmms, c.world, c.dCofs, c.dTotals, c.totals, c.dScratch, c.scratchBytes, stream)))
        return L.failure("[Failure] - sharded layout failed");
    uint64_t mx = 0; for (uint32_t r = 0; r < c.world; ++r) mx = c.totals[r] > mx ? c.totals[r] : mx;
    c.strideBytes = (mx + 255) & ~255ull; if (c.strideBytes == 0) c.strideBytes = 256;
    // contribution + (for the RCCL path) the gather staging of all ranks: one grow-only block of the session's working set
    const size_t gatherBytes = c.asyncBegin ? (size_t)c.strideBytes * c.world + 4096 : 0;
    // (RCCL path) the contributions cross the links as codec streams of at most half their size: this rank's and the gathered ones
    c.compCap = c.asyncBegin ? pad256((size_t)(c.strideBytes / 2) + 4096) : 0;
    // (the codec's count / scan scratch grows with the contribution -- one word per 4 KiB of it plus rocPRIM's -- while the bake's own scratch block is sized
    //  from the triangle count: a few large blocks, level 10 and up, need more than that, so it comes from this arena)
    c.codecScratchBytes = c.asyncBegin ? pad256(shard_codec_scratch_bytes(c.strideBytes)) : 0;
    const size_t codecBytes = c.asyncBegin ? (size_t)c.compCap * (c.world + 1) + 1024 + c.codecScratchBytes : 0;
    if (!sb->ses.set->xchg.reserve((size_t)c.strideBytes + 256 + gatherBytes + codecBytes)) return L.failure("[Failure] - out of device memory for the shard contribution");
    c.dContrib = sb->ses.set->xchg.take<uint8_t>((size_t)c.strideBytes);
    c.dGathered = gatherBytes ? sb->ses.set->xchg.take<uint8_t>(gatherBytes) : nullptr;
    if (codecBytes) { c.dComp = sb->ses.set->xchg.take<uint8_t>((size_t)c.compCap); c.dGatherComp = sb->ses.set->xchg.take<uint8_t>((size_t)c.compCap * c.world); c.dCompSize = sb->ses.set->xchg.take<uint32_t>(1); c.dCodecScratch = sb->ses.set->xchg.take<uint8_t>(c.codecScratchBytes); }
    // (the padding behind this rank's blocks travels too: zeros, which the codec folds away)
    if (c.strideBytes > c.totals[c.rank] && !HIP_OK(hipMemsetAsync(c.dContrib + c.totals[c.rank], 0, (size_t)(c.strideBytes - c.totals[c.rank]), stream))) return L.failure("[Failure] - device memset failed");
    launch_shard_gather(c.dStates, c.dStateOfs, c.dActive, c.dOwner, c.rank, c.to.order, c.dCofs, c.to.sizes, c.counts.numOmms, c.dContrib, stream);
    return HIP_OK(hipGetLastError()) ? ommResult_SUCCESS : L.failure("[Failure] - shard gather failed");
}

// phase 3: result buffers, descriptors, index buffer; `scatter` places the gathered blocks (one call or one per chunk)
template <class ScatterFn>
ommResult sharded_finish(ShardedBake* sb, ScatterFn&& scatter, ommxDeviceBakeResult* outResult, bool wantArray = true)
{
    ShardCtx& c = sb->ctx; Baker* b = sb->baker; const Logger& L = b->log; hipStream_t stream = sb->ses.stream;
    const uint32_t E = c.counts.numOmms, T = c.T;
    DeviceBakeResult* res = b->mem.make<DeviceBakeResult>();
    if (!res) return ommResult_FAILURE;
    res->mem = b->mem; memset(&res->desc, 0, sizeof res->desc);
    DeviceResult& R = res->R;
    R.pool = b->devPool;
    R.bits = c.bits; R.numDescs = E; R.arrayDataSize = E ? c.counts.arrayDataSize : 0; R.numTris = T;
    bool ok = true;
    if (E) {
        R.arrayData = wantArray ? (uint8_t*)R.dev_alloc((size_t)c.counts.arrayDataSize) : nullptr; R.descs = (ommCpuOpacityMicromapDesc*)R.dev_alloc(sizeof(ommCpuOpacityMicromapDesc) * (size_t)E);
        ok = (R.arrayData != nullptr || !wantArray) && R.descs != nullptr;
        ok = scatter(ok ? R.arrayData : nullptr) && ok;   // (called either way: the RCCL form agrees on the allocation across ranks before its all-gathers)
        if (ok) launch_write_descs(c.to.order, c.to.dstOfs, c.dLevel, c.bits, E, R.descs, stream);
    }
    const bool allow8 = (c.flags & (1u << 6)) != 0, force32 = (c.flags & (1u << 2)) != 0;
    int idxBytes = 4; R.indexFormat = ommIndexFormat_UINT_32;
    if (allow8 && T <= 127 && !force32) { idxBytes = 1; R.indexFormat = ommIndexFormat_UINT_8; }
    else if (T <= 32767 && !force32) { idxBytes = 2; R.indexFormat = ommIndexFormat_UINT_16; }
    R.index = R.dev_alloc((size_t)(T ? T : 1) * 4);
    ok = ok && R.index != nullptr;
    if (ok) launch_narrow_indices(c.dIndex, T, idxBytes, R.index, stream);
    ok = ok && HIP_OK(hipMemcpyAsync(R.hist, c.dArrayHist, sizeof(uint32_t) * kNumLevels, hipMemcpyDeviceToHost, stream));
    ok = ok && HIP_OK(hipMemcpyAsync(R.hist + kNumLevels, c.dIndexHist, sizeof(uint32_t) * kNumLevels, hipMemcpyDeviceToHost, stream));
    ok = ok && HIP_OK(hipStreamSynchronize(stream));
    if (!ok) { (void)hipStreamSynchronize(stream); b->mem.destroy(res); return L.failure("[Failure] - could not materialise the merged bake result on the device"); }
    uint32_t nAH = 0, nIH = 0;
    for (uint32_t l = 0; l < (uint32_t)kNumLevels; ++l) {
        if (R.hist[l]) { res->arrayHist[nAH].count = R.hist[l]; res->arrayHist[nAH].subdivisionLevel = (uint16_t)l; res->arrayHist[nAH].format = (uint16_t)R.bits; nAH++; }
        if (R.hist[kNumLevels + l]) { res->indexHist[nIH].count = R.hist[kNumLevels + l]; res->indexHist[nIH].subdivisionLevel = (uint16_t)l; res->indexHist[nIH].format = (uint16_t)R.bits; nIH++; }
    }
    res->desc.arrayData = R.arrayData; res->desc.arrayDataSize = (uint32_t)R.arrayDataSize;
    res->desc.descArray = R.descs; res->desc.descArrayCount = R.numDescs;
    res->desc.descArrayHistogram = res->arrayHist; res->desc.descArrayHistogramCount = nAH;
    res->desc.indexBuffer = R.index; res->desc.indexCount = R.numTris; res->desc.indexFormat = R.indexFormat;
    res->desc.indexHistogram = res->indexHist; res->desc.indexHistogramCount = nIH;
    // the bake's HIP events are complete now (Begin may have returned without synchronising)
    const int* e = c.ev;
    sb->tm.setupMs = sb->et->ms(e[0], e[1]); sb->tm.triageMs = sb->et->ms(e[1], e[2]); sb->tm.classifyMs = sb->et->ms(e[2], e[3]); sb->tm.digestMs = sb->et->ms(e[3], e[4]);
    sb->tm.totalMs = (float)(now_ms() - sb->t0);
    { std::lock_guard<std::mutex> g(b->timingsMu); b->timings = sb->tm; b->haveTimings = true; }
    *outResult = (ommxDeviceBakeResult)res;
    return ommResult_SUCCESS;
}

// ================================================================================================
// ommCpuBake over several devices of one process (ommxBakerKnob_Devices = N; SURVEY.md section 8e behind the SDK's entry point, omm.h:574).
//
// One host thread per device ("rank"; rank 0 = the caller's thread on the baker's device, rank r on device (primary + r) mod the device count -- with fewer
// devices than ranks several ranks share one, which is how the path is tested on a one-GPU box).  Every rank uploads the triangle data over its own PCIe
// link and runs the sharded bake's phases on its device: set-up and triage replicated, classification and digests of ITS share of the active work items
// (sharded_begin), the 16 bytes of metadata per active item summed across the ranks THROUGH HOST MEMORY (no device-to-device traffic at all), the
// deterministic tail replicated (sharded_tail).  Then every rank turns the blocks it owns into ONE codec stream and sends it over its own link; the host
// writes every block of the result from its owner's stream (codec_scatter_omms, the helper threads).  Descriptors, index buffer and histograms come from rank 0.
// No all-gather of the array: the 1.27 GB of the metric configuration cross the links as N streams of ~82 / N MB.
// ================================================================================================
struct RankTeam {
    uint32_t n = 1; std::mutex mu; std::condition_variable cv; uint32_t arrived = 0; uint64_t generation = 0; bool allOk = true, result = true, aborted = false;
    // every rank arrives with its status; all leave with the conjunction.  A rank that cannot go on at all (an exception) calls abort(): every rank that waits
    // or arrives later leaves with false at once, so nobody waits for a rank that is gone.
    bool barrier(bool ok) {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        allOk = allOk && ok;
        if (++arrived == n) { result = allOk; allOk = true; arrived = 0; ++generation; cv.notify_all(); return result; }
        const uint64_t g = generation;
        cv.wait(lk, [&] { return generation != g || aborted; });
        return aborted ? false : result;
    }
    void abort() { std::lock_guard<std::mutex> g(mu); aborted = true; cv.notify_all(); }
};
// the texture of `d` on device `dev` (the original when that is where it lives)
Texture* texture_on_device(Texture* t, int dev)
{
    if (t->device < 0 || t->device == dev) return t;
    std::lock_guard<std::mutex> g(t->replicaMu);
    for (auto& r : t->replicas) if (r->device == dev) return r.get();
    const DeviceScope onDev(dev);
    std::unique_ptr<Texture> r(new (std::nothrow) Texture());
    if (!r) return nullptr;
    r->mem = t->mem; r->log = t->log; r->format = t->format; r->flags = t->flags; r->alphaCutoff = t->alphaCutoff; r->device = dev;
    const size_t px = t->format == ommCpuTextureFormat_FP32 ? 4 : 1;
    for (const TexMip& m : t->mips) {
        TexMip c = m; c.texels = nullptr; c.sat = nullptr;
        const size_t n = (size_t)m.w * (size_t)m.h;
        bool ok = HIP_OK(hipMalloc((void**)&c.texels, n * px)) && HIP_OK(hipMemcpy(c.texels, m.texels, n * px, hipMemcpyDefault));
        if (ok && m.sat) ok = HIP_OK(hipMalloc((void**)&c.sat, n * 4)) && HIP_OK(hipMemcpy(c.sat, m.sat, n * 4, hipMemcpyDefault));
        r->mips.push_back(c);   // (freed by ~Texture, also when the copy failed half way)
        if (!ok) { (void)hipGetLastError(); return nullptr; }
    }
    t->replicas.push_back(std::move(r));
    return t->replicas.back().get();
}
Baker* peer_baker(Baker& b, uint32_t rank, int dev)
{
    if (rank == 0) return &b;
    std::lock_guard<std::mutex> g(b.peersMu);
    while (b.peers.size() < rank) {
        std::unique_ptr<Baker> p(new (std::nothrow) Baker());
        if (!p) return nullptr;
        p->mem = b.mem; p->log = b.log; p->type = b.type;
        b.peers.push_back(std::move(p));
    }
    Baker* p = b.peers[rank - 1].get();
    p->device.store(dev);
    for (int k = 0; k < (int)ommxBakerKnob_MAX_NUM; ++k) p->knobs[k].store(k == (int)ommxBakerKnob_Devices ? 0 : b.knobs[k].load());
    p->arenas->retain.store(b.arenas->retain.load()); p->devPool->retain.store(b.devPool->retain.load()); p->hostPool->retain.store(b.hostPool->retain.load());
    return p;
}

ommResult bake_impl_multi(Baker& baker, const ommCpuBakeInputDesc& d, ommCpuBakeResult* out, uint32_t N)
{
    const Logger& L = baker.log;
    const double t0 = now_ms();
    const uint32_t T = d.indexCount / 3u;
    const int primary = baker.bind_device();
    int deviceCount = 0;
    if (primary < 0 || !HIP_OK(hipGetDeviceCount(&deviceCount)) || deviceCount < 1) return L.failure("[Failure] - no usable HIP device (the MI355X baker has no CPU fallback)");
    if (N > (uint32_t)kMaxRanks) N = kMaxRanks;
    Texture* tex0 = untag<Texture>(d.texture);
    // ---- per rank: device, baker, texture (made before the threads start: replicas and shadow bakers are created once and kept) ----
    struct Rank {
        int dev = 0; Baker* baker = nullptr; Texture* tex = nullptr; ShardedBake* sb = nullptr; uint8_t* dRaw = nullptr; uint8_t* dCodec = nullptr;
        std::vector<uint32_t> meta; const uint8_t* hStream = nullptr; bool raw = false; ommResult status = ommResult_SUCCESS;
    };
    std::vector<Rank> ranks(N);
    for (uint32_t r = 0; r < N; ++r) {
        ranks[r].dev = (primary + (int)r) % deviceCount;
        ranks[r].baker = peer_baker(baker, r, ranks[r].dev);
        ranks[r].tex = texture_on_device(tex0, ranks[r].dev);
        if (!ranks[r].baker || !ranks[r].tex) return L.failure("[Failure] - multi-device bake: could not set up a device (texture copy / memory)");
    }
    const size_t idxSize = d.indexFormat == ommIndexFormat_UINT_8 ? 1 : (d.indexFormat == ommIndexFormat_UINT_16 ? 2 : 4);
    const uint32_t maxIndex = max_index(d.indexBuffer, d.indexFormat, 3ull * T);
    const uint32_t stride = d.texCoordStrideInBytes ? d.texCoordStrideInBytes : (d.texCoordFormat == ommTexCoordFormat_UV32_FLOAT ? 8u : 4u);
    const size_t elem = d.texCoordFormat == ommTexCoordFormat_UV32_FLOAT ? 8 : 4;
    const size_t uvBytes = T ? (size_t)stride * maxIndex + elem : 0, idxBytes = idxSize * 3ull * T, lvlBytes = d.subdivisionLevels ? T : 0;

    RankTeam team; team.n = N;
    std::vector<uint32_t> metaSum; size_t metaWords = 0;
    HostCodecLayout layout{}; uint64_t strideBytes = 0;
    // phase A (all ranks): upload, share of the classification, metadata through the host, tail, own contribution as a codec stream on the host
    auto body = [&](uint32_t r) {
        Rank& R = ranks[r];
        const DeviceScope onDev(R.dev);
        Baker& sub = *R.baker;
        bool ok = true;
        R.dRaw = (uint8_t*)sub.devPool->acquire(pad256(uvBytes) + pad256(idxBytes) + pad256(lvlBytes) + 256);
        ok = R.dRaw != nullptr;
        if (ok && uvBytes) ok = HIP_OK(hipMemcpy(R.dRaw, d.texCoords, uvBytes, hipMemcpyHostToDevice));
        if (ok && idxBytes) ok = HIP_OK(hipMemcpy(R.dRaw + pad256(uvBytes), d.indexBuffer, idxBytes, hipMemcpyHostToDevice));
        if (ok && lvlBytes) ok = HIP_OK(hipMemcpy(R.dRaw + pad256(uvBytes) + pad256(idxBytes), d.subdivisionLevels, lvlBytes, hipMemcpyHostToDevice));
        ommCpuBakeInputDesc dd = d;
        dd.texture = (ommCpuTexture)((uintptr_t)R.tex | kTexture);
        dd.texCoords = R.dRaw; dd.indexBuffer = R.dRaw + pad256(uvBytes); dd.subdivisionLevels = lvlBytes ? R.dRaw + pad256(uvBytes) + pad256(idxBytes) : nullptr;
        if (ok) { R.status = sharded_begin(&sub, &dd, r, N, false, &R.sb); ok = R.status == ommResult_SUCCESS; if (!ok) R.sb = nullptr; }
        else R.status = ommResult_FAILURE;
        if (!team.barrier(ok)) return;
        // ---- metadata of the active items (mask, known count, digest: 4 words each; zero outside a rank's share): summed across the ranks in host memory ----
        ShardCtx& c = R.sb->ctx; hipStream_t stream = R.sb->ses.stream;
        const size_t words = 4ull * c.hc.activeStart[kNumLevels];
        R.meta.resize(words ? words : 1);
        if (words) ok = HIP_OK(hipMemcpy(R.meta.data(), c.dMeta, words * 4, hipMemcpyDeviceToHost));
        if (r == 0) { metaWords = words; metaSum.assign(words ? words : 1, 0u); }
        if (!team.barrier(ok)) return;
        if (!team.barrier(words == metaWords)) return;   // (every rank ran the same set-up: anything else is an internal error -- and nobody may read a shorter copy)
        for (size_t k = metaWords * r / N; k < metaWords * (r + 1) / N; ++k) { uint32_t a = 0; for (uint32_t q = 0; q < N; ++q) a += ranks[q].meta[k]; metaSum[k] = a; }
        if (!team.barrier(true)) return;
        if (words) ok = HIP_OK(hipMemcpy(c.dMeta, metaSum.data(), words * 4, hipMemcpyHostToDevice));
        // ---- replicated tail, layout, this rank's blocks packed into its contribution ----
        if (ok) { R.status = sharded_tail(R.sb); ok = R.status == ommResult_SUCCESS; }
        // ---- the contribution as a codec stream, over this device's own link ----
        const uint64_t padded = c.strideBytes;
        const HostCodecLayout Lc = host_codec_layout(padded);
        const uint64_t cap = Lc.offRaw + padded / 2u + 16u;
        const size_t scratchBytes = pad256(shard_codec_scratch_bytes(padded));
        if (ok) { R.dCodec = (uint8_t*)sub.devPool->acquire(256 + scratchBytes + (size_t)cap); ok = R.dCodec != nullptr && Lc.blocks < 0x7FFFFFFFull; }
        uint64_t streamBytes = 0;
        if (ok) {
            uint8_t* dComp = R.dCodec + 256 + scratchBytes;
            ok = HIP_OK(run_shard_compress(c.dContrib, padded, dComp, cap, (uint32_t*)R.dCodec, R.dCodec + 256, scratchBytes, stream))
              && HIP_OK(hipMemcpyAsync(&streamBytes, dComp, 8, hipMemcpyDeviceToHost, stream)) && HIP_OK(hipStreamSynchronize(stream));
            R.raw = ok && (streamBytes > cap || streamBytes < Lc.offRaw);
            const uint64_t take = R.raw ? c.totals[r] : streamBytes;
            ok = ok && R.sb->ses.set->pinned.reserve((size_t)take + 4096);
            if (ok && take) ok = HIP_OK(hipMemcpyAsync(R.sb->ses.set->pinned.base, R.raw ? c.dContrib : dComp, (size_t)take, hipMemcpyDeviceToHost, stream)) && HIP_OK(hipStreamSynchronize(stream));
            R.hStream = R.sb->ses.set->pinned.base;
        }
        if (r == 0) { layout = Lc; strideBytes = padded; }
        (void)team.barrier(ok);
    };
    // rank 0 is the calling thread; a rank whose thread cannot be started is run by the caller AFTER the others would deadlock the barriers: refuse instead
    std::vector<std::thread> threads;
    bool spawned = true;
    // (nothing may leave a rank's thread: an exception -- std::bad_alloc of a host vector -- fails the rank and releases the others from their barriers)
    auto guardedBody = [&](uint32_t r) { try { body(r); } catch (...) { ranks[r].status = ommResult_FAILURE; team.abort(); } };
    try { threads.reserve(N); for (uint32_t r = 1; r < N; ++r) threads.emplace_back(guardedBody, r); } catch (...) { spawned = false; }
    if (!spawned) {
        team.abort();   // (threads that did start wait in the first barrier for ranks that never come)
        for (auto& t : threads) t.join();
        for (Rank& R : ranks) { const DeviceScope onDev(R.dev); if (R.sb) R.baker->mem.destroy(R.sb); if (R.dRaw) R.baker->devPool->release(R.dRaw); if (R.dCodec) R.baker->devPool->release(R.dCodec); }
        return L.failure("[Failure] - multi-device bake: could not start a thread per device");
    }
    guardedBody(0);
    for (auto& t : threads) t.join();
    struct Cleanup { std::vector<Rank>& ranks; ~Cleanup() { for (Rank& R : ranks) { const DeviceScope onDev(R.dev); if (R.sb) R.baker->mem.destroy(R.sb); if (R.dRaw) R.baker->devPool->release(R.dRaw); if (R.dCodec) R.baker->devPool->release(R.dCodec); } } } cleanup{ ranks };
    for (const Rank& R : ranks) if (R.status != ommResult_SUCCESS) return R.status;
    for (const Rank& R : ranks) if (!R.sb || !R.hStream) return L.failure("[Failure] - multi-device bake: a device failed");
    const double tA = now_ms();

    // ---- rank 0: layout tables to the host, descriptors / index buffer / histograms, then the blocks from their owners' streams ----
    const DeviceScope onPrimary(primary);
    ShardedBake* sb0 = ranks[0].sb; ShardCtx& c0 = sb0->ctx; hipStream_t stream0 = sb0->ses.stream;
    const uint32_t E = c0.counts.numOmms, U = c0.ti.numItems;
    std::vector<uint32_t> hOrder(E ? E : 1), hDstOfs(E ? E : 1), hSizes(E ? E : 1), hMask(U ? U : 1); std::vector<uint64_t> hCofs(E ? E : 1);
    std::vector<uint8_t> hActive(U ? U : 1), hOwner(U ? U : 1), hLevel(U ? U : 1);
    // Asynchronous copies into the vectors above (and, further down, into the result's arrays) are queued on stream0: whatever way this function is left --
    // a failed copy half way down a chain, an exception of a host container or of the thread pool -- the stream is drained BEFORE their memory is released
    // (destructors run in reverse order of declaration: each guard is declared right behind the memory it protects).
    struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } };
    const SyncOnExit drainBeforeVectors{ stream0 };
    bool ok = true;
    if (E) ok = HIP_OK(hipMemcpyAsync(hOrder.data(), c0.to.order, (size_t)E * 4, hipMemcpyDeviceToHost, stream0)) && HIP_OK(hipMemcpyAsync(hDstOfs.data(), c0.to.dstOfs, (size_t)E * 4, hipMemcpyDeviceToHost, stream0))
              && HIP_OK(hipMemcpyAsync(hSizes.data(), c0.to.sizes, (size_t)E * 4, hipMemcpyDeviceToHost, stream0)) && HIP_OK(hipMemcpyAsync(hCofs.data(), c0.dCofs, (size_t)E * 8, hipMemcpyDeviceToHost, stream0));
    if (ok && U) ok = HIP_OK(hipMemcpyAsync(hActive.data(), c0.dActive, U, hipMemcpyDeviceToHost, stream0)) && HIP_OK(hipMemcpyAsync(hOwner.data(), c0.dOwner, U, hipMemcpyDeviceToHost, stream0))
                   && HIP_OK(hipMemcpyAsync(hLevel.data(), c0.dLevel, U, hipMemcpyDeviceToHost, stream0)) && HIP_OK(hipMemcpyAsync(hMask.data(), c0.dMask, (size_t)U * 4, hipMemcpyDeviceToHost, stream0));
    ommxDeviceBakeResult dres = nullptr;
    if (ok) { const ommResult fr = sharded_finish(sb0, [](uint8_t*) { return true; }, &dres, false); if (fr != ommResult_SUCCESS) return fr; }   // (synchronises the stream: the copies above are complete)
    if (!ok) return L.failure("[Failure] - device to host transfer of the bake result failed");
    DeviceBakeResult* dr = (DeviceBakeResult*)dres;
    struct DresGuard { Baker& b; DeviceBakeResult* p; ~DresGuard() { b.mem.destroy(p); } } dresGuard{ baker, dr };
    DeviceResult& DR = dr->R;

    BakeResult* res = baker.mem.make<BakeResult>();
    if (!res) return ommResult_FAILURE;
    res->mem = baker.mem;
    struct ResGuard { Baker& b; BakeResult*& r; ~ResGuard() { if (r) b.mem.destroy(r); } } resGuard{ baker, res };
    const SyncOnExit drainBeforeResult{ stream0 };
    if (E) {
        if (baker.mem.alloc == default_alloc && baker.hostPool->wants((size_t)DR.arrayDataSize)) { res->arrayData = baker.hostPool->acquire((size_t)DR.arrayDataSize); if (res->arrayData) res->pool = baker.hostPool; }
        if (!res->arrayData) res->arrayData = baker.mem.allocate((size_t)DR.arrayDataSize, 64);
        res->descs = (ommCpuOpacityMicromapDesc*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapDesc) * (size_t)E, 16);
    }
    res->index = (int32_t*)baker.mem.allocate(sizeof(int32_t) * (size_t)(T ? T : 1), 16);
    res->triArea = (float*)baker.mem.allocate(sizeof(float) * (size_t)(T ? T : 1), 16);
    res->arrayHist = (ommCpuOpacityMicromapUsageCount*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels, 16);
    res->indexHist = (ommCpuOpacityMicromapUsageCount*)baker.mem.allocate(sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels, 16);
    if ((E && (!res->arrayData || !res->descs)) || !res->index || !res->triArea || !res->arrayHist || !res->indexHist) return L.failure("[Failure] - the memory allocator returned null for the bake result");
    const size_t outIdx = DR.indexFormat == ommIndexFormat_UINT_8 ? 1 : (DR.indexFormat == ommIndexFormat_UINT_16 ? 2 : 4);
    if (E) ok = HIP_OK(hipMemcpyAsync(res->descs, DR.descs, sizeof(ommCpuOpacityMicromapDesc) * (size_t)E, hipMemcpyDeviceToHost, stream0));
    if (ok && T) ok = HIP_OK(hipMemcpyAsync(res->index, DR.index, outIdx * T, hipMemcpyDeviceToHost, stream0)) && HIP_OK(hipMemcpyAsync(res->triArea, c0.dTriArea, sizeof(float) * (size_t)T, hipMemcpyDeviceToHost, stream0));
    // the blocks: OMM ranges of ~2 MiB each to the helper threads, while the small arrays above are on their way
    uint32_t threadsUsed = 1;
    if (ok && E) {
        HostScatter S; memset(&S, 0, sizeof S);
        S.world = N; S.L = layout; S.bits = c0.bits;
        for (uint32_t r = 0; r < N; ++r) { S.stream[r] = ranks[r].hStream; S.raw[r] = ranks[r].raw; }
        S.active = hActive.data(); S.owner = hOwner.data(); S.level = hLevel.data(); S.stateMask = hMask.data(); S.order = hOrder.data(); S.dstOfs = hDstOfs.data();
        S.sizes = hSizes.data(); S.cofs = hCofs.data(); S.arrayData = (uint8_t*)res->arrayData;
        std::vector<uint32_t> cut; cut.push_back(0);
        uint64_t acc = 0; for (uint32_t j = 0; j < E; ++j) { acc += hSizes[j]; if (acc >= ((uint64_t)2 << 20)) { cut.push_back(j + 1); acc = 0; } }
        if (cut.back() != E) cut.push_back(E);
        unsigned nth = effective_cpus() * 3u / 4u; nth = nth > 12u ? 12u : (nth < 1u ? 1u : nth);
        if (const uint64_t k = baker.knob(ommxBakerKnob_ExpandThreads)) nth = (unsigned)k;
        const std::shared_ptr<WorkerPool> pool = baker.worker_pool(nth);
        if (baker.knob(ommxBakerKnob_HelperAffinity) == 0) (void)pool->bind_near(res->arrayData);
        pool->run((uint32_t)cut.size() - 1u, [&](uint32_t t) { codec_scatter_omms(S, cut[t], cut[t + 1]); });
        threadsUsed = pool->workers() + 1u;
    }
    ok = ok && HIP_OK(hipStreamSynchronize(stream0));
    if (!ok) return L.failure("[Failure] - device to host transfer of the bake result failed");
    memcpy(res->arrayHist, dr->arrayHist, sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels);
    memcpy(res->indexHist, dr->indexHist, sizeof(ommCpuOpacityMicromapUsageCount) * 2 * kNumLevels);
    res->desc.arrayData = E ? res->arrayData : nullptr; res->desc.arrayDataSize = E ? (uint32_t)DR.arrayDataSize : 0;
    res->desc.descArray = E ? res->descs : nullptr; res->desc.descArrayCount = E;
    res->desc.descArrayHistogram = res->arrayHist; res->desc.descArrayHistogramCount = dr->desc.descArrayHistogramCount;
    res->desc.indexBuffer = res->index; res->desc.indexCount = T; res->desc.indexFormat = DR.indexFormat;
    res->desc.indexHistogram = res->indexHist; res->desc.indexHistogramCount = dr->desc.indexHistogramCount;
    {
        std::lock_guard<std::mutex> g(baker.timingsMu);   // (sharded_finish stored rank 0's phase clocks)
        uint64_t wire = 0; for (uint32_t r = 0; r < N; ++r) wire += ranks[r].raw ? c0.totals[r] : (ranks[r].hStream ? *(const uint64_t*)ranks[r].hStream : 0);
        baker.timings.devices = N; baker.timings.resultTransfer = ommxResultTransfer_Compressed; baker.timings.compressedBytes = wire; baker.timings.expandThreads = threadsUsed;
        baker.timings.expandMs = (float)(now_ms() - tA); baker.timings.totalMs = (float)(now_ms() - t0); baker.timings.contributionBytes = strideBytes;
    }
    *out = (ommCpuBakeResult)res;
    res = nullptr;
    return ommResult_SUCCESS;
}
} // namespace

OMM_MI355X_API ommResult ommxShardedBegin(ommBaker baker, const ommCpuBakeInputDesc* desc, uint32_t rank, uint32_t worldSize, ommxShardedBake* out)
{
    Baker* b = nullptr;
    if (out == 0) return ommResult_INVALID_ARGUMENT;
    const ommResult r = sharded_checks(baker, desc, &b, false);
    if (r != ommResult_SUCCESS) return r;
    if (worldSize == 0 || worldSize > (uint32_t)kMaxRanks || rank >= worldSize) return b->log.invalid("[Invalid Argument] - rank / worldSize out of range (at most 16 ranks)");
    return guarded(&b->log, [&]() -> ommResult {
        const DeviceScope onBakersDevice(b->bind_device());
        ShardedBake* sb = nullptr;
        const ommResult rr = sharded_begin(b, desc, rank, worldSize, false, &sb);
        if (rr == ommResult_SUCCESS) *out = (ommxShardedBake)sb;
        return rr;
    });
}

OMM_MI355X_API ommResult ommxShardedGetMeta(ommxShardedBake h, void** deviceWords, uint64_t* numWords)
{
    if (h == 0 || deviceWords == nullptr || numWords == nullptr) return ommResult_INVALID_ARGUMENT;
    ShardedBake* sb = (ShardedBake*)h;
    *deviceWords = sb->ctx.dMeta; *numWords = 4ull * sb->ctx.hc.activeStart[kNumLevels];
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxShardedTail(ommxShardedBake h, void** contribution, uint64_t* contributionBytes, uint64_t* strideBytes)
{
    if (h == 0 || contribution == nullptr || contributionBytes == nullptr || strideBytes == nullptr) return ommResult_INVALID_ARGUMENT;
    ShardedBake* sb = (ShardedBake*)h; ShardCtx& c = sb->ctx;
    return guarded(&sb->baker->log, [&]() -> ommResult {
        const DeviceScope onBakersDevice(sb->baker->bind_device());
        const double t1 = now_ms();
        const ommResult r = sharded_tail(sb);   // (a second call re-uses the same exchange block: nothing leaks)
        if (r != ommResult_SUCCESS) return r;
        if (!HIP_OK(hipStreamSynchronize(sb->ses.stream))) return sb->baker->log.failure("[Failure] - shard gather failed");
        *contribution = c.dContrib; *contributionBytes = c.totals[c.rank]; *strideBytes = c.strideBytes;
        sb->tm.tailMs = (float)(now_ms() - t1);
        return ommResult_SUCCESS;
    });
}

OMM_MI355X_API ommResult ommxShardedFinish(ommxShardedBake h, const void* gathered, ommxDeviceBakeResult* outResult)
{
    if (h == 0 || outResult == nullptr) return ommResult_INVALID_ARGUMENT;
    ShardedBake* sb = (ShardedBake*)h; ShardCtx& c = sb->ctx;
    if (c.counts.numOmms && gathered == nullptr) return sb->baker->log.invalid("[Invalid Argument] - gathered contributions missing");
    return guarded(&sb->baker->log, [&]() -> ommResult {
        const DeviceScope onBakersDevice(sb->baker->bind_device());
        const double t1 = now_ms();
        const ommResult r = sharded_finish(sb, [&](uint8_t* arrayData) {
            if (!arrayData) return false;   // (the result could not be allocated)
            // same chunk walk as the RCCL path (there each chunk arrives separately): a block that straddles a chunk boundary is placed in pieces
            const uint64_t chunkBytes = shard_chunk_bytes(*sb->baker, c.strideBytes);
            for (uint64_t lo = 0; lo < c.strideBytes; lo += chunkBytes) {
                const uint64_t hi = lo + chunkBytes < c.strideBytes ? lo + chunkBytes : c.strideBytes;
                launch_shard_scatter((const uint8_t*)gathered + lo, c.strideBytes, lo, hi, c.dActive, c.dOwner, c.dMask, c.dLevel, c.bits, c.to.order, c.dCofs, c.to.dstOfs, c.to.sizes,
                                     c.counts.numOmms, arrayData, sb->ses.stream);
            }
            return true;
        }, outResult);
        sb->tm.gatherMs = (float)(now_ms() - t1);
        return r;
    });
}

OMM_MI355X_API ommResult ommxShardedDestroy(ommxShardedBake h)
{
    if (h == 0) return ommResult_INVALID_ARGUMENT;
    ShardedBake* sb = (ShardedBake*)h;
    const Allocator mem = sb->mem;
    mem.destroy(sb);   // (the session waits for its streams before its working set returns to the baker's pool)
    return ommResult_SUCCESS;
}

// ---- RCCL communicator helpers + the one-call sharded bake ----
OMM_MI355X_API ommResult ommxRcclGetUniqueId(void* outId, size_t idBytes)
{
    if (outId == nullptr || idBytes < sizeof(RcclUniqueId)) return ommResult_INVALID_ARGUMENT;
    if (!rccl().ok()) return ommResult_FAILURE;
    return rccl().getUniqueId((RcclUniqueId*)outId) == 0 ? ommResult_SUCCESS : ommResult_FAILURE;
}

OMM_MI355X_API ommResult ommxRcclCommInitRank(const void* id, size_t idBytes, uint32_t rank, uint32_t worldSize, ommxRcclComm* outComm)
{
    if (id == nullptr || idBytes < sizeof(RcclUniqueId) || outComm == nullptr || worldSize == 0 || worldSize > (uint32_t)kMaxRanks || rank >= worldSize) return ommResult_INVALID_ARGUMENT;
    if (!rccl().ok()) return ommResult_FAILURE;
    RcclUniqueId uid; memcpy(&uid, id, sizeof uid);
    RcclComm* c = new (std::nothrow) RcclComm();
    if (!c) return ommResult_FAILURE;
    if (rccl().commInitRank(&c->comm, (int)worldSize, uid, (int)rank) != 0) { delete c; return ommResult_FAILURE; }
    c->owned = true; c->rank = (int)rank; c->world = (int)worldSize;
    rccl_prepare_status(c);
    *outComm = (ommxRcclComm)c;
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxRcclCommWrap(void* ncclComm, ommxRcclComm* outComm)
{
    if (ncclComm == nullptr || outComm == nullptr) return ommResult_INVALID_ARGUMENT;
    if (!rccl().ok()) return ommResult_FAILURE;
    RcclComm* c = new (std::nothrow) RcclComm();
    if (!c) return ommResult_FAILURE;
    c->comm = ncclComm; c->owned = false;
    if (rccl().commCount(c->comm, &c->world) != 0 || rccl().commUserRank(c->comm, &c->rank) != 0 || c->world < 1 || c->world > kMaxRanks) { delete c; return ommResult_INVALID_ARGUMENT; }
    rccl_prepare_status(c);
    *outComm = (ommxRcclComm)c;
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxCommFromCollectives(const ommxCollectives* collectives, uint32_t rank, uint32_t worldSize, ommxRcclComm* outComm)
{
    if (collectives == nullptr || collectives->allReduceU32 == nullptr || collectives->allGatherBytes == nullptr || outComm == nullptr ||
        worldSize == 0 || worldSize > (uint32_t)kMaxRanks || rank >= worldSize) return ommResult_INVALID_ARGUMENT;
    RcclComm* c = new (std::nothrow) RcclComm();
    if (!c) return ommResult_FAILURE;
    c->custom = true; c->user = *collectives; c->rank = (int)rank; c->world = (int)worldSize;
    rccl_prepare_status(c);
    *outComm = (ommxRcclComm)c;
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxRcclCommInfo(ommxRcclComm comm, uint32_t* outRank, uint32_t* outWorldSize)
{
    if (comm == 0 || outRank == nullptr || outWorldSize == nullptr) return ommResult_INVALID_ARGUMENT;
    RcclComm* c = (RcclComm*)comm;
    int rank = c->rank, world = c->world;
    if (!c->custom && c->comm && rccl().ok() && (rccl().commCount(c->comm, &world) != 0 || rccl().commUserRank(c->comm, &rank) != 0)) return ommResult_FAILURE;
    *outRank = (uint32_t)rank; *outWorldSize = (uint32_t)world;
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxRcclCommDestroy(ommxRcclComm comm)
{
    if (comm == 0) return ommResult_INVALID_ARGUMENT;
    RcclComm* c = (RcclComm*)comm;
    if (c->statusStream) { (void)hipStreamSynchronize(c->statusStream); (void)hipStreamDestroy(c->statusStream); }
    if (c->dStatus) (void)hipFree(c->dStatus);
    if (c->owned && c->comm) (void)rccl().commDestroy(c->comm);
    delete c;
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxShardedBakeRccl(ommBaker baker, const ommCpuBakeInputDesc* desc, ommxRcclComm comm, ommxDeviceBakeResult* outResult)
{
    Baker* b = nullptr;
    if (comm == 0 || outResult == nullptr) return ommResult_INVALID_ARGUMENT;
    const ommResult r0 = sharded_checks(baker, desc, &b, true);
    if (r0 != ommResult_SUCCESS) return r0;
    const Logger& L = b->log;
    RcclComm* rc = (RcclComm*)comm;
    if (!rc->custom && !rccl().ok()) return L.failure(("[Failure] - RCCL is not available: " + rccl().error).c_str());
    return guarded(&L, [&]() -> ommResult {
        const DeviceScope onBakersDevice(b->bind_device());
        rccl_status_on_device(rc, b->bind_device());          // (status words / idle-rank stream on the bake's device, whatever was current when the communicator was made)
        auto nccl_fail = [&](int code, const char* what) {
            char buf[256]; snprintf(buf, sizeof buf, "[Failure] - %s: %s", what, rc->error_string(code));
            return L.failure(buf);
        };
        const bool hostTail = wants_host_tail(*desc);
        ShardedBake* sb = nullptr;
        ommResult r = sharded_begin(b, desc, (uint32_t)rc->rank, (uint32_t)rc->world, true, &sb, hostTail);
        if (r != ommResult_SUCCESS) sb = nullptr;
        struct Owner { Baker* b; ShardedBake* sb; ~Owner() { if (sb) b->mem.destroy(sb); } } owner{ b, sb };
        if (!rccl_agree(rc, sb ? sb->ses.stream : nullptr, r == ommResult_SUCCESS, L, "classification of its share")) return r != ommResult_SUCCESS ? r : ommResult_FAILURE;
        ShardCtx& c = sb->ctx; hipStream_t stream = sb->ses.stream;
        const double t1 = now_ms();
        if (hostTail) {
            // opt-in lossy reducers (near-duplicate merge, maxArrayDataSize): the serial reference algorithms need the states of ALL work items.
            // Every rank classified its share into a zeroed buffer, so SUM all-reduces of the metadata words and of the packed states merge
            // them; each rank then runs the identical serial tail on the host and uploads the (identical) result.
            const size_t words = 4ull * c.hc.activeStart[kNumLevels], stateWords = (size_t)(c.hc.stateBytes / 4);
            int e = words ? rc->all_reduce(c.dMeta, c.dMeta, words, kRcclSum, stream) : 0;
            if (e == 0 && stateWords) e = rc->all_reduce(c.dStates, c.dStates, stateWords, kRcclSum, stream);
            if (e != 0) return nccl_fail(e, "ncclAllReduce of the micro-triangle states");
            launch_shard_unpack_meta(c.bounds, c.dActiveIds, c.hc.activeStart[kNumLevels], c.dMeta, c.dMask, (uint32_t*)c.ti.knownCount, c.ti.digests, c.dOwner, stream);
            std::vector<HostItem> items;
            r = gather_host_items(L, stream, c.ti.numItems, c.T, c.hc, c.ti.uv, c.dLevel, c.dActive, c.dMask, c.dStateOfs, c.dStates, c.ti.triToItem, c.bits, items);
            HostTailResult hres; ommIndexFormat ifmt = ommIndexFormat_UINT_32;
            if (r == ommResult_SUCCESS) r = run_host_tail_for(*desc, c.T, items, hres, ifmt);
            if (r != ommResult_SUCCESS) return r;
            DeviceBakeResult* res = b->mem.make<DeviceBakeResult>();
            if (!res) return ommResult_FAILURE;
            res->mem = b->mem; memset(&res->desc, 0, sizeof res->desc);
            r = upload_host_tail(*b, hres, ifmt, c.T, c.bits, stream, res);
            if (r != ommResult_SUCCESS) { b->mem.destroy(res); return r; }
            const int* ev = c.ev;
            sb->tm.setupMs = sb->et->ms(ev[0], ev[1]); sb->tm.triageMs = sb->et->ms(ev[1], ev[2]); sb->tm.classifyMs = sb->et->ms(ev[2], ev[3]);
            sb->tm.tailMs = (float)(now_ms() - t1); sb->tm.totalMs = (float)(now_ms() - sb->t0);
            { std::lock_guard<std::mutex> g(b->timingsMu); b->timings = sb->tm; b->haveTimings = true; }
            *outResult = (ommxDeviceBakeResult)res;
            return ommResult_SUCCESS;
        }
        // exchange 1: per-item metadata, SUM all-reduce in place, on the bake's own stream right behind the digests
        const size_t words = 4ull * c.hc.activeStart[kNumLevels];
        if (words) { const int e = rc->all_reduce(c.dMeta, c.dMeta, words, kRcclSum, stream); if (e != 0) return nccl_fail(e, "ncclAllReduce of the work-item metadata"); }
        r = sharded_tail(sb);
        if (!rccl_agree(rc, stream, r == ommResult_SUCCESS, L, "tail of the bake")) return r != ommResult_SUCCESS ? r : ommResult_FAILURE;
        sb->tm.tailMs = (float)(now_ms() - t1);
        const double t2 = now_ms();
        // exchange 2: the padded contributions, all-gathered in chunks on a second stream; every chunk is scattered to its final arrayData
        // offsets on the bake's stream while the next one is on the wire
        const uint32_t E = c.counts.numOmms;
        r = sharded_finish(sb, [&](uint8_t* arrayData) -> bool {
            // (a one-rank communicator takes the same route: the collectives degenerate to copies, the plumbing is the same)
            // (arrayData is null when this rank could not allocate the result: it still takes part in the agreement, so that nobody waits for it)
            hipEvent_t ready = nullptr, done[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
            bool ok = arrayData != nullptr && sb->ses.open_comm() && HIP_OK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
            // the contribution as a codec stream (tail_kernels.hip "block exchange codec"); the agreement on the ranks' status carries the stream sizes:
            // MAX over the ranks = what every rank sends (the all-gather needs equal counts), or "one of them does not shrink" = everybody sends raw
            ok = ok && HIP_OK(run_shard_compress(c.dContrib, c.strideBytes, c.dComp, c.compCap, c.dCompSize, c.dCodecScratch, c.codecScratchBytes, stream));
            uint32_t maxUnits = 0;
            ok = rccl_agree_max(rc, stream, ok, c.dCompSize, &maxUnits, L, "allocation of the result") && ok;
            if (!ok) { if (ready) (void)hipEventDestroy(ready); return false; }
            sb->tm.contributionBytes = c.strideBytes;
            if (maxUnits != kCodecIncompressible) {
                const size_t sendBytes = (size_t)maxUnits * 16u;   // (<= compCap by construction)
                sb->tm.exchangeBytes = sendBytes;
                const int e = rc->all_gather(c.dComp, c.dGatherComp, sendBytes, stream);
                if (e != 0) { (void)nccl_fail(e, "ncclAllGather of the OMM blocks"); (void)hipEventDestroy(ready); return false; }
                // every block straight from its owner's stream to its final offset (no expanded copy of the contributions)
                launch_shard_scatter_streams(c.dGatherComp, sendBytes, c.strideBytes, c.dActive, c.dOwner, c.dMask, c.dLevel, c.bits, c.to.order, c.dCofs, c.to.dstOfs, c.to.sizes, E, arrayData, stream);
                ok = HIP_OK(hipGetLastError()) && HIP_OK(hipStreamSynchronize(stream));
                (void)hipEventDestroy(ready);
                return ok;
            }
            sb->tm.exchangeBytes = c.strideBytes;
            hipStream_t cs = sb->ses.commStream;
            const uint64_t chunkBytes = shard_chunk_bytes(*sb->baker, c.strideBytes);              // per rank and chunk; at most 8 chunks
            uint64_t chunks = (c.strideBytes + chunkBytes - 1) / chunkBytes;
            ok = HIP_OK(hipEventRecord(ready, stream)) && HIP_OK(hipStreamWaitEvent(cs, ready, 0));
            int ncclErr = 0;
            for (uint64_t k = 0; ok && k < chunks; ++k) {
                const uint64_t lo = k * chunkBytes, hi = lo + chunkBytes < c.strideBytes ? lo + chunkBytes : c.strideBytes;
                if (lo >= hi) { chunks = k; break; }
                uint8_t* stage = c.dGathered + lo * (uint64_t)rc->world;               // chunk k of all ranks: world x (hi - lo) bytes
                ncclErr = rc->all_gather(c.dContrib + lo, stage, (size_t)(hi - lo), cs);
                ok = ncclErr == 0 && HIP_OK(hipEventCreateWithFlags(&done[k], hipEventDisableTiming)) && HIP_OK(hipEventRecord(done[k], cs)) && HIP_OK(hipStreamWaitEvent(stream, done[k], 0));
                if (ok) launch_shard_scatter(stage, hi - lo, lo, hi, c.dActive, c.dOwner, c.dMask, c.dLevel, c.bits, c.to.order, c.dCofs, c.to.dstOfs, c.to.sizes, E, arrayData, stream);
            }
            ok = ok && HIP_OK(hipStreamSynchronize(stream));
            (void)hipStreamSynchronize(cs);
            if (ready) (void)hipEventDestroy(ready);
            for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
            if (ncclErr != 0) (void)nccl_fail(ncclErr, "ncclAllGather of the OMM blocks");
            return ok;
        }, outResult);
        sb->tm.gatherMs = (float)(now_ms() - t2);
        if (r == ommResult_SUCCESS) { std::lock_guard<std::mutex> g(b->timingsMu); b->timings.tailMs = sb->tm.tailMs; b->timings.gatherMs = sb->tm.gatherMs; b->timings.exchangeBytes = sb->tm.exchangeBytes; b->timings.contributionBytes = sb->tm.contributionBytes; }
        return r;
    });
}

OMM_MI355X_API ommResult ommxGetDeviceBakeResultDesc(ommxDeviceBakeResult result, const ommCpuBakeResultDesc** desc)
{
    if (result == 0 || desc == nullptr) return ommResult_INVALID_ARGUMENT;
    *desc = &((DeviceBakeResult*)result)->desc;
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxDestroyDeviceBakeResult(ommxDeviceBakeResult result)
{
    if (result == 0) return ommResult_INVALID_ARGUMENT;
    DeviceBakeResult* r = (DeviceBakeResult*)result;
    const Allocator mem = r->mem;
    mem.destroy(r);
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxSetBakerKnob(ommBaker baker, ommxBakerKnob knob, uint64_t value)
{
    if (baker == 0 || tag_of(baker) != kCpuBaker || (unsigned)knob >= (unsigned)ommxBakerKnob_MAX_NUM) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_ShardChunkBytes && value != 0 && value < 256) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_StreamChunks && value > kMaxStreamRanges) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_GenericPass && value > 2) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_ResultTransfer && value > (uint64_t)ommxResultTransfer_Compressed) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_ExpandThreads && value > 64) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_Devices && value > (uint64_t)kMaxRanks) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_HelperAffinity && value > 1) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_ZeroAhead && value > 1) return ommResult_INVALID_ARGUMENT;
    if (knob == ommxBakerKnob_RetainMemory) {
        if (value > 1) return ommResult_INVALID_ARGUMENT;
        Baker* bk = untag<Baker>(baker);
        const bool keep = value == 0;
        bk->hostPool->retain.store(keep); bk->devPool->retain.store(keep); bk->arenas->retain.store(keep);
        if (!keep) { const DeviceScope onBakersDevice(bk->device.load()); bk->hostPool->trim(); bk->devPool->trim(0); bk->arenas->trim(); }
    }
    untag<Baker>(baker)->knobs[knob].store(value);
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxTrimBaker(ommBaker baker)
{
    if (baker == 0 || tag_of(baker) != kCpuBaker) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    const DeviceScope onBakersDevice(b->device.load());
    b->hostPool->trim(); b->devPool->trim(0); b->arenas->trim();
    return ommResult_SUCCESS;
}

OMM_MI355X_API ommResult ommxGetLastBakeTimingsSized(ommBaker baker, void* out, size_t outBytes, size_t* libraryBytes)
{
    if (libraryBytes) *libraryBytes = sizeof(ommxBakeTimings);
    if (baker == 0 || out == nullptr || tag_of(baker) != kCpuBaker) return ommResult_INVALID_ARGUMENT;
    Baker* b = untag<Baker>(baker);
    std::lock_guard<std::mutex> g(b->timingsMu);
    if (!b->haveTimings) return ommResult_FAILURE;
    // the struct only ever grows at its end: a caller built against an older header gets the prefix it knows, one built against a newer header zeros
    const size_t n = outBytes < sizeof(ommxBakeTimings) ? outBytes : sizeof(ommxBakeTimings);
    memcpy(out, &b->timings, n);
    if (outBytes > n) memset((uint8_t*)out + n, 0, outBytes - n);
    return ommResult_SUCCESS;
}
OMM_MI355X_API ommResult ommxGetLastBakeTimings(ommBaker baker, ommxBakeTimings* out)
{
    // The unsized getter is the round-3 symbol: binaries (and ctypes mirrors) built against that header hold a struct that ends in front of
    // streamPreviewMs, so this symbol never writes more than that prefix.  Everything newer is read through ommxGetLastBakeTimingsSized.
    return ommxGetLastBakeTimingsSized(baker, out, offsetof(ommxBakeTimings, streamPreviewMs), nullptr);
}

#include "serialize.inc"
