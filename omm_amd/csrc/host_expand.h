// host_expand.h -- host side of the compressed result of ommCpuBake: the finished arrayData crosses PCIe as a codec stream (tail_kernels.hip "block
// exchange codec": one nibble per 16-byte unit = which state it repeats, or "raw") and a few host threads expand it into the caller's array.
// Plain C++ (no HIP): compiled like host_tail.cpp.
#pragma once
#include <sched.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ommx {

// CPUs this process can run on at once: scheduler affinity capped by the cgroup CPU quota (cpu.max of cgroup v2, the cfs quota of v1); >= 1
unsigned effective_cpus();

// A baker's helper threads (ommCpuBakeFlags_EnableInternalThreads, omm.h:303: the reference spends that permission on its OpenMP loops).  Threads are
// started on first use and sleep between calls; run() hands out task indices 0 .. tasks-1 to the workers AND the calling thread and returns when all are done.
// One run() at a time (concurrent bakes on one baker take turns).  A thread that cannot be started is simply missing: the caller does the work itself.
class WorkerPool {
public:
    explicit WorkerPool(unsigned workers);
    ~WorkerPool();
    WorkerPool(const WorkerPool&) = delete; WorkerPool& operator=(const WorkerPool&) = delete;
    unsigned workers() const { return (unsigned)threads_.size(); }
    void run(uint32_t tasks, const std::function<void(uint32_t)>& fn);
    // The same tasks on the WORKERS only, in the background: start() returns at once (false: no workers -- nothing was started), wait() returns when every
    // task is done.  Between the two the pool is taken (run() and bind_near() of any thread wait for wait()); whoever owns the started run calls wait() exactly once (any
    // thread) and must not call run() / bind_near() before that.
    bool start(uint32_t tasks, std::function<void(uint32_t)> fn);
    void wait();
    // Moves the workers onto the NUMA node that holds `memory` (the array they are about to fill), each onto its own slice of the node's cores; the calling
    // thread's affinity is not touched.  Best effort: returns the node, or -1 when it could not be found / nothing was changed.
    int bind_near(const void* memory);
private:
    void loop();
    std::vector<std::thread> threads_;
    // one run() / start()..wait() / bind_near() at a time.  Not a plain mutex: wait() may be called by another thread than start() was
    struct Turn {
        std::mutex m; std::condition_variable cv; bool taken = false;
        void take() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !taken; }); taken = true; }
        void give() { { std::lock_guard<std::mutex> l(m); taken = false; } cv.notify_one(); }
    } turn_;
    struct TurnGuard { Turn& t; explicit TurnGuard(Turn& x) : t(x) { t.take(); } ~TurnGuard() { t.give(); } };
    std::mutex mu_; std::condition_variable wake_, done_;
    const std::function<void(uint32_t)>* fn_ = nullptr;
    std::function<void(uint32_t)> background_; bool backgroundActive_ = false;   // start() / wait()
    uint32_t tasks_ = 0; uint64_t generation_ = 0; unsigned active_ = 0; bool stop_ = false;
    std::atomic<uint32_t> next_{ 0 };
    int boundNode_ = -2;
    cpu_set_t boundMask_;   // the process's affinity mask when the workers were bound (bind_near binds again when it has changed)
};

// Layout of a codec stream (the same arithmetic as tail_kernels.hip: codec_layout): header 16 B | first raw unit of every 256-unit block (uint32, blocks + 1) |
// one nibble per unit (0..3: 16 bytes of 0x00 / 0x55 / 0xAA / 0xFF, 4: raw) | the raw units
struct HostCodecLayout { uint64_t units, blocks, offOfs, offCodes, offRaw; };
HostCodecLayout host_codec_layout(uint64_t paddedBytes);

// Expands codec blocks [b0, b1) of `stream` (layout L) into dst[0 .. dstBytes): block b covers dst bytes [4096 b, 4096 (b + 1)); bytes beyond dstBytes are not
// written (the device pads the array to a multiple of 256 bytes).  Write-only on dst (non-temporal stores when dst is 16-byte aligned).
// `zeroed` (optional): the destination was filled with zeros in pieces of 2 MiB while the device was baking -- piece j (bytes [j << 21, (j + 1) << 21) of dst)
// is complete when zeroed->done[j] is set; a codec block of 4 KiB that repeats state 0 and lies in a complete piece is not written again (a quarter of the
// blocks of the metric configuration).  *skipped (optional) += the bytes left as they were.
struct ZeroedPieces { const std::atomic<uint8_t>* done; size_t pieces; };
void codec_expand_blocks(uint8_t* dst, uint64_t dstBytes, const uint8_t* stream, const HostCodecLayout& L, uint64_t b0, uint64_t b1,
                         const ZeroedPieces* zeroed = nullptr, uint64_t* skipped = nullptr);
// zero bytes [lo, hi) of a 4 KiB-aligned block with non-temporal stores (the pre-fill itself)
void fill_zero_nt(uint8_t* dst, size_t lo, size_t hi);

// Multi-device ommCpuBake: every device hands its own surviving blocks to the host as ONE codec stream of its contribution (the blocks it owns, back to
// back in the order of the result; all contributions padded to the same size, hence one layout); the host writes every block of the result from its owner's
// stream -- the host form of tail_kernels.hip: shard_scatter_streams.  OMMs [j0, j1) of the result.
struct HostScatter {
    uint32_t world; const uint8_t* stream[16]; bool raw[16];   // raw[r]: rank r's contribution did not shrink and arrived as it is
    HostCodecLayout L;
    const uint8_t *active, *owner, *level; const uint32_t *stateMask, *order, *dstOfs, *sizes; const uint64_t* cofs; int bits;
    uint8_t* arrayData;
};
void codec_scatter_omms(const HostScatter& S, uint32_t j0, uint32_t j1);

} // namespace ommx
