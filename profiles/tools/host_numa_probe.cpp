// host_numa_probe.cpp -- why does the expansion of a compressed result run at 300 GB/s in one process and at 140 in the next?  One process = one sample:
// a pinned block like the baker's (hipHostMalloc), 12 threads filling it with non-temporal stores, unbound and bound to the block's NUMA node; prints the
// node of the block (move_pages), the CPUs the threads ran on, the times.   build: hipcc -O2 -o host_numa_probe host_numa_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <emmintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int node_of(void* p) { void* page = (void*)((uintptr_t)p & ~(uintptr_t)4095); int node = -99; long r = syscall(SYS_move_pages, 0, 1ul, &page, (const int*)nullptr, &node, 0); return r != 0 ? -100 : node; }
static bool cpus_of(int n, cpu_set_t* set) {
    CPU_ZERO(set); char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n);
    FILE* f = fopen(path, "r"); if (!f) return false; char list[4096] = { 0 }; const bool got = fgets(list, sizeof list, f) != nullptr; fclose(f); if (!got) return false;
    for (char* p = list; *p; ) { char* e = nullptr; long a = strtol(p, &e, 10); if (e == p) break; long b = a; p = e; if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; } for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, set); if (*p == ',') ++p; else break; }
    return CPU_COUNT(set) > 0;
}
int main()
{
    const size_t bytes = (size_t)1216 << 20; const unsigned nt = 12;
    uint8_t* pinned = nullptr;
    if (hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    printf("main thread on cpu %d; pinned block: node of first page %d, middle %d, last %d\n", sched_getcpu(), node_of(pinned), node_of(pinned + bytes / 2), node_of(pinned + bytes - 4096));
    for (int mode = 0; mode < 3; ++mode) {   // 0 unbound, 1 bound to the block's node, 2 bound to the other node
        const int node = node_of(pinned + bytes / 2);
        cpu_set_t set; bool have = false;
        if (mode && node >= 0) have = cpus_of(mode == 1 ? node : 1 - node, &set);
        for (int rep = 0; rep < 3; ++rep) {
            std::vector<int> cpus(nt, -1);
            const double t0 = now();
            std::vector<std::thread> th;
            for (unsigned p = 0; p < nt; ++p) th.emplace_back([&, p] {
                if (have) pthread_setaffinity_np(pthread_self(), sizeof set, &set);
                cpus[p] = sched_getcpu();
                const __m128i v = _mm_set1_epi32(0x55555555); uint8_t* d = pinned + bytes / nt * p / 16 * 16; const size_t n = bytes / nt / 16;
                for (size_t k = 0; k < n; ++k) _mm_stream_si128((__m128i*)(d + 16 * k), v);
                _mm_sfence();
            });
            for (auto& t : th) t.join();
            const double dt = now() - t0;
            printf("mode %d (%s) %.2f ms %.0f GB/s  cpus:", mode, mode == 0 ? "unbound" : (mode == 1 ? "block's node" : "other node"), dt, bytes / dt / 1e6);
            for (int c : cpus) printf(" %d", c); printf("\n");
        }
    }
    return 0;
}
