// expand_check.cpp -- host side of ommCpuBake's compressed result (omm_amd/csrc/host_expand.cpp) against a scalar model of the device codec
// (tail_kernels.hip: codec_code / shard_codec_write): random arrays with long constant runs, every size class incl. tails that are not a multiple of 16,
// unaligned destinations, worker pools of 0 .. 7 threads.  Prints "ok <cases>" or the first mismatch.   (CPU only; built and run by tests/test_host_expand.py)
#include "../../omm_amd/csrc/host_expand.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
using namespace ommx;
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main()
{
    int cases = 0;
    const size_t sizes[] = { 1, 15, 16, 17, 255, 256, 4095, 4096, 4097, 65536 + 3, (1u << 20) + 777, (5u << 20) + 16 };
    for (size_t bytes : sizes) for (int variant = 0; variant < 4; ++variant) for (unsigned workers : { 0u, 3u, 7u }) {
        const size_t padded = (bytes + 255) & ~(size_t)255;
        std::vector<uint8_t> src(padded, 0);
        // runs of a repeated state byte with raw noise in between (variant 3: all noise = incompressible, variant 0: one state only)
        static const uint8_t pat[4] = { 0x00, 0x55, 0xAA, 0xFF };
        for (size_t o = 0; o < padded; ) {
            size_t len = variant == 0 ? padded : (variant == 3 ? 16 : 16 * (1 + rnd() % (variant == 1 ? 2000 : 40)));
            if (o + len > padded) len = padded - o;
            if (variant == 3 || (variant != 0 && rnd() % 8 == 0)) for (size_t k = 0; k < len; ++k) src[o + k] = (uint8_t)rnd();
            else memset(&src[o], pat[rnd() % 4], len);
            o += len;
        }
        const HostCodecLayout L = host_codec_layout(padded);
        std::vector<uint8_t> stream(L.offRaw + 16 * L.units + 64, 0xEE);
        uint32_t* ofs = (uint32_t*)(stream.data() + L.offOfs); uint32_t nraw = 0;
        for (uint64_t u = 0; u < L.units; ++u) {
            if (u % 256 == 0) ofs[u / 256] = nraw;
            uint32_t w[4]; memcpy(w, &src[u * 16], 16);
            const bool same = w[0] == w[1] && w[0] == w[2] && w[0] == w[3];
            const uint32_t code = !same ? 4u : (w[0] == 0u ? 0u : (w[0] == 0x55555555u ? 1u : (w[0] == 0xAAAAAAAAu ? 2u : (w[0] == 0xFFFFFFFFu ? 3u : 4u))));
            uint8_t& c = stream[L.offCodes + u / 2];
            c = (uint8_t)((u & 1) ? ((c & 0x0F) | (code << 4)) : code);
            if (code == 4u) { memcpy(&stream[L.offRaw + 16ull * nraw], w, 16); ++nraw; }
        }
        ofs[L.blocks] = nraw;
        for (int misalign = 0; misalign < 2; ++misalign) {
            std::vector<uint8_t> out(bytes + 64 + 16, 0xCD);
            uint8_t* base = out.data(); while (((uintptr_t)base & 15u) != (misalign ? 4u : 0u)) ++base;
            WorkerPool pool(workers);
            const uint32_t tasks = (uint32_t)((L.blocks + 63) / 64);
            pool.run(tasks, [&](uint32_t t) { const uint64_t b0 = (uint64_t)t * 64, b1 = b0 + 64 < L.blocks ? b0 + 64 : L.blocks; codec_expand_blocks(base, bytes, stream.data(), L, b0, b1); });
            if (memcmp(base, src.data(), bytes) != 0) { printf("MISMATCH bytes=%zu variant=%d workers=%u misalign=%d\n", bytes, variant, workers, misalign); return 1; }
            for (int k = 0; k < 16; ++k) if (base[bytes + k] != 0xCD) { printf("OVERRUN bytes=%zu variant=%d at +%d\n", bytes, variant, k); return 1; }
            ++cases;
        }
    }
    // ---- zeroing ahead (round 6): the destination holds STALE bytes (the previous bake's result); helper threads zero it in pieces of 2 MiB in the background
    //      (WorkerPool::start / wait), some pieces are cancelled; the expansion may leave a block of zeros alone only inside a piece that was completed ----
    int zeroCases = 0;
    for (int variant = 1; variant <= 2; ++variant) for (unsigned workers : { 1u, 3u, 7u }) for (int cancelAt : { -1, 0, 2 }) {
        const size_t bytes = (9u << 20) + 4096 * 3 + 48, padded = (bytes + 255) & ~(size_t)255;
        std::vector<uint8_t> src(padded, 0);
        static const uint8_t pat[4] = { 0x00, 0x55, 0xAA, 0xFF };
        for (size_t o = 0; o < padded; ) {   // long runs (whole 4 KiB blocks of one state, many of them zeros) with noise in between
            size_t len = 16 * (1 + rnd() % (variant == 1 ? 4000 : 600));
            if (o + len > padded) len = padded - o;
            if (rnd() % 10 == 0) for (size_t k = 0; k < len; ++k) src[o + k] = (uint8_t)rnd(); else memset(&src[o], pat[rnd() % 2 ? 0 : rnd() % 4], len);
            o += len;
        }
        const HostCodecLayout L = host_codec_layout(padded);
        std::vector<uint8_t> stream(L.offRaw + 16 * L.units + 64, 0xEE);
        uint32_t* ofs = (uint32_t*)(stream.data() + L.offOfs); uint32_t nraw = 0;
        for (uint64_t u = 0; u < L.units; ++u) {
            if (u % 256 == 0) ofs[u / 256] = nraw;
            uint32_t w[4]; memcpy(w, &src[u * 16], 16);
            const bool same = w[0] == w[1] && w[0] == w[2] && w[0] == w[3];
            const uint32_t code = !same ? 4u : (w[0] == 0u ? 0u : (w[0] == 0x55555555u ? 1u : (w[0] == 0xAAAAAAAAu ? 2u : (w[0] == 0xFFFFFFFFu ? 3u : 4u))));
            uint8_t& c = stream[L.offCodes + u / 2];
            c = (uint8_t)((u & 1) ? ((c & 0x0F) | (code << 4)) : code);
            if (code == 4u) { memcpy(&stream[L.offRaw + 16ull * nraw], w, 16); ++nraw; }
        }
        ofs[L.blocks] = nraw;
        std::vector<uint8_t> out(bytes + 8192, 0x77);   // stale content everywhere
        uint8_t* base = out.data(); while (((uintptr_t)base & 4095u) != 0u) ++base;
        const size_t pieces = (bytes + (2u << 20) - 1) >> 21;
        std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[pieces]);
        for (size_t j = 0; j < pieces; ++j) done[j].store(0);
        WorkerPool pool(workers);
        std::atomic<bool> cancel{ false };
        const bool started = pool.start((uint32_t)pieces, [&](uint32_t j) {
            if (cancel.load() || (int)j == cancelAt) return;   // (a cancelled / skipped piece keeps its stale bytes and stays unmarked)
            const size_t lo = (size_t)j << 21, hi = lo + (2u << 20) < bytes ? lo + (2u << 20) : bytes;
            fill_zero_nt(base, lo, hi);
            done[j].store(1, std::memory_order_release);
        });
        if (!started) { printf("START FAILED workers=%u\n", workers); return 1; }
        pool.wait();
        const ZeroedPieces z{ done.get(), pieces };
        std::atomic<uint64_t> skipped{ 0 };
        const uint32_t tasks = (uint32_t)((L.blocks + 63) / 64);
        pool.run(tasks, [&](uint32_t t) { const uint64_t b0 = (uint64_t)t * 64, b1 = b0 + 64 < L.blocks ? b0 + 64 : L.blocks; uint64_t sk = 0; codec_expand_blocks(base, bytes, stream.data(), L, b0, b1, &z, &sk); skipped += sk; });
        if (memcmp(base, src.data(), bytes) != 0) { printf("ZERO-AHEAD MISMATCH variant=%d workers=%u cancelAt=%d\n", variant, workers, cancelAt); return 1; }
        if (skipped.load() == 0) { printf("ZERO-AHEAD skipped nothing variant=%d\n", variant); return 1; }
        ++zeroCases;
    }
    // ---- codec_scatter_omms: blocks of a result from the codec streams of their owners' contributions (multi-device ommCpuBake) ----
    int scatterCases = 0;
    for (uint32_t world : { 1u, 2u, 3u, 8u }) for (int bits = 1; bits <= 2; ++bits) for (int rep = 0; rep < 3; ++rep) {
        // OMMs in descending level order (sizes 4^level * bits / 8, at least 1 byte), like the result's; items: some uniform (no stored states)
        std::vector<uint32_t> order, dstOfs, sizes, stateMask; std::vector<uint64_t> cofs; std::vector<uint8_t> active, owner, level;
        std::vector<uint64_t> fill(world, 0);
        uint64_t total = 0;
        for (int lv = 8; lv >= 0; --lv) {
            const uint32_t count = lv >= 7 ? 3 + rnd() % 6 : 5 + rnd() % 40;
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t item = (uint32_t)active.size(), bytes = ((1u << (2 * lv)) * (uint32_t)bits) >> 3 ? ((1u << (2 * lv)) * (uint32_t)bits) >> 3 : 1u;
                const bool uni = rnd() % 7 == 0;
                active.push_back(uni ? 0 : 1); level.push_back((uint8_t)lv); stateMask.push_back(1u << (rnd() % (bits == 2 ? 4 : 2)));
                const uint32_t r = (uint32_t)(rnd() % world); owner.push_back(uni ? 0xFF : (uint8_t)r);
                order.push_back(item); dstOfs.push_back((uint32_t)total); sizes.push_back(bytes); cofs.push_back(uni ? 0 : fill[r]);
                if (!uni) fill[r] += bytes;
                total += bytes;
            }
        }
        uint64_t stride = 0; for (uint32_t r = 0; r < world; ++r) stride = fill[r] > stride ? fill[r] : stride;
        stride = (stride + 255) & ~255ull; if (!stride) stride = 256;
        // expected result + the contributions
        std::vector<uint8_t> want(total, 0); std::vector<std::vector<uint8_t>> contrib(world, std::vector<uint8_t>(stride, 0));
        static const uint8_t pat[4] = { 0x00, 0x55, 0xAA, 0xFF };
        for (size_t j = 0; j < order.size(); ++j) {
            uint8_t* d = &want[dstOfs[j]];
            if (!active[j]) { uint32_t st = 0; for (uint32_t m = stateMask[j]; m > 1; m >>= 1) ++st; uint32_t used = (1u << (2 * level[j])) * bits; if (used > 8) used = 8; uint32_t p = 0; for (uint32_t b = 0; b < used; b += bits) p |= st << b; memset(d, (int)p, sizes[j]); continue; }
            for (uint32_t o = 0; o < sizes[j]; ) { uint32_t len = 16 * (1 + rnd() % 64); if (o + len > sizes[j]) len = sizes[j] - o; if (rnd() % 5 == 0) for (uint32_t k = 0; k < len; ++k) d[o + k] = (uint8_t)rnd(); else memset(d + o, pat[rnd() % 4], len); o += len; }
            memcpy(&contrib[owner[j]][cofs[j]], d, sizes[j]);
        }
        const HostCodecLayout L = host_codec_layout(stride);
        std::vector<std::vector<uint8_t>> streams(world);
        HostScatter S; memset(&S, 0, sizeof S);
        S.world = world; S.L = L; S.bits = bits;
        for (uint32_t r = 0; r < world; ++r) {
            S.raw[r] = rep == 2 && r == 0;   // (one rank's contribution travels uncompressed)
            if (S.raw[r]) { S.stream[r] = contrib[r].data(); continue; }
            streams[r].assign(L.offRaw + 16 * L.units + 64, 0xEE);
            uint32_t* ofs = (uint32_t*)(streams[r].data() + L.offOfs); uint32_t nraw = 0;
            for (uint64_t u = 0; u < L.units; ++u) {
                if (u % 256 == 0) ofs[u / 256] = nraw;
                uint32_t w[4]; memcpy(w, &contrib[r][u * 16], 16);
                const bool same = w[0] == w[1] && w[0] == w[2] && w[0] == w[3];
                const uint32_t code = !same ? 4u : (w[0] == 0u ? 0u : (w[0] == 0x55555555u ? 1u : (w[0] == 0xAAAAAAAAu ? 2u : (w[0] == 0xFFFFFFFFu ? 3u : 4u))));
                uint8_t& c = streams[r][L.offCodes + u / 2];
                c = (uint8_t)((u & 1) ? ((c & 0x0F) | (code << 4)) : code);
                if (code == 4u) { memcpy(&streams[r][L.offRaw + 16ull * nraw], w, 16); ++nraw; }
            }
            ofs[L.blocks] = nraw; S.stream[r] = streams[r].data();
        }
        std::vector<uint8_t> got(total + 64, 0xCD);
        S.active = active.data(); S.owner = owner.data(); S.level = level.data(); S.stateMask = stateMask.data(); S.order = order.data(); S.dstOfs = dstOfs.data();
        S.sizes = sizes.data(); S.cofs = cofs.data(); S.arrayData = got.data();
        WorkerPool pool(3);
        const uint32_t E = (uint32_t)order.size(), tasks = (E + 15) / 16;
        pool.run(tasks, [&](uint32_t t) { codec_scatter_omms(S, t * 16, t * 16 + 16 < E ? t * 16 + 16 : E); });
        if (memcmp(got.data(), want.data(), total) != 0) { printf("SCATTER MISMATCH world=%u bits=%d rep=%d\n", world, bits, rep); return 1; }
        for (int k = 0; k < 64; ++k) if (got[total + k] != 0xCD) { printf("SCATTER OVERRUN world=%u\n", world); return 1; }
        ++scatterCases;
    }
    printf("ok %d scatter %d zero-ahead %d effective_cpus %u\n", cases, scatterCases, zeroCases, effective_cpus());
    return 0;
}
