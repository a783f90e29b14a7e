// host_tail.cpp -- the reference's serial tail for the two opt-in, lossy size reducers:
//   * near-duplicate merging  (ommCpuBakeFlags_EnableNearDuplicateDetection: LSH by Hamming bit sampling, bake_cpu_impl.cpp:1068-1352;
//                              internal brute-force variant :1354-1430)
//   * array-size budget       (maxArrayDataSize: greedy one-level down-sampling, :1474-1688)
// Both are sequential greedy algorithms over whole OMMs (std::mt19937, std::sort, libm powf/logf) and stay on the host, as
// in the reference; they run on the per-micro-triangle states the HIP kernels produced, in the
// device's own 2-bit packed form (Hamming distances by XOR + population count), uniform work items as (state, level).
// The default bake never comes here: its tail is tail_kernels.hip.
#include "host_tail.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <limits>
#include <random>
#include <set>
#include <unordered_map>
#include <xmmintrin.h>
#include <emmintrin.h>

namespace ommx {

namespace {
inline uint32_t f2u(float f) { return (uint32_t)_mm_cvttss_si64(_mm_set_ss(f)); }
inline int f2i(float f) { return _mm_cvtt_ss2si(_mm_set_ss(f)); }
inline bool is_known(uint8_t s) { return s < 2; }
inline bool is_unknown(uint8_t s) { return s >= 2; }
inline uint8_t three(uint8_t s) { return s == 2 ? 3 : s; } // OmmArrayDataView::SetState (bake_cpu_impl.cpp:374-377)

// XXH64 (xxHash spec) -- digest of the 3-state byte stream, seed 42
const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rnd(uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; }
inline uint64_t mrg(uint64_t h, uint64_t v) { h ^= rnd(0, v); return h * P1 + P4; }
uint64_t xxh64(const uint8_t* p, size_t len, uint64_t seed)
{
    const uint8_t* end = p + len; uint64_t h;
    auto rd64 = [](const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return v; };
    auto rd32 = [](const uint8_t* q) { uint32_t v; memcpy(&v, q, 4); return v; };
    if (len >= 32) {
        uint64_t a = seed + P1 + P2, b = seed + P2, c = seed, d = seed - P1;
        do { a = rnd(a, rd64(p)); b = rnd(b, rd64(p + 8)); c = rnd(c, rd64(p + 16)); d = rnd(d, rd64(p + 24)); p += 32; } while (p + 32 <= end);
        h = rotl(a, 1) + rotl(b, 7) + rotl(c, 12) + rotl(d, 18);
        h = mrg(h, a); h = mrg(h, b); h = mrg(h, c); h = mrg(h, d);
    } else h = seed + P5;
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= rnd(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p) * P5; h = rotl(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}


// ---- state storage of a work item (host_tail.h): uniform items are (state, level); the others hold 2 bits per micro-triangle, LSB first -- the layout of
//      the device's packed states and of a 4-state arrayData block.  (The reference keeps one BYTE per micro-triangle of every item, bake_cpu_impl.cpp:401-411:
//      65 GB at the metric configuration; this form needs 2 GB there.) ----
inline size_t num_micro(const HostItem& it) { return (size_t)1 << (2 * it.level); }
inline uint8_t get_state(const HostItem& it, size_t u) { return it.uniform >= 0 ? (uint8_t)it.uniform : (uint8_t)((it.packed[u >> 2] >> ((u & 3) << 1)) & 3u); }
inline void set_state(HostItem& it, size_t u, uint8_t v) { uint8_t& b = it.packed[u >> 2]; const int sh = (int)((u & 3) << 1); b = (uint8_t)((b & ~(3u << sh)) | ((uint32_t)v << sh)); }
void materialize(HostItem& it)   // uniform -> explicit states (before a merge writes into it)
{
    if (it.uniform < 0) return;
    const size_t n = num_micro(it);
    it.packed.assign(n >= 4 ? n / 4 : 1, (uint8_t)((0x55u * (uint32_t)it.uniform) & (n >= 4 ? 0xFFu : 0x03u)));   // (level 0: one field, the rest of the byte stays clear)
    it.uniform = -1;
}
inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint64_t fold3(uint64_t w) { return w | ((w >> 1) & 0x5555555555555555ull); }   // per 2-bit field: UT (2) -> UO (3), as three()
// number of micro-triangles of a (non-uniform) item whose state is T or O
size_t count_known(const HostItem& it)
{
    const size_t n = num_micro(it);
    if (n < 32) { size_t k = 0; for (size_t u = 0; u < n; ++u) k += is_known(get_state(it, u)); return k; }
    size_t unknown = 0;
    for (size_t o = 0; o < n / 4; o += 8) unknown += (size_t)__builtin_popcountll(load64(it.packed.data() + o) & 0xAAAAAAAAAAAAAAAAull);
    return n - unknown;
}

// bake_cpu_impl.cpp:1432-1472
void promote(const HostTailDesc& d, std::vector<HostItem>& items)
{
    for (HostItem& it : items) {
        if (it.special != 0) continue;
        const size_t n = num_micro(it);
        bool allEqual = true; uint8_t common = get_state(it, 0);
        if (it.uniform < 0) {
            if (n < 32) { for (size_t u = 1; u < n; ++u) allEqual &= common == get_state(it, u); }
            else { const uint64_t pat = 0x5555555555555555ull * common; for (size_t o = 0; o < n / 4 && allEqual; o += 8) allEqual = load64(it.packed.data() + o) == pat; }
        }
        if (!allEqual && d.rejectionThreshold > 0.f) {
            const uint32_t known = (uint32_t)count_known(it);
            const float frac = known / (float)(uint32_t)n;
            if (frac < d.rejectionThreshold) { allEqual = true; common = 2; }
        }
        if (allEqual && !d.disableSpecial) it.special = -(int32_t)common - 1;
    }
}

// bake_cpu_impl.cpp:1031-1066
void dedup_exact(const HostTailDesc& d, std::vector<HostItem>& items)
{
    if (d.disableDedup) return;
    std::unordered_map<uint64_t, uint32_t> seen;
    seen.reserve(items.size() * 2);
    std::vector<uint8_t> tmp;
    uint64_t uniformDigest[13][4]; bool haveUniform[13][4]; memset(haveUniform, 0, sizeof haveUniform);
    for (uint32_t i = 0; i < items.size(); ++i) {
        HostItem& it = items[i];
        const size_t n = num_micro(it);
        uint64_t dg;
        if (it.uniform >= 0) {   // XXH64 of 4^level equal bytes: once per (level, state)
            const uint8_t s3 = three((uint8_t)it.uniform);
            if (!haveUniform[it.level][s3]) { tmp.assign(n, s3); uniformDigest[it.level][s3] = xxh64(tmp.data(), n, 42); haveUniform[it.level][s3] = true; }
            dg = uniformDigest[it.level][s3];
        } else {
            tmp.resize(n);
            for (size_t u = 0; u < n; ++u) tmp[u] = three(get_state(it, u));
            dg = xxh64(tmp.data(), n, 42);
        }
        auto f = seen.find(dg);
        if (f == seen.end()) seen.emplace(dg, i);
        else {
            HostItem& ex = items[f->second];
            ex.prims.insert(ex.prims.end(), it.prims.begin(), it.prims.end());
            it.prims.clear(); it.special = -1;
        }
    }
}

float hamming3(const HostItem& a, const HostItem& b) // :1068-1083 (XOR of the folded 2-bit fields, population count of the fields that differ)
{
    const size_t n = num_micro(a);
    uint32_t diff = 0;
    if (n < 32 || a.uniform >= 0 || b.uniform >= 0) { for (size_t u = 0; u < n; ++u) if (three(get_state(a, u)) != three(get_state(b, u))) diff++; return float(diff); }
    for (size_t o = 0; o < n / 4; o += 8) {
        const uint64_t x = fold3(load64(a.packed.data() + o)) ^ fold3(load64(b.packed.data() + o));
        diff += (uint32_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
    return float(diff);
}

void merge_items(HostItem& to, HostItem& from) // :1093-1132
{
    to.prims.insert(to.prims.end(), from.prims.begin(), from.prims.end());
    from.prims.clear(); from.special = -1;
    materialize(to);
    const size_t n = num_micro(from);
    for (size_t u = 0; u < n; ++u) {
        const uint8_t ts = get_state(to, u), fs = get_state(from, u);
        if (ts != fs) {
            if (is_known(fs) && is_known(ts)) set_state(to, u, 3);
            else if (is_known(ts) && is_unknown(fs)) set_state(to, u, fs);
        }
    }
}

// bake_cpu_impl.cpp:1134-1352
void dedup_lsh(const HostTailDesc& d, std::vector<HostItem>& items, uint32_t iterations)
{
    if (d.disableDedup || !d.nearDup || d.nearDupBrute) return;
    std::mt19937 mt(42);
    const size_t N = items.size();
    std::vector<uint32_t> batch, bits, samples; std::vector<uint64_t> hashes;
    for (uint32_t attempt = 0; attempt < iterations; ++attempt) {
        for (uint32_t lvl = 1; lvl <= 12; ++lvl) {
            batch.clear();
            for (uint32_t i = 0; i < N; ++i) { const HostItem& it = items[i]; if (it.special == 0 && it.format == 2 && it.level == lvl) batch.push_back(i); }
            if (batch.empty()) continue;
            const uint32_t numMicro = 1u << (2 * lvl), n = (uint32_t)batch.size(), dd = numMicro;
            const float r = d.nearDupFactor * dd;
            const float c = 4.0f, p = 1.f / c;
            const uint32_t L = f2u(ceilf(powf((float)n, p)));
            if (L == 0) continue;
            const uint32_t k = f2u(ceilf((logf((float)n) * dd) / (c * r)));
            if (k == 0) continue;
            // (hashes are indexed by position in the batch, not by item: the reference's L x N table would be 8 bytes x L x all items)
            bits.resize((size_t)L * k); hashes.assign((size_t)L * n, 0); samples.resize(k);
            for (uint32_t l = 0; l < L; ++l) for (uint32_t j = 0; j < k; ++j) bits[(size_t)l * k + j] = (uint32_t)mt() & (numMicro - 1);
            // bucket lists keep insertion order (std::vector push_back in batch order)
            std::vector<std::unordered_map<uint64_t, std::vector<uint32_t>>> buckets(L);
            for (uint32_t bi = 0; bi < n; ++bi) {
                const uint32_t wi = batch[bi];
                const HostItem& it = items[wi];
                for (uint32_t l = 0; l < L; ++l) {
                    for (uint32_t j = 0; j < k; ++j) samples[j] = three(get_state(it, bits[(size_t)l * k + j]));
                    const uint64_t h = xxh64((const uint8_t*)samples.data(), sizeof(uint32_t) * k, 42);
                    hashes[(size_t)l * n + bi] = h;
                    buckets[l][h].push_back(wi);
                }
            }
            std::set<uint32_t> potential;
            for (uint32_t bi = 0; bi < n; ++bi) {
                const uint32_t wi = batch[bi];
                HostItem& it = items[wi];
                if (it.special != 0) continue;
                potential.clear();
                for (uint32_t l = 0; l < L; ++l) {
                    const auto& lst = buckets[l][hashes[(size_t)l * n + bi]];
                    for (uint32_t cand : lst) {
                        if (cand == wi) continue;
                        if (items[cand].special != 0) continue;
                        if (potential.size() > 3 * (size_t)L) break;
                        potential.insert(cand);
                    }
                }
                float minDist = std::numeric_limits<float>::max(); int32_t nearest = -1;
                for (uint32_t cand : potential) {
                    const float dist = hamming3(it, items[cand]);
                    if (dist < r && dist < minDist) { minDist = dist; nearest = (int32_t)cand; }
                }
                if (nearest >= 0) merge_items(it, items[nearest]);
            }
        }
    }
}

// bake_cpu_impl.cpp:1354-1430
void dedup_brute(const HostTailDesc& d, std::vector<HostItem>& items)
{
    if (d.disableDedup || !d.nearDup || !d.nearDupBrute || items.empty()) return;
    std::set<uint32_t> merged;
    for (uint32_t a = 0; a + 1 < items.size(); ++a) {
        HostItem& A = items[a];
        if (A.special != 0 || A.format != 2) continue;
        const uint32_t start = a + 1, end = std::min<uint32_t>(2048 + start, (uint32_t)items.size());
        float minDist = std::numeric_limits<float>::max(); int32_t nearest = -1;
        for (uint32_t bI = start; bI < end; ++bI) {
            const HostItem& B = items[bI];
            if (B.special != 0 || B.format != 2 || B.prims.empty() || A.level != B.level || merged.count(bI)) continue;
            const float dist = hamming3(A, B) / (uint32_t)num_micro(A);
            if (dist < 0.1f && dist < minDist) { minDist = dist; nearest = (int32_t)bI; }
        }
        if (nearest >= 0) { merged.insert(a); merged.insert((uint32_t)nearest); merge_items(A, items[nearest]); }
    }
}

// ---- Compress, bake_cpu_impl.cpp:1474-1688 ----
struct Info { float knownRatio = 0, knownRatioDown = 0, totalArea = 0, coveragePerByte = 0; size_t mem = 0, memDown = 0; };
float area2d(const float* p) // util/geometry.h:141-145
{
    const float v0x = p[4] - p[0], v0y = p[5] - p[1], v1x = p[2] - p[0], v1y = p[3] - p[1];
    const float nx = v0y * 0.f - v1y * 0.f, ny = 0.f * v1x - 0.f * v0x, nz = v0x * v1y - v1x * v0y;
    return 0.5f * sqrtf(nx * nx + ny * ny + nz * nz);
}
// the state of a parent micro-triangle from its four children (one byte of the packed states), :1499-1529
inline uint8_t parent_state(uint8_t childByte)
{
    const uint8_t f = (uint8_t)(childByte | ((childByte >> 1) & 0x55u));   // three() per field
    return f == 0x00u ? 0 : (f == 0x55u ? 1 : 3);
}
void downsample(HostItem& it) // :1499-1529
{
    it.level -= 1;
    if (it.uniform >= 0) { it.uniform = is_known(three((uint8_t)it.uniform)) ? three((uint8_t)it.uniform) : 3; return; }
    const size_t n = num_micro(it);
    std::vector<uint8_t> next(n >= 4 ? n / 4 : 1, 0);
    for (size_t i = 0; i < n; ++i) next[i >> 2] = (uint8_t)(next[i >> 2] | (parent_state(it.packed[i]) << ((i & 3) << 1)));
    it.packed.swap(next);
}
void compute_info(const HostItem& it, Info& o) // :1572-1595
{
    const uint32_t total = (uint32_t)num_micro(it);
    const size_t nd = (size_t)1 << (2 * (it.level - 1));
    uint32_t known, kd = 0;
    if (it.uniform >= 0) { known = is_known((uint8_t)it.uniform) ? total : 0; kd = is_known((uint8_t)it.uniform) ? (uint32_t)nd : 0; }
    else {
        known = (uint32_t)count_known(it);
        for (size_t i = 0; i < nd; ++i) kd += parent_state(it.packed[i]) != 3;
    }
    o.knownRatio = (float)known / total;
    o.knownRatioDown = kd / (float)nd;
    o.totalArea = 0;
    for (size_t k = 0; k < it.prims.size(); ++k) o.totalArea += area2d(it.uv);
    o.mem = std::max<size_t>(1, ((size_t)total * 2) / 8);
    o.memDown = std::max<size_t>(1, (nd * 2) / 8);
    const size_t memDelta = o.mem - o.memDown;
    const float covDelta = o.knownRatio - o.knownRatioDown;
    o.coveragePerByte = o.totalArea * covDelta / memDelta;
}
int compress(const HostTailDesc& d, std::vector<HostItem>& items)
{
    if (d.maxArrayDataSize == 0xFFFFFFFFu) return 0;
    std::vector<std::pair<int, Info>> act;
    for (int i = 0; i < (int)items.size(); ++i) {
        const HostItem& it = items[i];
        if (it.level == 0 || it.prims.empty() || it.special != 0) continue;
        Info inf; compute_info(it, inf); act.push_back(std::make_pair(i, inf));
    }
    size_t total = 0; for (const auto& a : act) total += a.second.mem;
    if (total < d.maxArrayDataSize) return 0;
    auto cmp = [](const std::pair<int, Info>& a, const std::pair<int, Info>& b) { return a.second.coveragePerByte < b.second.coveragePerByte; };
    std::sort(act.begin(), act.end(), cmp);
    while (total >= d.maxArrayDataSize && !act.empty()) {
        const int Nn = (int)act.size();
        for (int i = 0; i < Nn; ++i) {
            HostItem& it = items[act[i].first];
            total -= act[i].second.mem;
            if (it.level == 0) return 1;
            downsample(it);
            total += act[i].second.memDown;
            if (it.level == 0) { act[i].first = -1; continue; }
            compute_info(it, act[i].second);
            if (total < d.maxArrayDataSize) break;
            if (i + 1 != Nn && act[i].second.coveragePerByte < act[i + 1].second.coveragePerByte) i--;
        }
        for (int i = 0; i < (int)act.size(); ++i)
            if (act[i].first == -1) { std::swap(act[i], act[act.size() - 1]); act.pop_back(); i--; }
        std::sort(act.begin(), act.end(), cmp);
    }
    return 0;
}

uint32_t spread16(uint32_t x)
{
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x;
}
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (hi < v ? hi : v); }
} // namespace

// bake_cpu_impl.cpp:1957-1981 from the first PromoteToSpecialIndices on
int run_host_tail(const HostTailDesc& d, std::vector<HostItem>& items, HostTailResult& out)
{
    promote(d, items);
    dedup_exact(d, items);
    dedup_lsh(d, items, 3);
    dedup_brute(d, items);
    promote(d, items);
    if (compress(d, items)) return 1;
    dedup_exact(d, items);
    promote(d, items);

    // CreateUsageHistograms (:1690-1705)
    uint32_t arrH[3][13], idxH[3][13]; memset(arrH, 0, sizeof arrH); memset(idxH, 0, sizeof idxH);
    for (const HostItem& it : items) if (it.special == 0) { arrH[it.format][it.level] += 1; idxH[it.format][it.level] += (uint32_t)it.prims.size(); }
    // MicromapSpatialSort (:1707-1754)
    std::vector<std::pair<uint64_t, uint32_t>> keys(items.size());
    for (uint32_t i = 0; i < items.size(); ++i) {
        const HostItem& it = items[i];
        uint64_t key;
        if (it.special != 0) key = (1ull << 63) | (uint64_t)i;
        else {
            const float* p = it.uv;
            const float cx = (p[0] + p[2] + p[4]) / 3.f, cy = (p[1] + p[3] + p[5]) / 3.f;
            const int qx = f2i(8192.f * cx), qy = f2i(8192.f * cy);
            const int mx = clampi(f2i(fabsf((float)qx + 0.5f)), 0, 8191), my = clampi(f2i(fabsf((float)qy + 0.5f)), 0, 8191);
            key = ((uint64_t)it.level << 60) | (uint64_t)(spread16((uint32_t)mx) | (spread16((uint32_t)my) << 1));
        }
        keys[i] = std::make_pair(key, i);
    }
    std::sort(keys.begin(), keys.end(), std::greater<std::pair<uint64_t, uint32_t>>());
    // Serialize (:1756-1920)
    const uint32_t bitCount = (uint32_t)d.format;
    uint32_t descCount = 0; size_t dataSize = 0;
    for (uint32_t l = 0; l < 13; ++l) {
        const uint32_t cnt = arrH[d.format][l];
        descCount += cnt;
        const size_t bits = ((size_t)1 << (2 * l)) * bitCount;
        dataSize += (size_t)cnt * std::max<size_t>(bits >> 3, 1);
    }
    if (dataSize > 0xFFFFFFFFull) return 1;
    out.arrayData.assign(descCount ? dataSize : 0, 0);
    out.descs.resize(descCount);
    std::vector<uint32_t> descOffset(items.size(), 0xFFFFFFFFu);
    uint32_t off = 0, di = 0;
    if (descCount) for (const auto& kv : keys) {
        const HostItem& it = items[kv.second];
        if (it.special != 0) continue;
        if (off >= dataSize || di >= descCount) return 1;
        out.descs[di].offset = off; out.descs[di].subdivisionLevel = (uint16_t)it.level; out.descs[di].format = (uint16_t)it.format;
        descOffset[kv.second] = di++;
        const uint32_t nM = (uint32_t)num_micro(it), is2 = it.format == 1;
        uint8_t* dst = out.arrayData.data() + off;
        if (!is2 && it.uniform < 0) memcpy(dst, it.packed.data(), nM >= 4 ? nM / 4 : 1);   // (4-state: the packed states ARE the block, :1806-1816)
        else for (uint32_t u = 0; u < nM; ++u) {
            const uint32_t st = get_state(it, u);
            dst[u >> (2 + is2)] |= is2 ? (uint8_t)(st << (u & 7)) : (uint8_t)(st << ((u & 3) << 1));
        }
        off += std::max((nM * bitCount) >> 3u, 1u);
    }
    for (int f = 1; f <= 2; ++f) for (uint32_t l = 0; l < 13; ++l) {
        if (arrH[f][l]) out.arrayHist.push_back({ arrH[f][l], (uint16_t)l, (uint16_t)f });
        if (idxH[f][l]) out.indexHist.push_back({ idxH[f][l], (uint16_t)l, (uint16_t)f });
    }
    out.index.assign(d.numTris ? d.numTris : 1, d.unresolved);
    for (uint32_t i = 0; i < items.size(); ++i)
        for (uint32_t p : items[i].prims) out.index[p] = items[i].special != 0 ? items[i].special : (int32_t)descOffset[i];
    return 0;
}

} // namespace ommx
