/* Finds pairs of DIFFERENT UV points with the same std::hash<glm::vec2> (libstdc++ _Hash_bytes + glm/gtx/hash.inl's hash_combine), hence pairs of different
 * triangles with the same work-item id in the reference's SetupWorkItems (libraries/omm-lib/src/bake_cpu_impl.cpp:626-649: the map is keyed by the 64-bit
 * hash chain and trusts it).  For a base point (x, y) every float x' is tried: the y' that would complete the collision follows by inverting the hash
 * (every step is a bijection on 64 bits; a solution exists when the pre-image fits in 32 bits: about one x' in 2^32).
 *   gcc -O2 -fopenmp -o vmid_collision_search vmid_collision_search.c && ./vmid_collision_search 0.3 0.7
 * Test tool (generator of tests/golden/vmid_collisions.json); not part of the product. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
static const uint64_t MUL = (((uint64_t)0xc6a4a793UL) << 32) + (uint64_t)0x5bd1e995UL, C = 0x9e3779b9u;
static uint64_t sm(uint64_t v) { return v ^ (v >> 47); }
static uint64_t inv_mul(void) { uint64_t x = MUL; for (int i = 0; i < 6; ++i) x *= 2 - MUL * x; return x; }   /* Newton: MUL is odd */
static uint64_t hf_bits(uint32_t b) { uint64_t h = (uint64_t)0xc70f6907UL ^ (4 * MUL); h ^= b; h *= MUL; h = sm(h) * MUL; return sm(h); }
static uint64_t hf(float f) { uint32_t b; memcpy(&b, &f, 4); return f != 0.0f ? hf_bits(b) : 0; }
static uint64_t hv2(float x, float y) { uint64_t s = 0; s ^= hf(x) + C + (s << 6) + (s >> 2); s ^= hf(y) + C + (s << 6) + (s >> 2); return s; }
int main(int argc, char** argv)
{
    const float x = argc > 1 ? (float)atof(argv[1]) : 0.3f, y = argc > 2 ? (float)atof(argv[2]) : 0.7f, lim = argc > 3 ? (float)atof(argv[3]) : 4.f;
    const uint64_t T = hv2(x, y), IM = inv_mul(), K = (uint64_t)0xc70f6907UL ^ (4 * MUL);
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (1ll << 32); ++i) {
        const uint32_t xb = (uint32_t)i; float xc; memcpy(&xc, &xb, 4);
        if (!(fabsf(xc) < lim) || xc == 0.0f || xc == x) continue;
        const uint64_t s1 = hf_bits(xb) + C;
        const uint64_t W = (T ^ s1) - C - (s1 << 6) - (s1 >> 2);
        const uint64_t data = (sm(sm(W) * IM) * IM) ^ K;
        if (data >> 32) continue;
        const uint32_t yb = (uint32_t)data; float yc; memcpy(&yc, &yb, 4);
        if (!(fabsf(yc) < lim) || yc == 0.0f) continue;
        if (hv2(xc, yc) != T) continue;
        #pragma omp critical
        printf("{\"a\": [\"%08x\", \"%08x\"], \"b\": [\"%08x\", \"%08x\"], \"a_f\": [%.9g, %.9g], \"b_f\": [%.9g, %.9g], \"hash\": \"%016llx\"}\n",
               *(uint32_t*)&x, *(uint32_t*)&y, xb, yb, x, y, xc, yc, (unsigned long long)T);
    }
    return 0;
}
