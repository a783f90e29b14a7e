// Exhaustive host-side check of the split bird-curve decode of omm_amd/csrc/classify_device.h (bird_group + bird_table_entry, what
// micro_triangle_grouped() builds its integer barycentrics from) against the direct decode of util/bird.h:73-118, for EVERY micro-triangle
// index of every level 3..12 (22 M cases).  Pure integer code, runs without a GPU:
//   hipcc -O2 -I omm_amd/csrc tests/native/bird_check.hip -o /tmp/bird_check && /tmp/bird_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "classify_device.h"
using namespace ommx;

int main()
{
    uint8_t table[256];
    for (uint32_t c = 0; c < 4; ++c) for (uint32_t l = 0; l < 64; ++l) table[c * 64 + l] = (uint8_t)bird_table_entry(c, l);
    unsigned long long checked = 0, bad = 0;
    for (uint32_t level = 3; level <= 12; ++level) {
        const uint32_t mask = (1u << level) - 1u;
        for (uint32_t index = 0; index < (1u << (2 * level)); ++index) {
            // direct decode (util/bird.h:73-118 as restated in micro_triangle())
            const uint32_t b0 = even_bits(index), b1 = even_bits(index >> 1);
            const uint32_t fx = prefix_xor(b0), fy = prefix_xor(b0 & ~b1);
            const uint32_t tt = fy ^ b1;
            uint32_t iu = ((fx & ~tt) | (b0 & ~tt) | (~b0 & ~fx & tt)) & mask, iv = (fy ^ b0) & mask, iw = ((~fx & ~tt) | (b0 & ~tt) | (~b0 & fx & tt)) & mask;
            const bool upright = ((iu ^ iv ^ iw) & 1u) != 0;
            if (!upright) { iu += 1; iv += 1; }
            // split decode
            const BirdGroup g = bird_group(index >> 6, level - 3);
            const uint32_t e = table[((g.word >> 24) & 3u) * 64 + (index & 63u)];
            const bool up2 = (e >> 6) != 0;
            const uint32_t adj = up2 ? 0u : 1u;
            const uint32_t iu2 = (((g.word & 0xFFFu) << 3) | (e & 7u)) + adj, iv2 = ((((g.word >> 12) & 0xFFFu) << 3) | ((e >> 3) & 7u)) + adj;
            checked++;
            if (iu2 != iu || iv2 != iv || up2 != upright) { if (bad++ < 5) printf("MISMATCH level %u index %u: (%u,%u,%d) vs (%u,%u,%d)\n", level, index, iu, iv, (int)upright, iu2, iv2, (int)up2); }
        }
    }
    printf("bird_check: %llu indices, %llu mismatches\n", checked, bad);
    return bad ? 1 : 0;
}
