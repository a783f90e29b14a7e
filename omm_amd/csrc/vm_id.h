// The work-item id of the reference's SetupWorkItems (libraries/omm-lib/src/bake_cpu_impl.cpp:626-631; the same chain as util/geometry.h:151-156):
//     std::size_t seed = 42; hash_combine(seed, p0); hash_combine(seed, p1); hash_combine(seed, p2); hash_combine(seed, subdivisionLevel); hash_combine(seed, ommFormat);
//     hash_combine(seed, v): seed ^= std::hash<T>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2)
// The reference keys its triangle -> work item map by this value ALONE (:633-649): two triangles with the same id are one work item even when their
// coordinates differ, and the dedup of the bake has to group triangles exactly like that to be a drop-in.  The std::hash specialisations are those of the
// reference's Linux build (neither dependency is vendored under /root/reference):
//   * std::hash<float>, libstdc++ functional_hash.h: 0 for +-0, else _Hash_bytes(&v, 4, 0xc70f6907) -- libsupc++ hash_bytes.cc, the 64-bit Murmur-style variant:
//     h = seed ^ (len * m); [tail of 4 bytes:] h ^= bytes; h *= m; h = shift_mix(h) * m; h = shift_mix(h)   with m = 0xc6a4a7935bd1e995, shift_mix(v) = v ^ (v >> 47)
//   * std::hash<glm::vec2>, glm/gtx/hash.inl: seed = 0; glm::detail::hash_combine(seed, hash(x)); hash_combine(seed, hash(y))   (the same combine step)
//   * std::hash<int32_t>, std::hash<ommFormat>: the value converted to size_t
// Checked against the container's libstdc++ (tests/native/std_hash_probe.cpp) and against the oracle's restatement on colliding inputs
// (tests/golden/vmid_collisions.json).  Plain C++: compiled for the host and for the device.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define OMMX_VMID_FN __host__ __device__ inline
#else
#define OMMX_VMID_FN inline
#endif

namespace ommx {

OMMX_VMID_FN uint64_t vmid_shift_mix(uint64_t v) { return v ^ (v >> 47); }
OMMX_VMID_FN uint64_t vmid_hash_float(float f)
{
    const uint64_t mul = (((uint64_t)0xc6a4a793u) << 32) + (uint64_t)0x5bd1e995u;
    if (!(f != 0.0f)) return 0;   // +0 and -0 (NaN hashes its bytes; such triangles are invalid and never get here)
    uint32_t bits;
#ifdef __HIP_DEVICE_COMPILE__
    bits = __float_as_uint(f);
#else
    memcpy(&bits, &f, 4);
#endif
    uint64_t h = (uint64_t)0xc70f6907u ^ (4ull * mul);
    h ^= (uint64_t)bits; h *= mul;
    h = vmid_shift_mix(h) * mul;
    return vmid_shift_mix(h);
}
OMMX_VMID_FN void vmid_combine(uint64_t& seed, uint64_t h) { seed ^= h + 0x9e3779b9u + (seed << 6) + (seed >> 2); }
OMMX_VMID_FN uint64_t vmid_hash_vec2(float x, float y) { uint64_t s = 0; vmid_combine(s, vmid_hash_float(x)); vmid_combine(s, vmid_hash_float(y)); return s; }
OMMX_VMID_FN uint64_t vm_id(const float uv[6], int32_t level, int32_t format)
{
    uint64_t seed = 42;
    vmid_combine(seed, vmid_hash_vec2(uv[0], uv[1]));
    vmid_combine(seed, vmid_hash_vec2(uv[2], uv[3]));
    vmid_combine(seed, vmid_hash_vec2(uv[4], uv[5]));
    vmid_combine(seed, (uint64_t)(int64_t)level);
    vmid_combine(seed, (uint64_t)(int64_t)format);
    return seed;
}

} // namespace ommx
