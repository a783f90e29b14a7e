// host_tail.h -- serial tail for the opt-in near-duplicate merge / array-size budget (see host_tail.cpp)
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/omm_mi355x.h"

namespace ommx {

struct HostItem {
    uint32_t level = 0; int format = 2; float uv[6];
    std::vector<uint32_t> prims;   // referencing triangles, ascending
    int32_t special = 0;           // 0 = none
    int uniform = -1;              // >= 0: every micro-triangle has this state and `packed` is empty
    std::vector<uint8_t> packed;   // else 2 bits per micro-triangle, LSB first (max(1, 4^level / 4) bytes), whatever the output format
};
struct HostTailDesc {
    int format; bool disableSpecial, disableDedup, nearDup, nearDupBrute;
    float rejectionThreshold, nearDupFactor; uint32_t maxArrayDataSize; uint32_t numTris; int32_t unresolved;
};
struct HostTailResult {
    std::vector<uint8_t> arrayData; std::vector<ommCpuOpacityMicromapDesc> descs;
    std::vector<ommCpuOpacityMicromapUsageCount> arrayHist, indexHist; std::vector<int32_t> index;
};
// returns 0 on success, 1 for ommResult_FAILURE (array data > 4 GiB or an undersized budget walk)
int run_host_tail(const HostTailDesc& d, std::vector<HostItem>& items, HostTailResult& out);

} // namespace ommx
