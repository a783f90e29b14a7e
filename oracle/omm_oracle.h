/*
 * omm_oracle.h -- entry points of the CPU restatement of the reference CPU baker.
 * TEST INFRASTRUCTURE ONLY (see omm_oracle.c header).  The oracle mirrors the C ABI of
 * include/omm_mi355x.h one-to-one with an `oracle_` prefix so the same test driver can run
 * a bake through either library.
 */
#ifndef OMM_ORACLE_H
#define OMM_ORACLE_H
#include "../include/omm_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif
ommLibraryDesc oracle_ommGetLibraryDesc(void);
ommResult oracle_ommCreateBaker(const ommBakerCreationDesc* desc, ommBaker* outBaker);
ommResult oracle_ommDestroyBaker(ommBaker baker);
ommResult oracle_ommCpuCreateTexture(ommBaker baker, const ommCpuTextureDesc* desc, ommCpuTexture* outTexture);
ommResult oracle_ommCpuDestroyTexture(ommBaker baker, ommCpuTexture texture);
ommResult oracle_ommCpuBake(ommBaker baker, const ommCpuBakeInputDesc* desc, ommCpuBakeResult* out);
ommResult oracle_ommCpuDestroyBakeResult(ommCpuBakeResult r);
ommResult oracle_ommCpuGetBakeResultDesc(ommCpuBakeResult r, const ommCpuBakeResultDesc** desc);
ommResult oracle_ommDebugGetStats(ommBaker baker, const ommCpuBakeResultDesc* res, ommDebugStats* out);
ommResult oracle_ommDebugGetStats2(ommBaker baker, ommCpuBakeResult res, ommDebugStats* out);
const float* oracle_bake_result_areas(ommCpuBakeResult r);
/* bench support: phase wall times of the last bake (see omm_oracle.c) and the OpenMP thread count of the following ones */
void oracle_ommxGetLastBakeTimings(double out[8]);
void oracle_ommxSetThreads(int n);

/* unit-level probes used by tests */
void     orc_index2bary(uint32_t index, uint32_t level, float uv[6]);
void     orc_micro_triangle(const float tri[6], uint32_t index, uint32_t level, float out[6]);
void     orc_get_tex_coord(int mode, int pow2, int x, int y, int w, int h, int out[2]);
uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed);
uint64_t orc_sort_key(const float uv[6], uint32_t level);
uint64_t oracle_std_hash_float(float f);                                     /* libstdc++ std::hash<float> */
uint64_t oracle_vm_id(const float uv[6], int32_t level, int32_t format);     /* the work-item id of SetupWorkItems (bake_cpu_impl.cpp:626-631) */
#ifdef __cplusplus
}
#endif
#endif
