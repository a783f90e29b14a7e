"""support/tests/test_omm_log.cpp restated: the LogTest fixture (:56-103) and its eight cases (:146-209).
std::uniform_real_distribution<float>(0,1) over std::mt19937(32) is restated from libstdc++'s generate_canonical<float, 24>:
one 32-bit draw / 2^32 in float, results >= 1 pulled to nextafter(1, 0)."""
import numpy as np
import ommtest as ot


def libstdcxx_uniform01(seed, n):
    raw = np.frombuffer(np.random.RandomState(seed).bytes(4 * n), dtype="<u4")   # RandomState(seed) == mt19937(seed), raw 32-bit outputs
    f = raw.astype(np.float32) / np.float32(4294967296.0)
    f[f >= 1.0] = np.nextafter(np.float32(1.0), np.float32(0.0))
    return f


def default_desc(lib, baker, triangle_count=256, alpha_cutoff=0.3, force_invalid=False):
    """CreateDefaultBakeInputDesc (:56-103): 1024^2 FP32 1-px checker with SAT, random triangles over [0,1)^2, Clamp/Nearest, level 4"""
    yy, xx = np.mgrid[0:1024, 0:1024]
    tex = np.where((xx % 2) != (yy % 2), 0.0, 1.0).astype(np.float32)
    t = lib.create_texture(baker, [tex], alpha_cutoff=alpha_cutoff, disable_zorder=True)
    uv = libstdcxx_uniform01(32, triangle_count * 6).reshape(-1, 2).astype(np.float32)
    if force_invalid:
        uv[:, 0] = np.inf
    ix = np.arange(triangle_count * 3, dtype=np.uint32)
    d = ot.make_desc(t, uv, ix, 4, alpha_cutoff=alpha_cutoff, addr=ot.CLAMP, filt=ot.NEAREST, dyn_scale=0.0,
                     flags=ot.FLAG_VALIDATION | ot.FLAG_NO_SPECIAL | ot.FLAG_NO_DEDUP)
    return d, t


PERF_WARNING = ("[Perf Warning] - The workload consists of 137972015 work items (number of texels to classify), which corresponds to roughly 131 1024x1024 textures."
                " This is unusually large and may result in long bake times.")
INVALID_TRIS = "[Info] - The workload consists of 256 unclassifiable triangles, these will be classified as unresolvedTriState = Fully Unknown Opaque."


def run_log_cases(lib):
    """every LogTest case: (expected messages, expected result) -- the callback must receive exactly these, in order"""
    def bake(mutator, expected_msgs, expected_result, with_callback=True, **kw):
        msgs = []
        b = lib.create_baker(callback=(lambda s, m, u: msgs.append(m.decode())) if with_callback else None)
        d, t = default_desc(lib, b, **kw)
        if mutator:
            mutator(d)
        r, out = lib.bake_raw(b, d)
        assert r == expected_result, (r, msgs)
        assert msgs == expected_msgs, msgs
        if r == ot.SUCCESS:
            assert out.value
            assert lib.fn("ommCpuDestroyBakeResult")(out) == ot.SUCCESS
        lib.destroy_texture(b, t)
        lib.destroy_baker(b)

    def no_texture(d): d.texture = None
    def bad_index(d): d.indexFormat = 3
    def level13(d): d.maxSubdivisionLevel = 13
    def cutoff04(d): d.alphaCutoff = 0.4
    def bad_states(d): d.alphaCutoffGreater, d.alphaCutoffLessEqual, d.format = ot.O, ot.UO, ot.FMT_2STATE
    bake(no_texture, ["[Invalid Argument] - ommCpuBakeInputDesc has no texture set"], ot.INVALID_ARGUMENT)                      # :146
    bake(bad_index, ["[Invalid Argument] - indexFormat is not set"], ot.INVALID_ARGUMENT)                                       # :154
    bake(level13, ["[Invalid Argument] - maxSubdivisionLevel (13) is greater than maximum supported (12)"], ot.INVALID_ARGUMENT)  # :162
    bake(cutoff04, ["[Invalid Argument] - Texture object alpha cutoff threshold (0.300000) is different from alpha cutoff threshold in bake input (0.400000)"],
         ot.INVALID_ARGUMENT)                                                                                                   # :170
    bake(bad_states, ["[Invalid Argument] - alphaCutoffLessEqual=UnknownOpaque is not compatible with OC1_2_State"], ot.INVALID_ARGUMENT)  # :178
    bake(None, [PERF_WARNING], ot.SUCCESS, triangle_count=511)                                                                  # :189
    bake(None, [INVALID_TRIS], ot.SUCCESS, triangle_count=256, alpha_cutoff=0.5, force_invalid=True)                            # :197
    bake(None, [], ot.INVALID_ARGUMENT, with_callback=False)                                                                    # :204  validation needs a log callback
