"""support/tests/test_basic.cpp restated: baker handles (:28-51) and texture creation limits (:217-279)."""
import ctypes as C
import numpy as np
import ommtest as ot


def run_basic_cases(lib):
    fn = lib.fn
    # Lib.VersionCheck (:19-26) is covered by tests/test_abi_exports.py
    assert fn("ommDestroyBaker")(None) == ot.INVALID_ARGUMENT                                   # Baker.DestroyNull
    b = lib.create_baker(); assert b.value; assert lib.destroy_baker(b) == ot.SUCCESS          # Baker.CreateDestroy
    d = ot.BakerCreationDesc(); d.type = 2                                                       # ommBakerType_MAX_NUM
    out = C.c_void_p()
    assert fn("ommCreateBaker")(C.byref(d), C.byref(out)) == ot.INVALID_ARGUMENT                # Baker.CreateInvalid
    g = lib.create_baker(baker_type=0); assert g.value; assert lib.destroy_baker(g) == ot.SUCCESS   # Baker.CreateDestroyGPU
    b = lib.create_baker()
    assert fn("ommCpuDestroyTexture")(b, None) == ot.INVALID_ARGUMENT                           # TextureTest.DestroyNull
    for (w, h, zorder, expect) in [(64, 100, True, ot.SUCCESS), (100, 100, True, ot.SUCCESS), (100, 64, True, ot.SUCCESS),   # :222-249
                                   (0, 64, True, ot.INVALID_ARGUMENT), (0, 0, True, ot.INVALID_ARGUMENT),                     # :251-263
                                   (65536, 1, False, ot.SUCCESS), (65537, 1, False, ot.INVALID_ARGUMENT)]:                    # :265-277
        t = lib.create_texture(b, [np.zeros((h, w), np.float32)], alpha_cutoff=-1.0, disable_zorder=not zorder, expect=expect)
        if expect == ot.SUCCESS:
            assert t.value and lib.destroy_texture(b, t) == ot.SUCCESS
    assert lib.destroy_baker(b) == ot.SUCCESS
