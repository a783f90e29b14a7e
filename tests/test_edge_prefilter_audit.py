"""Audit of the two places where the HIP kernel does NOT evaluate the reference's expression for every input: point_on_edge() in
omm_amd/csrc/classify_device.h discards points that a sqrt-free bound proves to be off the segment, and root_rejected()
skips the division of a level-curve root that provably cannot be accepted (section 5.3b).
The audit build of the oracle evaluates the reference expression AND the bound for every call and counts disagreements."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
import ommtest as ot

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def audit():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libomm_oracle_audit.so"])
    saved = ot.oracle_path
    ot.oracle_path = lambda: os.path.join(ROOT, "oracle", "libomm_oracle_audit.so")
    try:
        lib = ot.Lib("oracle")
    finally:
        ot.oracle_path = saved
    lib.dll.orc_audit_counter.restype = C.c_longlong
    lib.dll.orc_audit_counter.argtypes = [C.c_int]
    lib.dll.orc_audit_min_discarded.restype = C.c_float
    return lib


def test_bound_never_discards_a_point_the_reference_accepts(audit):
    audit.dll.orc_audit_reset()
    tex8 = (ot.value_noise(5, 512, 512, octaves=5, base_cell=32) * 255).astype(np.uint8)
    texf = ot.value_noise(6, 300, 200, octaves=3, base_cell=16).astype(np.float32)
    b = audit.create_baker()
    for tx in (tex8, texf):
        t = audit.create_texture(b, [tx], alpha_cutoff=0.5)
        # triangle extent (UV), level, count: from 150-texel micro-triangles down to 1e-5 texels, incl. far-away UV tiles
        for seed, (ext, level, n) in enumerate([(0.3, 0, 60), (0.2, 1, 60), (0.1, 3, 80), (0.05, 5, 80), (0.01, 7, 60), (0.002, 8, 30), (1e-5, 6, 60), (3.0, 2, 10)]):
            for off in (0.0, 1000.0, -70000.0):
                uv, ix = ot.random_triangles(100 + seed, n, ext)
                uv = (uv + np.float32(off)).astype(np.float32)
                for addr in (ot.WRAP, ot.MIRROR, ot.CLAMP):
                    d = ot.make_desc(t, uv, ix, level, addr=addr, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP)
                    audit.bake(b, d, want_stats=False)
        audit.destroy_texture(b, t)
    audit.destroy_baker(b)
    calls, discarded, bad = (audit.dll.orc_audit_counter(i) for i in range(3))
    assert calls > 1000000 and discarded > calls // 2
    assert bad == 0
    assert audit.dll.orc_audit_min_discarded() > 1e-4        # the reference's threshold is 1e-5: >= 10x margin observed
    # the root filter (classify_device.h root_rejected): decided from the operands of the division, before the division
    roots, rejected, bad_roots = (audit.dll.orc_audit_counter(i) for i in (4, 5, 6))
    assert roots > 1000000 and rejected > 1000000   # (large micro-triangles, M > 2, are not filtered: most of this sweep)
    assert bad_roots == 0
