"""Audit of the curve-free-region test (omm_amd/csrc/region_curve.h): the HIP kernels settle a whole bird-curve sub-triangle -- a work item, a tile of 4096
micro-triangles, a group of 64 -- without classifying its micro-triangles when the level curve alpha == cutoff provably cannot reach it.  The audit build
of the oracle includes that header as it is, evaluates the test for EVERY sub-triangle of EVERY level of every work item it bakes (the hierarchy levels the
kernels use, and all the others) and compares a verdict "all descendants have the pure state s" with the states the reference algorithm produced for them.
counters (oracle/omm_oracle.c: audit_region_item): [0] sub-triangles tested, [1] settled, [2] micro-triangles under settled sub-triangles, [3] micro-triangles
whose reference state differs (must be 0), [4] / [5] work items whose shape admits the test / all, [6] / [7] = [1] / [2] for sub-triangles whose cells hold
texels on both sides of the cutoff (what the summed-area table cannot settle); work items without the corner bound (thin ones, or micro-triangles smaller
than the rounding of their vertices: RcShape::fat == 0) get the weaker verdict "pure state unless PointInTriangle puts a wrong-side cell corner inside":
[8] sub-triangles with that verdict, [9] descendants it covers, [10] descendants left to the full pass (a wrong-side corner inside), [11] covered ones whose
reference state differs (must be 0)."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
import ommtest as ot
import workloads as wl

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
    lib.dll.orc_audit_region_counter.restype = C.c_longlong
    lib.dll.orc_audit_region_counter.argtypes = [C.c_int]
    return lib


def counters(lib):
    return [lib.dll.orc_audit_region_counter(i) for i in range(12)]


def test_region_verdicts_on_the_bench_workloads(audit):
    """the first triangles of the BASELINE configurations' own streams, with and without the summed-area table"""
    audit.dll.orc_audit_region_reset()
    for kind, n in (("c2", 120), ("c1", 1200), ("c4", 700), ("cards", 40)):
        tex, uv, ix, lv, kw = wl.workload(kind, n)
        k = dict(kw); level = k.pop("level")
        b = audit.create_baker()
        for cutoff in ((0.5, -1.0) if kind == "c2" else (0.5,)):
            t = audit.create_texture(b, [tex], alpha_cutoff=cutoff)
            audit.bake(b, ot.make_desc(t, uv, ix, level, levels=lv, filt=ot.LINEAR, flags=ot.FLAG_THREADS, **k), want_stats=False)
            audit.destroy_texture(b, t)
        audit.destroy_baker(b)
    c = counters(audit)
    assert c[0] > 10_000_000 and c[4] > 1000, c
    assert c[6] > 100_000 and c[7] > 1_000_000, c      # settled where the texels around the sub-triangle are NOT all on one side
    assert c[8] > 1_000_000 and c[9] > 10_000_000, c   # the weaker verdict (no corner bound): configs[4]'s level-9 / 10 items are all of that kind
    assert c[3] == 0 and c[11] == 0, c


def test_region_verdicts_on_adversarial_inputs(audit):
    """textures that stress the error bounds (alpha hugging the cutoff, nearly flat patches with a twist around the 1e-6 branch threshold, 0 / 1 noise, FP32
    values far from [0, 1], a non-power-of-two size), triangles with thin shapes, nearly vertical / horizontal edges, sizes from far below to above a texel,
    UV offsets, every address mode, 2-state and every promotion"""
    audit.dll.orc_audit_region_reset()
    rng = np.random.RandomState(11)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    texs = [
        (0.5 + 1e-6 * (xx - 128) + 3e-7 * (yy - 128) + 1e-8 * (xx - 128) * (yy - 128)).astype(np.float32),
        (0.5 + 1e-3 * np.sin(xx * 0.7) * np.cos(yy * 0.9) + 2e-6 * rng.rand(256, 256)).astype(np.float32),
        (rng.rand(256, 256) > 0.5).astype(np.float32),
        (0.5 + 0.25 * np.sin(xx * 0.05) + 1e-5 * xx * yy / 256).astype(np.float32),
        (1000.0 * np.sin(xx * 0.11) * np.cos(yy * 0.07) + 0.5).astype(np.float32),                                  # FP32 alpha far outside [0, 1]
        (ot.value_noise(5, 300, 200, octaves=3, base_cell=16) * 255).astype(np.uint8),                             # 200 x 300, not a power of two
        np.where(((xx.astype(np.int32) // 3 + yy.astype(np.int32) // 5) & 1) == 1, np.float32(0.5000001), np.float32(0.4999999)).astype(np.float32),  # steps of 2 ulp
    ]
    b = audit.create_baker()
    case = 0
    for ti, tx in enumerate(texs):
        for cutoff in (0.5, -1.0):
            t = audit.create_texture(b, [tx], alpha_cutoff=cutoff)
            for ext, level, n in ((0.02, 6, 30), (0.006, 5, 40), (0.05, 8, 6), (0.0015, 3, 60), (0.3, 7, 3), (0.004, 9, 2)):
                case += 1
                uv, ix = ot.random_triangles(2000 + case, n, ext)
                tri = uv.reshape(-1, 3, 2)
                tri[::4, 1, 0] = tri[::4, 0, 0] + np.float32(1e-7)                       # nearly vertical edge
                tri[1::4, 2, 1] = tri[1::4, 0, 1]                                         # exactly horizontal edge
                tri[2::4, 2] = tri[2::4, 0] + (tri[2::4, 1] - tri[2::4, 0]) * np.float32(1.02) + np.float32(ext * 0.01)   # thin sliver
                off = (0.0, 3.0, -17.0, 900.0)[case % 4]
                addr = (ot.WRAP, ot.CLAMP, ot.MIRROR, ot.BORDER, ot.MIRROR_ONCE)[case % 5]
                promo = (ot.PROMO_NEAREST, ot.PROMO_FORCE_OPAQUE, ot.PROMO_FORCE_TRANSPARENT)[case % 3]
                fmt = ot.FMT_2STATE if case % 7 == 0 else ot.FMT_4STATE
                d = ot.make_desc(t, (tri.reshape(-1, 2) + np.float32(off)).astype(np.float32), ix, level, addr=addr, promo=promo, fmt=fmt,
                                 flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP)
                audit.bake(b, d, want_stats=False)
            audit.destroy_texture(b, t)
    audit.destroy_baker(b)
    c = counters(audit)
    assert c[0] > 500_000 and c[1] > 400_000 and c[6] > 2_000, tuple(c)
    assert c[3] == 0 and c[11] == 0, c
