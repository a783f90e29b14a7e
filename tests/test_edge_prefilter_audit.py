"""Audit of the three places where the HIP kernel does NOT evaluate the reference's expression for every input: point_on_edge() in
omm_amd/csrc/classify_device.h discards points that a sqrt-free bound proves to be off the segment, root_rejected()
skips the division of a level-curve root that provably cannot be accepted (section 5.3b), and curve_excluded() skips all three edge tests of a
micro-triangle whose fattened bounding box the level curve provably stays away from (section 5.3c).
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
    lib.dll.orc_audit_curve_counter.restype = C.c_longlong
    lib.dll.orc_audit_curve_counter.argtypes = [C.c_int]
    return lib


def test_bound_never_discards_a_point_the_reference_accepts(audit):
    audit.dll.orc_audit_reset(); audit.dll.orc_audit_curve_reset()
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
    # curve exclusion (classify_device.h curve_excluded): never true when one of the three edge tests succeeds
    calls3, excluded3, bad3, crossings3 = (audit.dll.orc_audit_curve_counter(i) for i in range(4))
    assert calls3 > 1000000 and excluded3 > 100000 and crossings3 > 10000
    assert bad3 == 0


def test_curve_exclusion_on_adversarial_patches(audit):
    """curve_excluded() against textures that stress its error bounds: tiny bilinear twist (|hd| around the 1e-6 branch threshold), alpha values
    within a few ulp of the cutoff, steep two-level ramps, and micro-triangles with nearly horizontal / vertical edges, at sub-texel sizes"""
    audit.dll.orc_audit_curve_reset()
    rng = np.random.RandomState(7)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    texs = [
        (0.5 + 1e-6 * (xx - 128) + 3e-7 * (yy - 128) + 1e-8 * (xx - 128) * (yy - 128)).astype(np.float32),        # nearly flat, twist ~1e-8
        (0.5 + 1e-3 * np.sin(xx * 0.7) * np.cos(yy * 0.9) + 2e-6 * rng.rand(256, 256)).astype(np.float32),         # alpha hugging the cutoff
        (rng.rand(256, 256) > 0.5).astype(np.float32),                                                                # 0 / 1 noise: steep patches, |hd| up to 2
        (0.5 + 0.25 * np.sin(xx * 0.05) + 1e-5 * xx * yy / 256).astype(np.float32),
    ]
    b = audit.create_baker()
    for tx in texs:
        t = audit.create_texture(b, [tx], alpha_cutoff=-1.0)
        for seed, (ext, level, n) in enumerate([(0.02, 6, 40), (0.006, 5, 60), (0.05, 8, 12)]):
            uv, ix = ot.random_triangles(900 + seed, n, ext)
            # a third of the triangles get an axis-aligned edge (nearly vertical / horizontal carrier lines at every micro-triangle)
            tri = uv.reshape(-1, 3, 2); tri[::3, 1, 0] = tri[::3, 0, 0] + np.float32(1e-7); tri[1::3, 2, 1] = tri[1::3, 0, 1]
            d = ot.make_desc(t, tri.reshape(-1, 2), ix, level, addr=ot.WRAP, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP)
            audit.bake(b, d, want_stats=False)
        audit.destroy_texture(b, t)
    audit.destroy_baker(b)
    calls3, excluded3, bad3, crossings3 = (audit.dll.orc_audit_curve_counter(i) for i in range(4))
    assert calls3 > 200000 and excluded3 > 1000 and crossings3 > 1000, (calls3, excluded3, crossings3)
    assert bad3 == 0


def _cell_counts(audit):
    audit.dll.orc_audit_cell_counter.restype = C.c_longlong
    audit.dll.orc_audit_cell_counter.argtypes = [C.c_int]
    return [audit.dll.orc_audit_cell_counter(i) for i in range(8)]


def test_cell_exclusion_on_asset_sized_triangles(audit):
    """cell_excluded() (classify_device.h): micro-triangles LARGER than a texel -- the generic texel-loop path of asset-sized
    triangles.  Every visit of the level-line kernel evaluates the predicate next to the three edge tests: it must never hold where one of them finds
    the curve.  Axis-aligned quads (vertical and exactly horizontal edges), random triangles, smooth / noisy / adversarial alpha, UNORM8 and FP32,
    micro-triangles of 1 .. 60 texels.  The kernel ships the bounds with 8x the rounding unit they are derived for; the derivation itself (1x) and a probe
    below it (0.25x) are evaluated alongside."""
    import workloads as wl
    audit.dll.orc_audit_cell_reset()
    rng = np.random.RandomState(11)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    texs = [
        ot.foliage_texture(77, 512, 512, feature=48),
        (ot.value_noise(5, 300, 200, octaves=5, base_cell=24)).astype(np.float32),
        (0.5 + 1e-6 * (xx - 128) + 3e-7 * (yy - 128) + 1e-8 * (xx - 128) * (yy - 128)).astype(np.float32),        # nearly flat, twist ~1e-8
        (0.5 + 1e-3 * np.sin(xx * 0.7) * np.cos(yy * 0.9) + 2e-6 * rng.rand(256, 256)).astype(np.float32),         # alpha hugging the cutoff
        (rng.rand(256, 256) > 0.5).astype(np.float32),                                                                # 0 / 1 noise: steep patches, |hd| up to 2
        (0.5 + 0.25 * np.sin(xx * 0.05) + 1e-5 * xx * yy / 256).astype(np.float32),
    ]
    b = audit.create_baker()
    for ti, tx in enumerate(texs):
        t = audit.create_texture(b, [tx], alpha_cutoff=-1.0)
        size = tx.shape[0]
        # quads of 20 .. 400 texels at levels 3 .. 6 (micro-triangles of 1 .. 50 texels), and random (not axis-aligned) triangles
        uv, ix, lv = wl.card_quads(30 + ti, 12, size, lo_texels=20.0, hi_texels=min(400.0, size * 0.8))
        for level in (3, 5):
            audit.bake(b, ot.make_desc(t, uv, ix, level, addr=ot.CLAMP, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP), want_stats=False)
        uv2, ix2 = ot.random_triangles(400 + ti, 40, 0.15)
        for level, addr in ((3, ot.WRAP), (5, ot.MIRROR)):
            audit.bake(b, ot.make_desc(t, uv2, ix2, level, addr=addr, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP), want_stats=False)
        # nearly vertical / nearly horizontal edges: the quads sheared by a few 1e-7 .. 1e-3
        uvs = uv.copy(); uvs[:, 0] += (uvs[:, 1] * np.float32(1e-4)).astype(np.float32)
        audit.bake(b, ot.make_desc(t, uvs, ix, 4, addr=ot.CLAMP, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP), want_stats=False)
        audit.destroy_texture(b, t)
    audit.destroy_baker(b)
    calls, excluded, bad, crossings, derived, derived_bad, probe, probe_bad = _cell_counts(audit)
    assert calls > 1000000 and excluded > calls // 10 and crossings > 10000, (calls, excluded, crossings)
    assert bad == 0, (calls, excluded, bad)                 # the shipped predicate (8x)
    assert derived_bad == 0, (derived, derived_bad)         # the bounds as derived (1x)
    assert probe >= derived >= excluded                     # (0.25x: below the derivation; it may -- and does -- exclude next to a crossing: probe_bad is not asserted)


def _corner_counts(audit):
    audit.dll.orc_audit_corner_counter.restype = C.c_longlong
    audit.dll.orc_audit_corner_counter.argtypes = [C.c_int]
    return [audit.dll.orc_audit_corner_counter(i) for i in range(3)]


def test_corner_votes_skipped_only_where_no_corner_can_be_inside(audit):
    """rc_corners_far() (omm_amd/csrc/region_curve.h, round 6): fine_single_texel() skips the four PointInTriangle tests of a cell visit when the work item has the
    corner bound and every corner of the cell lies outside the micro-triangle's box fattened by rho.  The audit build evaluates the predicate on EVERY cell
    visit of the level-line kernel, with the box formed exactly as the kernel forms it, next to the four tests of the reference: a visit the predicate calls
    "far" must never have a corner inside.  Sweep: the BASELINE workloads' first triangles, micro-triangles from 1e-5 to 150 texels, UV offsets of +-1000 and
    -70000 (coarse fp32 vertices), slivers and axis-aligned edges, non-power-of-two sizes, every address mode, FP32 and UNORM8 texels, alpha hugging the cutoff."""
    import workloads as wl
    audit.dll.orc_audit_corner_reset()
    rng = np.random.RandomState(3)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    b = audit.create_baker()
    # (1) the configurations of BASELINE.json: the first triangles of each stream, their own textures and levels
    for cfg, n in (("c1", 300), ("c2", 100), ("c4", 100), ("cards", 16)):
        tex, uv, ix, lv, kw = wl.workload(cfg, n)
        kw = dict(kw); level = kw.pop("level")
        t = audit.create_texture(b, [tex], alpha_cutoff=-1.0)
        audit.bake(b, ot.make_desc(t, uv, ix, level, levels=lv, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP, **kw), want_stats=False)
        audit.destroy_texture(b, t)
    # (2) sizes, offsets, address modes
    tex8 = (ot.value_noise(5, 512, 512, octaves=5, base_cell=32) * 255).astype(np.uint8)
    texf = ot.value_noise(6, 300, 200, octaves=3, base_cell=16).astype(np.float32)
    hug = (0.5 + 1e-3 * np.sin(xx * 0.7) * np.cos(yy * 0.9) + 2e-6 * rng.rand(256, 256)).astype(np.float32)
    for tx in (tex8, texf, hug):
        t = audit.create_texture(b, [tx], alpha_cutoff=-1.0)
        for seed, (ext, level, n) in enumerate([(0.3, 0, 20), (0.1, 3, 24), (0.05, 5, 20), (0.01, 7, 8), (0.004, 8, 4), (0.004, 6, 24), (1e-5, 6, 16), (3.0, 2, 6)]):
            for off in (0.0, 1000.0, -70000.0):
                uv, ix = ot.random_triangles(700 + seed, n, ext)
                tri = uv.reshape(-1, 3, 2)
                tri[::4, 1, 0] = tri[::4, 0, 0] + np.float32(1e-7); tri[1::4, 2, 1] = tri[1::4, 0, 1]                 # nearly vertical / exactly horizontal edges
                tri[2::4, 2] = tri[2::4, 0] + (tri[2::4, 1] - tri[2::4, 0]) * np.float32(0.5) + np.float32(ext * 1e-3)    # slivers (thin: no corner bound)
                uvo = (tri.reshape(-1, 2) + np.float32(off)).astype(np.float32)
                for addr in ((ot.WRAP, ot.MIRROR, ot.CLAMP, ot.BORDER, ot.MIRROR_ONCE) if off == 0.0 else (ot.WRAP, ot.CLAMP)):
                    audit.bake(b, ot.make_desc(t, uvo, ix, level, addr=addr, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP), want_stats=False)
        audit.destroy_texture(b, t)
    audit.destroy_baker(b)
    visits, far, bad = _corner_counts(audit)
    # (the long form of this sweep -- ~4x the triangles, every address mode at every offset, five minutes -- ran when the predicate went in: 0 disagreements)
    assert visits > 1000000 and far > visits // 4, (visits, far)
    assert bad == 0, (visits, far, bad)
