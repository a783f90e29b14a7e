"""Seeded synthetic workloads of the BASELINE.json configurations, shared by bench.py, the fixture generator
(tests/golden/make_oracle_fixtures.py) and the full-size GPU tests.  Everything is repository code: textures from tests/native/kat_textures.c,
triangles from the counter hash of tests/ommtest.py -- identical on the CPU (oracle) and the GPU box.

    c0     single quad, 256^2 checkerboard (32-texel squares, FP32), level 4                     -- BASELINE configs[0]
    c1     100 k random-UV triangles of ~10 texels, 2048^2 value noise (UNORM8), level 6           -- configs[1]
    c2     1 M random-UV triangles of ~8 texels, 4096^2 foliage-style alpha (UNORM8), level 8      -- configs[2] / [3], the metric configuration
    c4     4 M triangles of ~3 texels, per-triangle levels U{4..10} (75 %) / dynamic heuristic (25 %, scale 2, max 10), 8192^2 foliage  -- configs[4]
    cards  asset-shaped: axis-aligned quads ("foliage cards") that cover 256..1024 texels of the 4096^2 foliage texture each, levels 6..8, Clamp:
           a micro-triangle spans 1..16 texels, so the generic texel-loop path (conservative raster + level-line kernel per texel) carries the
           work -- the shape of the reference's own Leaflet KATs (support/tests/test_omm_bake_cpu.cpp:1733-2019), at production size
All: 4-state (c0: either), Linear filter, texture alphaCutoff 0.5 (summed-area table on), ForceOpaque promotion (the SDK default, omm.h:483)."""
import numpy as np
import ommtest as ot


def card_quads(seed, n, tex_size, lo_texels=256.0, hi_texels=1024.0):
    """n axis-aligned quads (2 triangles each, 4 unshared vertices per quad) inside [0,1]^2; edge length log-uniform in [lo, hi] texels,
    aspect ratio in [0.5, 2]; per-triangle level 6..8 (both triangles of a quad share it)"""
    u = [ot.uniform01(seed, n, k) for k in range(5)]
    edge = np.exp(np.log(lo_texels) + u[0] * (np.log(hi_texels) - np.log(lo_texels))).astype(np.float32) / np.float32(tex_size)
    aspect = np.exp((u[1] - np.float32(0.5)) * np.float32(2.0 * np.log(2.0))).astype(np.float32)
    w = np.minimum(edge * np.sqrt(aspect), np.float32(0.9)).astype(np.float32); h = np.minimum(edge / np.sqrt(aspect), np.float32(0.9)).astype(np.float32)
    x0 = (u[2] * (np.float32(1.0) - w)).astype(np.float32); y0 = (u[3] * (np.float32(1.0) - h)).astype(np.float32)
    uv = np.empty((n, 4, 2), np.float32)
    uv[:, 0] = np.stack([x0, y0], 1); uv[:, 1] = np.stack([x0, y0 + h], 1); uv[:, 2] = np.stack([x0 + w, y0], 1); uv[:, 3] = np.stack([x0 + w, y0 + h], 1)
    base = (np.arange(n, dtype=np.uint32) * 4)[:, None]
    ix = (base + np.array([0, 1, 2, 3, 1, 2], np.uint32)[None, :]).astype(np.uint32).reshape(-1)      # the SDK tests' quad split (test_omm_bake_cpu.cpp:594)
    lv = (6 + (ot.hash_u32(np.arange(n) + 77 * seed) % 3)).astype(np.uint8)
    return uv.reshape(-1, 2), ix, np.repeat(lv, 2)


def workload(kind, tris=None, fmt=ot.FMT_4STATE):
    """-> (texture, uv, indices, per-triangle levels or None, make_desc keywords incl. `level`)"""
    if kind == "c0":
        yy, xx = np.mgrid[0:256, 0:256]
        tex = (((xx // 32) + (yy // 32)) & 1).astype(np.float32)
        uv = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], np.float32)       # the SDK tests' quad (test_omm_bake_cpu.cpp:594-595)
        ix = np.array([0, 1, 2, 3, 1, 2], np.uint32)
        return tex, uv, ix, None, dict(level=4, fmt=fmt, addr=ot.CLAMP, promo=ot.PROMO_FORCE_OPAQUE)
    if kind == "c1":
        tex = (ot.value_noise(77, 2048, 2048, octaves=5, base_cell=128) * 255).astype(np.uint8)
        uv, ix = ot.random_triangles(78, tris or 100000, 10.0 / 2048)
        return tex, uv, ix, None, dict(level=6, fmt=ot.FMT_4STATE, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    if kind == "c2":
        tex = ot.foliage_texture(1234, 4096, 4096, feature=64)
        uv, ix = ot.random_triangles(1235, tris or 1000000, 8.0 / 4096)
        return tex, uv, ix, None, dict(level=8, fmt=ot.FMT_4STATE, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    if kind == "c4":
        n = tris or 4000000
        tex = ot.foliage_texture(4321, 8192, 8192, feature=96)
        uv, ix = ot.random_triangles(9, n, 3.0 / 8192)
        h = ot.hash_u32(np.arange(n) + 9000)
        lv = (4 + (h >> 8) % 7).astype(np.uint8); lv[(h & 3) == 0] = 0xF
        return tex, uv, ix, lv, dict(level=10, fmt=ot.FMT_4STATE, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, dyn_scale=2.0)
    if kind == "cards":
        n = (tris or 40000) // 2
        tex = ot.foliage_texture(1234, 4096, 4096, feature=64)
        uv, ix, lv = card_quads(55, n, 4096)
        return tex, uv, ix, lv, dict(level=8, fmt=ot.FMT_4STATE, addr=ot.CLAMP, promo=ot.PROMO_FORCE_OPAQUE)
    raise ValueError(kind)


def subset(uv, ix, lv, lo, hi):
    """triangles [lo, hi) of a workload as their own (uv, ix, lv): vertices re-indexed, so that the sample is a self-contained bake"""
    tri = ix[3 * lo:3 * hi].astype(np.int64)
    used, inv = np.unique(tri, return_inverse=True)
    return np.ascontiguousarray(uv[used]), inv.astype(np.uint32), (None if lv is None else np.ascontiguousarray(lv[lo:hi]))
