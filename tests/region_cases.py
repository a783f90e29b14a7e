"""Adversarial inputs for the curve-free-region test (omm_amd/csrc/region_curve.h), shared by the oracle audit (tests/test_region_curve_audit.py: the
header's verdicts against the reference states, on the CPU) and the device parity test (tests/test_gpu_parity.py: the kernels' results against the oracle):
textures that stress the error bounds (alpha hugging the cutoff, nearly flat patches with a twist around the 1e-6 branch threshold, 0 / 1 noise, FP32
values far from [0, 1], a non-power-of-two size), triangles with thin shapes (RcShape::fat == 0: the weaker edge-free verdict), nearly vertical / exactly
horizontal edges, sizes from far below to above a texel, UV offsets, every address mode, both formats, every promotion, with and without the summed-area table."""
import numpy as np
import ommtest as ot


def textures(seed=11):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    return [
        (0.5 + 1e-6 * (xx - 128) + 3e-7 * (yy - 128) + 1e-8 * (xx - 128) * (yy - 128)).astype(np.float32),
        (0.5 + 1e-3 * np.sin(xx * 0.7) * np.cos(yy * 0.9) + 2e-6 * rng.rand(256, 256)).astype(np.float32),
        (rng.rand(256, 256) > 0.5).astype(np.float32),
        (0.5 + 0.25 * np.sin(xx * 0.05) + 1e-5 * xx * yy / 256).astype(np.float32),
        (1000.0 * np.sin(xx * 0.11) * np.cos(yy * 0.07) + 0.5).astype(np.float32),                                  # FP32 alpha far outside [0, 1]
        (ot.value_noise(5 + (seed - 11), 300, 200, octaves=3, base_cell=16) * 255).astype(np.uint8),               # 200 x 300, not a power of two
        np.where(((xx.astype(np.int32) // 3 + yy.astype(np.int32) // 5) & 1) == 1, np.float32(0.5000001), np.float32(0.4999999)).astype(np.float32),  # steps of 2 ulp
    ]


SHAPES = ((0.02, 6, 30), (0.006, 5, 40), (0.05, 8, 6), (0.0015, 3, 60), (0.3, 7, 3), (0.004, 9, 2))   # (extent in UV, level, triangles)


def cases(seed=11, tri_seed_base=2000):
    """yields (texture index, texture, cutoff, make_desc keyword arguments without the texture handle) -- consecutive cases share texture and cutoff"""
    case = 0
    for ti, tx in enumerate(textures(seed)):
        for cutoff in (0.5, -1.0):
            for ext, level, n in SHAPES:
                case += 1
                uv, ix = ot.random_triangles(tri_seed_base + case, n, ext)
                tri = uv.reshape(-1, 3, 2)
                tri[::4, 1, 0] = tri[::4, 0, 0] + np.float32(1e-7)                       # nearly vertical edge
                tri[1::4, 2, 1] = tri[1::4, 0, 1]                                         # exactly horizontal edge
                tri[2::4, 2] = tri[2::4, 0] + (tri[2::4, 1] - tri[2::4, 0]) * np.float32(1.02) + np.float32(ext * 0.01)   # thin sliver
                off = (0.0, 3.0, -17.0, 900.0)[case % 4]
                addr = (ot.WRAP, ot.CLAMP, ot.MIRROR, ot.BORDER, ot.MIRROR_ONCE)[case % 5]
                promo = (ot.PROMO_NEAREST, ot.PROMO_FORCE_OPAQUE, ot.PROMO_FORCE_TRANSPARENT)[case % 3]
                fmt = ot.FMT_2STATE if case % 7 == 0 else ot.FMT_4STATE
                yield ti, tx, cutoff, dict(uv=(tri.reshape(-1, 2) + np.float32(off)).astype(np.float32), ix=ix, level=level, addr=addr, promo=promo, fmt=fmt,
                                           flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP)


def run(lib, seed=11, tri_seed_base=2000, each=None):
    """bakes every case with `lib`; each(case index, result) is called per bake"""
    b = lib.create_baker()
    cur, t, k = None, None, 0
    for ti, tx, cutoff, kw in cases(seed, tri_seed_base):
        if cur != (ti, cutoff):
            if t is not None:
                lib.destroy_texture(b, t)
            t = lib.create_texture(b, [tx], alpha_cutoff=cutoff); cur = (ti, cutoff)
        kw = dict(kw)
        d = ot.make_desc(t, kw.pop("uv"), kw.pop("ix"), kw.pop("level"), **kw)
        r = lib.bake(b, d, want_stats=False)
        if each is not None:
            each(k, r)
        k += 1
    if t is not None:
        lib.destroy_texture(b, t)
    lib.destroy_baker(b)
    return k
