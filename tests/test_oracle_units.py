"""Unit-level pins of the oracle's building blocks against the reference's own unit tests / published algorithms."""
import ctypes as C
import json
import os
import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = dict(Wrap=0, Mirror=1, Clamp=2, Border=3, MirrorOnce=4)
B = 0x7FFFFFFE  # kTexCoordBorder, util/texture.h:23


def _texcoord(oracle, mode, x, y, w, h):
    out = (C.c_int * 2)()
    pow2 = int((w & (w - 1)) == 0 and (h & (h - 1)) == 0)
    oracle.dll.orc_get_tex_coord(MODES[mode], pow2, x, y, w, h, out)
    return out[0], out[1]


def test_get_tex_coord_reference_tables(oracle):
    """support/tests/test_texture.cpp:40-266"""
    kats = json.load(open(os.path.join(GOLDEN, "texcoord_kat.json")))
    assert len(kats) == 181
    for mode, x, y, w, h, ex, ey in kats:
        assert _texcoord(oracle, mode, x, y, w, h) == (ex, ey), (mode, x, y, w, h)
    border = [((512, 512), (B, 512)), ((-1, -1), (B, B)), ((0, -1), (0, B)), ((-1024, -1), (B, B)),
              ((-2048, -1), (B, B)), ((1024, 1024), (B, B)), ((2048, 1024), (B, B))]
    for (x, y), exp in border:
        assert _texcoord(oracle, "Border", x, y, 512, 1024) == exp


def test_xxh64_matches_published_implementation(oracle):
    xxhash = pytest.importorskip("xxhash")
    f = oracle.dll.orc_xxh64
    f.restype = C.c_uint64
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    rng = np.random.default_rng(7)
    for n in [0, 1, 3, 4, 7, 8, 15, 16, 31, 32, 33, 63, 64, 100, 256, 1024, 4096, 65536]:
        d = rng.integers(0, 4, n, dtype=np.uint8).tobytes()
        for seed in (0, 42):
            assert f(d, n, seed) == xxhash.xxh64(d, seed=seed).intdigest(), (n, seed)


def test_bird_curve_tiles_the_triangle(oracle):
    """util/bird.h: every level-N micro-triangle is distinct, has area 4^-N and lies inside the unit triangle."""
    uv = (C.c_float * 6)()
    for level in range(0, 6):
        n = 4 ** level
        seen = set()
        for i in range(n):
            oracle.dll.orc_index2bary(i, level, uv)
            pts = tuple(round(float(v) * (1 << level)) for v in uv)
            seen.add(pts)
            a = abs((uv[2] - uv[0]) * (uv[5] - uv[1]) - (uv[4] - uv[0]) * (uv[3] - uv[1])) / 2
            assert abs(a - 0.5 / n) < 1e-9
            for k in range(3):
                u, v = uv[2 * k], uv[2 * k + 1]
                assert -1e-7 <= u and -1e-7 <= v and u + v <= 1 + 1e-7
        assert len(seen) == n


def test_bird_curve_is_hierarchical(oracle):
    """micro-triangle i at level N lies inside micro-triangle i>>2 at level N-1 (what makes 4:1 downsampling valid)."""
    a, b = (C.c_float * 6)(), (C.c_float * 6)()
    for level in range(1, 5):
        for i in range(4 ** level):
            oracle.dll.orc_index2bary(i, level, a)
            oracle.dll.orc_index2bary(i >> 2, level - 1, b)
            cx, cy = (a[0] + a[2] + a[4]) / 3, (a[1] + a[3] + a[5]) / 3
            def side(p, q):
                return (q[0] - p[0]) * (cy - p[1]) - (q[1] - p[1]) * (cx - p[0])
            P = [(b[0], b[1]), (b[2], b[3]), (b[4], b[5])]
            s = [side(P[k], P[(k + 1) % 3]) for k in range(3)]
            assert all(v > 0 for v in s) or all(v < 0 for v in s)


def test_split_bird_decode_is_the_direct_decode_for_every_index(tmp_path):
    """omm_amd/csrc/classify_device.h splits the bird-curve decode at the 64-group boundary (group word + 4 x 64 table).  Its integer part
    is host-callable: tests/native/bird_check.hip compares it with the direct decode of util/bird.h:73-118 for every micro-triangle index
    of every level 3..12 (22 M cases).  Needs hipcc, no GPU."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "bird_check")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-I" + os.path.join(root, "omm_amd", "csrc"), os.path.join(root, "tests", "native", "bird_check.hip"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "22369600 indices, 0 mismatches" in r.stdout, r.stdout + r.stderr
