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


# ---- the work-item id of SetupWorkItems (bake_cpu_impl.cpp:626-631): std::hash chain, restated in oracle/omm_oracle.c and omm_amd/csrc/vm_id.h ----
def _probe_std_hash(bit_patterns, tmp_path):
    """std::hash<float> of THIS container's libstdc++ (the library the reference links on Linux), through tests/native/std_hash_probe.cpp"""
    import subprocess
    exe = str(tmp_path / "std_hash_probe")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "std_hash_probe.cpp")
    subprocess.run(["g++", "-O1", "-o", exe, src], check=True)
    out = subprocess.run([exe] + ["%08x" % b for b in bit_patterns], check=True, capture_output=True, text=True).stdout.split()
    return {int(out[2 * i], 16): int(out[2 * i + 1], 16) for i in range(len(bit_patterns))}, out[-2:]


def test_std_hash_float_restatement_against_the_real_libstdcxx(oracle, tmp_path):
    f = oracle.dll.oracle_std_hash_float
    f.restype = C.c_uint64; f.argtypes = [C.c_float]
    rng = np.random.default_rng(5)
    bits = [0x00000000, 0x80000000, 0x3F800000, 0x3E800000, 0x3F400000, 0xBC8869F9, 0x97C1D70F, 0x00000001, 0x7F7FFFFF, 0xFF7FFFFF] + [int(v) for v in rng.integers(0, 1 << 32, 200, dtype=np.uint64)]
    bits = [b for b in bits if (b & 0x7F800000) != 0x7F800000]   # finite (NaN / Inf triangles are invalid and never hashed)
    ref, ints = _probe_std_hash(bits, tmp_path)
    for b in bits:
        v = np.array([b], np.uint32).view(np.float32)[0]
        assert f(float(v)) == ref[b], hex(b)
    assert ints == ["fffffffffffffffd", "0000000000000007"]   # std::hash<int32_t>: the value converted to size_t (what oracle_vm_id adds for level and format)


def _vm_id(oracle, uv6, level, fmt):
    g = oracle.dll.oracle_vm_id
    g.restype = C.c_uint64; g.argtypes = [C.POINTER(C.c_float), C.c_int32, C.c_int32]
    return g((C.c_float * 6)(*[float(v) for v in uv6]), level, fmt)


def test_colliding_uv_points_share_a_work_item_id_and_a_micro_map(oracle):
    """tests/golden/vmid_collisions.json: different UV points with the same std::hash<glm::vec2> (found by tests/native/vmid_collision_search.c).  Triangles that
    differ only in such a point have the same id, and the reference's map -- keyed by the id alone, bake_cpu_impl.cpp:633-649 -- makes them one work item."""
    import ommtest as ot
    pairs = json.load(open(os.path.join(GOLDEN, "vmid_collisions.json")))["pairs"]
    f = lambda h: np.array([int(h, 16)], np.uint32).view(np.float32)[0]
    assert len(pairs) >= 3
    for pr in pairs:
        a, b = (f(pr["a"][0]), f(pr["a"][1])), (f(pr["b"][0]), f(pr["b"][1]))
        assert a != b
        for pos in range(3):
            t0 = [0.1, 0.2, 0.6, 0.3, 0.4, 0.8]; t1 = list(t0)
            t0[2 * pos:2 * pos + 2] = a; t1[2 * pos:2 * pos + 2] = b
            assert _vm_id(oracle, t0, 5, 2) == _vm_id(oracle, t1, 5, 2)
            assert _vm_id(oracle, t0, 5, 2) != _vm_id(oracle, t1, 6, 2) and _vm_id(oracle, t0, 5, 2) != _vm_id(oracle, t1, 5, 1)
    # the bake: triangle 1 collides with triangle 0 (one micro-map, triangle 0's), triangle 2 is the same as 1 at another level (its own)
    pr = pairs[0]
    uv = np.array([[0.1, 0.2], [0.6, 0.3], [f(pr["a"][0]), f(pr["a"][1])], [0.1, 0.2], [0.6, 0.3], [f(pr["b"][0]), f(pr["b"][1])],
                   [0.1, 0.2], [0.6, 0.3], [f(pr["b"][0]), f(pr["b"][1])]], np.float32)
    ix = np.arange(9, dtype=np.uint32)
    tex = ot.foliage_texture(8, 256, 256, feature=24)
    b = oracle.create_baker(); t = oracle.create_texture(b, [tex], alpha_cutoff=0.5)
    res = oracle.bake(b, ot.make_desc(t, uv, ix, 4, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=np.array([4, 4, 5], np.uint8), flags=0))
    alone = oracle.bake(b, ot.make_desc(t, uv[3:6], ix[:3], 4, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=0))
    first = oracle.bake(b, ot.make_desc(t, uv[0:3], ix[:3], 4, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=0))
    oracle.destroy_texture(b, t); oracle.destroy_baker(b)
    assert res.index[0] == res.index[1] and res.index[2] != res.index[0]
    # (the merged triangle carries triangle 0's micro-map, which is not the one it would get on its own)
    if res.index[0] >= 0 and alone.index[0] >= 0:
        assert bytes(first.array_data) != bytes(alone.array_data) or first.index[0] == alone.index[0]
