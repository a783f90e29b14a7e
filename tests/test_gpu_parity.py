"""GPU parity: the HIP library behind the C ABI must reproduce the reference's known answers and be bit-exact
(arrayData, descArray, indexBuffer, histograms, indexFormat) with the oracle on seeded workloads."""
import os
import numpy as np
import pytest
import kat_runner as kr
import ommtest as ot
from kat_cases import CASES, LEAFLET_MIP, LEAFLET_LEVEL
from test_golden_blob import BLOBS, PAIRS, STATS, bake_input_blob, check_against_output_blob
import blobfmt

pytestmark = pytest.mark.gpu
GPU_KATS = CASES  # incl. the near-duplicate-merge KATs (device classification + serial host tail, host_tail.cpp)


@pytest.mark.parametrize("case", GPU_KATS, ids=[c["name"] for c in GPU_KATS])
def test_kat(product, case):
    b = product.create_baker()
    for cfg in (["Default", "AlphaCutoff"] if case["slow"] else ["Default", "TextureAsUNORM8", "AlphaCutoff", "Serialize"]):
        res = kr.run_case(product, b, case, cfg)
        assert res.stats_tuple() == case["expect"], (cfg, "reference line %d" % case["ref"])
    product.destroy_baker(b)


@pytest.mark.parametrize("name,ref,mip_start,num_mip,cutoff,expect", LEAFLET_MIP, ids=[c[0] for c in LEAFLET_MIP])
def test_leaflet_mip(product, name, ref, mip_start, num_mip, cutoff, expect):
    b = product.create_baker()
    for cfg in ("Default", "AlphaCutoff"):
        assert kr.run_leaflet_mip(product, b, mip_start, num_mip, cutoff, cfg).stats_tuple() == expect, cfg
    product.destroy_baker(b)


@pytest.mark.parametrize("name,ref,level,expect", LEAFLET_LEVEL, ids=[c[0] for c in LEAFLET_LEVEL])
def test_leaflet_level(product, name, ref, level, expect):
    b = product.create_baker()
    for cfg in ("Default", "AlphaCutoff"):
        assert kr.run_leaflet_level(product, b, level, cfg).stats_tuple() == expect, cfg
    product.destroy_baker(b)


def test_workload_too_big(product):
    b = product.create_baker()
    assert kr.run_leaflet_level(product, b, 12, "Default", max_workload=512, expect=ot.WORKLOAD_TOO_BIG) is None
    product.destroy_baker(b)


@pytest.mark.parametrize("iname,oname", PAIRS)
def test_golden_blob_bit_exact(product, iname, oname):
    inp = blobfmt.parse_blob(BLOBS[iname])["inputs"][0]
    out = blobfmt.parse_blob(BLOBS[oname])["results"][0]
    res = bake_input_blob(product, inp)
    check_against_output_blob(res, out)
    assert res.stats_tuple() == STATS


# ---------------------------------------------------------------------------------------------
# oracle <-> product, full result arrays
# ---------------------------------------------------------------------------------------------
def both(product, oracle, mips, uv, ix, level, sat=True, zorder=False, cutoff=0.5, expect=ot.SUCCESS, knobs=(), **kw):
    out = []
    for lib in (product, oracle):
        b = lib.create_baker()
        if lib is product:
            for k, v in knobs:
                lib.set_knob(b, k, v)
        t = lib.create_texture(b, mips, alpha_cutoff=cutoff if sat else -1.0, disable_zorder=zorder)
        d = ot.make_desc(t, uv, ix, level, alpha_cutoff=cutoff, **kw)
        out.append(lib.bake(b, d, expect=expect))
        lib.destroy_texture(b, t)
        lib.destroy_baker(b)
    if expect == ot.SUCCESS:
        assert out[0].same_as(out[1]), out[0].diff(out[1])
    return out[0]


NOISE_U8 = None


def noise_u8(n=512):
    global NOISE_U8
    if NOISE_U8 is None or NOISE_U8.shape[0] != n:
        NOISE_U8 = (ot.value_noise(11, n, n, octaves=4, base_cell=32) * 255).astype(np.uint8)
    return NOISE_U8


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("fmt", [ot.FMT_2STATE, ot.FMT_4STATE])
def test_levels_formats(product, oracle, level, fmt):
    uv, ix = ot.random_triangles(100 + level, 200 if level < 7 else 40, 0.05)
    r = both(product, oracle, [noise_u8()], uv, ix, level, fmt=fmt, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    assert r.index.size == ix.size // 3


@pytest.mark.parametrize("addr", [ot.WRAP, ot.MIRROR, ot.CLAMP, ot.BORDER, ot.MIRROR_ONCE])
@pytest.mark.parametrize("pow2", [True, False])
@pytest.mark.parametrize("sat", [True, False])
def test_address_modes(product, oracle, addr, pow2, sat):
    tex = noise_u8()[:, :] if pow2 else np.ascontiguousarray(noise_u8()[:475, :500])
    uv, ix = ot.random_triangles(7 + addr, 150, 0.3, lo=-0.6, hi=1.6)  # triangles cross the texture border
    both(product, oracle, [tex], uv, ix, 4, sat=sat, addr=addr, promo=ot.PROMO_NEAREST, border_alpha=0.7)


@pytest.mark.parametrize("promo", [ot.PROMO_NEAREST, ot.PROMO_FORCE_OPAQUE, ot.PROMO_FORCE_TRANSPARENT])
@pytest.mark.parametrize("fp32", [True, False])
def test_promotion_and_texture_format(product, oracle, promo, fp32):
    tex = ot.value_noise(5, 256, 256, octaves=3, base_cell=16) if fp32 else noise_u8(256)
    uv, ix = ot.random_triangles(21 + promo, 300, 0.08)
    both(product, oracle, [tex], uv, ix, 5, promo=promo, addr=ot.CLAMP, cutoff=0.45)


def test_state_mapping_variants(product, oracle):
    uv, ix = ot.random_triangles(33, 200, 0.1)
    for le, gt in ((ot.T, ot.UO), (ot.O, ot.T), (ot.UT, ot.O), (ot.UO, ot.UT)):
        both(product, oracle, [noise_u8(256)], uv, ix, 4, le=le, gt=gt, addr=ot.WRAP)


def test_nearest_filter(product, oracle):
    uv, ix = ot.random_triangles(41, 300, 0.1)
    for fmt in (ot.FMT_2STATE, ot.FMT_4STATE):
        for promo in (ot.PROMO_NEAREST, ot.PROMO_FORCE_OPAQUE):
            both(product, oracle, [noise_u8(256)], uv, ix, 4, filt=ot.NEAREST, fmt=fmt, promo=promo, addr=ot.CLAMP)


def test_mip_chain(product, oracle):
    base = ot.value_noise(9, 256, 256, octaves=3, base_cell=32)
    mips = [base]
    while mips[-1].shape[0] > 4:
        t = mips[-1]
        mips.append(((t[0::2, 0::2] + t[1::2, 0::2]) + t[0::2, 1::2] + t[1::2, 1::2]) * np.float32(0.25))
    uv, ix = ot.random_triangles(55, 120, 0.15)
    for n in (2, 4, len(mips)):
        both(product, oracle, mips[:n], uv, ix, 5, sat=False, promo=ot.PROMO_NEAREST, addr=ot.WRAP)


def test_degenerate_and_invalid_triangles(product, oracle):
    uv, ix = ot.random_triangles(61, 64, 0.2)
    uv = uv.copy().reshape(-1, 3, 2)
    uv[0:16, 2] = uv[0:16, 1]                      # segments
    uv[16:24, 1] = uv[16:24, 0]; uv[16:24, 2] = uv[16:24, 0]  # points
    uv[24, 1, 0] = np.nan
    uv[25, 0, 1] = np.inf
    uv[26:32, :, 0] = 0.25                         # vertical segments
    uv[32:38, :, 1] = 0.75                         # horizontal segments
    for dyn in (0.0, 2.0):
        both(product, oracle, [ot.kat_texture("circle", 512, 512)], uv.reshape(-1, 2), ix, 6, addr=ot.WRAP, dyn_scale=dyn,
             unresolved=ot.SPECIAL_FUT)


def test_duplicates_and_flags(product, oracle):
    uv, ix = ot.random_triangles(71, 400, 0.02)
    uv = uv.reshape(-1, 3, 2).copy()
    uv[200:300] = uv[0:100]                        # exact UV duplicates -> shared work items
    uv[300:400] = uv[0:100] + np.float32(0.5)      # same content after a half-texture shift of a periodic texture -> digest duplicates
    tex = np.tile(noise_u8(256)[:128, :128], (2, 2))
    for flags in (ot.FLAG_THREADS, ot.FLAG_THREADS | ot.FLAG_NO_DEDUP, ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL, ot.FLAG_FORCE32,
                  ot.FLAG_NO_SPECIAL | ot.FLAG_NO_DEDUP | ot.FLAG_FORCE32):
        both(product, oracle, [tex], uv.reshape(-1, 2), ix, 3, addr=ot.WRAP, flags=flags)


def test_uniform_ut_uo_merge_keeps_first(product, oracle):
    """all-UT and all-UO OMMs share a 3-state digest: later ones adopt the first one's special index (bake_cpu_impl.cpp:374-377,1047-1062)."""
    tex = ot.kat_texture("diag8", 256, 256, 0.0)
    uv, ix = ot.random_triangles(81, 64, 0.2)
    both(product, oracle, [tex], uv, ix, 2, promo=ot.PROMO_NEAREST, addr=ot.WRAP, sat=False)


@pytest.mark.parametrize("count,allow8,force32,expect_fmt", [
    # every case of support/tests/test_omm_indexing.cpp:122-232, at the reference's triangle counts
    (1, False, False, ot.IDX_U16), (127, False, False, ot.IDX_U16), (128, False, False, ot.IDX_U16), (32766, False, False, ot.IDX_U16),
    (32767, False, False, ot.IDX_U16), (32768, False, False, ot.IDX_U32), (65536, False, False, ot.IDX_U32),
    (1, False, True, ot.IDX_U32), (127, False, True, ot.IDX_U32), (128, False, True, ot.IDX_U32), (32766, False, True, ot.IDX_U32),
    (32767, False, True, ot.IDX_U32), (32768, False, True, ot.IDX_U32),
    (1, True, False, ot.IDX_U8), (127, True, False, ot.IDX_U8), (128, True, False, ot.IDX_U16), (32766, True, False, ot.IDX_U16), (65536, True, False, ot.IDX_U32),
    (1, True, True, ot.IDX_U32), (127, True, True, ot.IDX_U32), (32766, True, True, ot.IDX_U32), (65536, True, True, ot.IDX_U32)])
def test_index_formats(product, oracle, count, allow8, force32, expect_fmt):
    """support/tests/test_omm_indexing.cpp:122-232"""
    uv, ix = ot.random_triangles(91, count, 0.3)
    flags = ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL | ot.FLAG_NO_DEDUP | (ot.FLAG_ALLOW8 if allow8 else 0) | (ot.FLAG_FORCE32 if force32 else 0)
    r = both(product, oracle, [ot.kat_texture("checker2", 256, 256)], uv, ix, 1 if count > 1000 else 3, filt=ot.NEAREST, flags=flags, cutoff=0.3, sat=True)
    assert r.index_format == expect_fmt and r.index.size == count


def test_per_triangle_levels_and_dynamic(product, oracle):
    """support/tests/test_subdiv.cpp:80-172 + dynamic subdivision heuristic"""
    n = 300
    uv, ix = ot.random_triangles(95, n, 0.3)
    lv = (ot.hash_u32(np.arange(n) + 5) % 6).astype(np.uint8)
    lv[lv == 5] = 0xF
    flags = ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL | ot.FLAG_FORCE32 | ot.FLAG_NO_DEDUP
    r = both(product, oracle, [ot.kat_texture("checker2", 512, 512)], uv, ix, 2, filt=ot.NEAREST, flags=flags, cutoff=0.3, levels=lv)
    assert sum(c for c, l, f in r.array_hist) == n
    both(product, oracle, [noise_u8()], uv, ix, 7, dyn_scale=2.0, addr=ot.WRAP)
    both(product, oracle, [noise_u8()], uv, ix, 6, dyn_scale=0.7, addr=ot.WRAP, levels=lv)


@pytest.mark.parametrize("count", [5000, 16300, 16500, 40000])
@pytest.mark.parametrize("flags", [ot.FLAG_THREADS, ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL, ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL | ot.FLAG_NO_DEDUP | ot.FLAG_FORCE32])
def test_descriptor_order_and_offsets_both_tail_paths(product, oracle, count, flags):
    """The spatial sort + Serialize (bake_cpu_impl.cpp:1707-1920) on either side of the tail's threshold (tail_kernels.hip: descriptor slots by counting up
    to 16 384 candidate OMMs, key sort + single-pass offset scan above), with mixed per-triangle levels (offsets from the level histogram), exact
    duplicates (emitted once) and uniform items (special indices, or emitted blocks with special indices disabled)."""
    uv, ix = ot.random_triangles(1234 + count, count, 0.004)
    uv = uv.reshape(-1, 3, 2).copy()
    uv[count // 2: count // 2 + count // 10] = uv[0: count // 10]          # exact UV duplicates
    uv[count - count // 10:] = uv[0: count // 10] + np.float32(0.5)           # digest duplicates (periodic texture)
    lv = (ot.hash_u32(np.arange(count) + 17) % 5).astype(np.uint8)          # levels 0..3 and 0xF (= the global level 4)
    lv[lv == 4] = 0xF
    tex = np.tile(noise_u8(256)[:128, :128], (2, 2))
    r = both(product, oracle, [tex], uv.reshape(-1, 2), ix, 4, addr=ot.WRAP, flags=flags, levels=lv)
    assert r.index.size == count


@pytest.mark.parametrize("glob,n_global,n0,n1,n2,n3,n4", [
    # the parameter sets of support/tests/test_subdiv.cpp:255-350 (BakeSubDiv: Mixed, Mixed2, Lvl0Only .. Lvl4Only, LvlGlobalOnly)
    (2, 8, 4, 7, 7, 7, 7), (4, 84, 234, 0, 23, 34, 57), (2, 0, 56, 0, 0, 0, 0), (2, 0, 0, 526, 0, 0, 0), (2, 0, 0, 0, 91, 0, 0), (2, 0, 0, 0, 0, 391, 0), (2, 0, 0, 0, 0, 0, 391), (4, 430, 0, 0, 0, 0, 0)])
def test_subdiv_level_histograms(product, oracle, glob, n_global, n0, n1, n2, n3, n4):
    """support/tests/test_subdiv.cpp:80-172: per-triangle subdivision levels (0xF = global), 1-texel checkerboard so that no OMM is uniform,
    Nearest filter, special indices / dedup off.  The numbers the reference test carries are its level distributions: the descriptor
    histogram must hold exactly that many OMMs per level (ValidateDesc, :55-78), descriptors come out in the spatial order with the
    level in the key's top bits, and the whole result equals the oracle's."""
    lv = np.array([0xF] * n_global + [0] * n0 + [1] * n1 + [2] * n2 + [3] * n3 + [4] * n4, np.uint8)
    n = lv.size
    lv = lv[np.argsort(ot.hash_u32(np.arange(n) + 32), kind="stable")]                      # a seeded shuffle
    uv, ix = ot.random_triangles(32, n, 0.6)
    yy, xx = np.mgrid[0:1024, 0:1024]
    tex = ((xx % 2) == (yy % 2)).astype(np.float32)                                          # test_subdiv.cpp:88-93
    flags = ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL | ot.FLAG_FORCE32 | ot.FLAG_NO_DEDUP
    r = both(product, oracle, [tex], uv, ix, glob, filt=ot.NEAREST, addr=ot.CLAMP, flags=flags, cutoff=0.3, levels=lv, sat=True)
    want = {0: n0, 1: n1, 2: n2, 3: n3, 4: n4}
    want[glob] += n_global
    assert {int(l): int(c) for c, l, f in r.array_hist} == {l: c for l, c in want.items() if c}
    assert len(r.descs) == n and r.index_format == ot.IDX_U32
    assert np.all(np.diff(r.descs[:, 1]) <= 0)                                               # std::sort(greater) on (level << 60 | morton): high levels first


def test_rejection_threshold(product, oracle):
    uv, ix = ot.random_triangles(97, 300, 0.06)
    both(product, oracle, [noise_u8()], uv, ix, 4, addr=ot.WRAP, rejection=0.6)


def test_uv_formats_and_strides(product, oracle):
    uv, ix = ot.random_triangles(99, 120, 0.1)
    for f in ("fp16", "unorm16"):
        packed, fmt = kr.pack_uv(uv, f)
        both(product, oracle, [noise_u8(256)], packed, ix, 4, uv_format=fmt, addr=ot.CLAMP)
    both(product, oracle, [noise_u8(256)], uv, ix.astype(np.uint16), 4, addr=ot.CLAMP)
    both(product, oracle, [noise_u8(256)], uv[:255], (np.arange(300) % 255).astype(np.uint8), 4, addr=ot.CLAMP)


def test_error_paths(product, oracle):
    """support/tests/test_omm_log.cpp:146-209, test_omm_bake_cpu.cpp:783-789"""
    for lib in (product, oracle):
        msgs = []
        b = lib.create_baker(callback=lambda sev, msg, user: msgs.append((sev, msg.decode())))
        d = ot.default_bake_desc()
        assert lib.bake(b, d, expect=ot.INVALID_ARGUMENT) is None
        assert msgs[-1] == (3, "[Invalid Argument] - ommCpuBakeInputDesc has no texture set")
        t = lib.create_texture(b, [noise_u8(256)], alpha_cutoff=0.3)
        uv, ix = ot.random_triangles(1, 8, 0.1)
        d = ot.make_desc(t, uv, ix, 13, alpha_cutoff=0.3)
        assert lib.bake(b, d, expect=ot.INVALID_ARGUMENT) is None
        assert msgs[-1][1] == "[Invalid Argument] - maxSubdivisionLevel (13) is greater than maximum supported (12)"
        d = ot.make_desc(t, uv, ix, 4, alpha_cutoff=0.4)
        assert lib.bake(b, d, expect=ot.INVALID_ARGUMENT) is None
        assert msgs[-1][1] == "[Invalid Argument] - Texture object alpha cutoff threshold (0.300000) is different from alpha cutoff threshold in bake input (0.400000)"
        d = ot.make_desc(t, uv, ix, 4, alpha_cutoff=0.3, fmt=ot.FMT_2STATE, le=ot.UO)
        assert lib.bake(b, d, expect=ot.INVALID_ARGUMENT) is None
        assert msgs[-1][1] == "[Invalid Argument] - alphaCutoffLessEqual=UnknownOpaque is not compatible with OC1_2_State"
        d = ot.make_desc(t, uv, ix, 4, alpha_cutoff=0.3)
        d.indexFormat = 3
        assert lib.bake(b, d, expect=ot.INVALID_ARGUMENT) is None
        assert msgs[-1][1] == "[Invalid Argument] - indexFormat is not set"
        lib.destroy_texture(b, t)
        lib.destroy_baker(b)


# ---------------------------------------------------------------------------------------------
# full-size properties (no oracle needed)
# ---------------------------------------------------------------------------------------------
def test_properties_at_scale(product):
    """100k triangles / 2K texture / level 6 / 4-state (BASELINE config 1): size-independent invariants."""
    n = 100000
    tex = ot.foliage_texture(3, 2048, 2048)
    uv, ix = ot.random_triangles(5, n, 12.0 / 2048)
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    r1 = product.bake(b, d)
    r2 = product.bake(b, d)
    assert r1.same_as(r2)                                       # idempotent / deterministic
    descs = r1.descs
    assert np.all(descs[:, 1] == 6) and np.all(descs[:, 2] == 2)
    assert np.array_equal(descs[:, 0], np.arange(len(descs)) * 1024)   # contiguous 4^6*2/8-byte blocks
    assert r1.array_data.size == len(descs) * 1024
    used = np.unique(r1.index[r1.index >= 0])
    assert used.size == len(descs)                              # every descriptor referenced, none dangling
    assert sum(c for c, l, f in r1.array_hist) == len(descs)
    assert sum(c for c, l, f in r1.index_hist) == int((r1.index >= 0).sum())
    blocks = r1.array_data.reshape(len(descs), 1024)
    assert len({bytes(x) for x in blocks}) == len(blocks)      # exact dedup left no two equal blocks (all of them)
    # baking a permutation of the triangles yields the same multiset of OMM blocks
    perm = np.argsort(ot.hash_u32(np.arange(n) + 17), kind="stable")
    uvp = uv.reshape(n, 3, 2)[perm].reshape(-1, 2)
    d2 = ot.make_desc(t, uvp, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    r3 = product.bake(b, d2)
    assert r3.array_data.size == r1.array_data.size
    assert sorted(map(bytes, r3.array_data.reshape(-1, 1024))) == sorted(map(bytes, blocks))   # the same multiset of OMM blocks
    s1, s3 = r1.stats_tuple(), r3.stats_tuple()
    assert s1 == s3
    product.destroy_texture(b, t)
    product.destroy_baker(b)


def omm_of_triangle(res, t, level_bytes):
    """special index (negative) or the packed block of triangle t"""
    v = int(res.index[t])
    if v < 0:
        return v
    off = int(res.descs[v][0])
    return bytes(res.array_data[off:off + level_bytes])


def test_full_size_bake_matches_oracle_on_a_subset(product, oracle):
    """BASELINE metric configuration at full size (1 M triangles, 4K foliage alpha, level 8, 4-state: 6.5e10 micro-triangles,
    more than 2^32 GPU threads).  Baking is per-triangle independent, so every triangle's OMM in the full bake must equal the
    OMM the oracle computes for it in a small bake of the first triangles."""
    n, k = 1000000, 1500
    tex = ot.foliage_texture(1234, 4096, 4096, feature=64)
    uv, ix = ot.random_triangles(1235, n, 8.0 / 4096)
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    full = product.bake(b, ot.make_desc(t, uv, ix, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE), want_stats=False)
    product.destroy_texture(b, t)
    product.destroy_baker(b)
    assert full.index.size == n and len(full.descs) > n // 50, "suspiciously few OMMs: %d" % len(full.descs)
    assert full.array_data.size == len(full.descs) * 16384
    ob = oracle.create_baker()
    otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
    small = oracle.bake(ob, ot.make_desc(otx, uv[:3 * k], ix[:3 * k], 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE), want_stats=False)
    oracle.destroy_texture(ob, otx)
    oracle.destroy_baker(ob)
    for tri in list(range(k)):
        assert omm_of_triangle(full, tri, 16384) == omm_of_triangle(small, tri, 16384), tri
    # the last triangles too (highest work-item ids = last tiles of the queue): against an ORACLE bake of just them
    ob = oracle.create_baker()
    otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
    tail = oracle.bake(ob, ot.make_desc(otx, uv[3 * (n - k):], ix[:3 * k], 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE), want_stats=False)
    oracle.destroy_texture(ob, otx)
    oracle.destroy_baker(ob)
    for j in range(k):
        assert omm_of_triangle(full, n - k + j, 16384) == omm_of_triangle(tail, j, 16384), j


def block_of_triangle(res, t):
    """special index (negative) or (level, packed block) of triangle t, for results with mixed levels"""
    v = int(res.index[t])
    if v < 0:
        return v
    off, lvl, fmt = (int(x) for x in res.descs[v])
    nbytes = max(1, (4 ** lvl) * (2 if fmt == 2 else 1) // 8)
    return lvl, bytes(res.array_data[off:off + nbytes])


def test_full_size_sharded_bake_8_ranks(product):
    """BASELINE configs[3] at FULL size (1 M triangles, 4K alpha, level 8, 1.27 GB arrayData) with the 8 ranks simulated on one GPU: the
    sharded protocol (per-rank classification of an eighth of the active items, metadata merge, replicated tail, block exchange, chunk-wise
    scatter) must leave EVERY rank with exactly the single-GPU result.  (The collectives are emulated by host copies here; the RCCL calls
    themselves run in test_cpp_user_of_the_rccl_entry_point.)"""
    import hashlib
    hip = ot.Hip()
    n, world = 1000000, 8
    tex = ot.foliage_texture(1234, 4096, 4096, feature=64)
    uv, ix = ot.random_triangles(1235, n, 8.0 / 4096)
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    ref = product.bake(b, d, want_stats=False)
    assert ref.array_data.size > 1 << 30 and len(ref.descs) > 50000
    digest = lambda r: hashlib.sha256(r.array_data.tobytes() + r.desc_bytes + r.index.tobytes()).hexdigest()
    want = digest(ref)
    per_rank = ot.bake_sharded_simulated(product, hip, b, d, uv, ix.astype(np.int32), world)
    product.destroy_texture(b, t)
    product.destroy_baker(b)
    for r, res in enumerate(per_rank):
        assert res.same_as(ref) and digest(res) == want, "rank %d: %s" % (r, res.diff(ref))


def test_full_size_mixed_levels_config4(product, oracle):
    """BASELINE configs[4] at FULL size on one GPU: 4 M triangles, per-triangle levels U{4..10} for 75 % and the dynamic heuristic (scale 2,
    max level 10) for 25 %, 8192^2 alpha, dedup on -- 6e11 micro-triangle slots, all levels >= 6 drained by one persistent launch.
    Checked: every one of the first and last 400 triangles against oracle bakes of just those, plus the size-independent invariants."""
    import xxhash
    n, k = 4000000, 400
    tex = ot.foliage_texture(4321, 8192, 8192, feature=96)
    uv, ix = ot.random_triangles(9, n, 3.0 / 8192)
    h = ot.hash_u32(np.arange(n) + 9000)
    lv = (4 + (h >> 8) % 7).astype(np.uint8); lv[(h & 3) == 0] = 0xF
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    kw = dict(addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, dyn_scale=2.0)
    d = ot.make_desc(t, uv, ix, 10, levels=lv, **kw)
    import bench, ctypes
    # default transfer of a bake that may use threads, on a host with >= 6 CPUs: the finished array crosses PCIe as a codec stream and is expanded by the
    # baker's helper threads (round 5) -- the bytes must be those of the device-resident entry
    full = product.bake(b, d, want_stats=False)
    tm = bench.get_timings(product, b)
    assert tm.resultTransfer in (ot.TRANSFER_COMPRESSED, ot.TRANSFER_STREAMED), tm.resultTransfer   # (the automatic choice: a classification this long hides a streamed copy)
    # the compressed transfer: the finished array crosses PCIe as a codec stream and is expanded by the baker's helper threads (round 5)
    product.set_knob(b, ot.KNOB_RESULT_TRANSFER, ot.TRANSFER_COMPRESSED)
    comp = product.bake(b, d, want_stats=False)
    tm = bench.get_timings(product, b)
    assert tm.resultTransfer == ot.TRANSFER_COMPRESSED and 0 < tm.compressedBytes < full.array_data.size // 4 and tm.expandThreads >= 1, (tm.resultTransfer, tm.compressedBytes, tm.expandThreads)
    assert comp.same_as(full), comp.diff(full)
    del comp
    # the streamed transfer (rounds 3 - 4: blocks placed and copied while the classification runs, tail_kernels.hip "Streamed result"): it must not have fallen
    # back -- at this size a family of possible duplicates once spanned two levels and pulled level-10 items behind their own placement
    product.set_knob(b, ot.KNOB_RESULT_TRANSFER, ot.TRANSFER_STREAMED)
    streamed = product.bake(b, d, want_stats=False)
    tm = bench.get_timings(product, b)
    assert tm.resultTransfer == ot.TRANSFER_STREAMED and tm.streamChunks > 1 and tm.streamedBytes == full.array_data.size, (tm.resultTransfer, tm.streamChunks, tm.streamedBytes)
    assert streamed.same_as(full), streamed.diff(full)
    del streamed
    dev = ot.bake_device(product, ot.Hip(), b, d, uv, ix, levels=lv)
    assert dev.same_as(full), dev.diff(full)
    del dev
    product.destroy_texture(b, t)
    product.destroy_baker(b)
    assert full.index.size == n and len(full.descs) > 30000 and full.array_data.size > (2 << 30)
    levels_seen = {int(l) for c, l, f in full.array_hist}
    assert levels_seen >= {4, 5, 6, 7, 8, 9, 10}, levels_seen
    # invariants: contiguous offsets in descriptor order, every descriptor referenced, histograms consistent, no two equal blocks per level
    sizes = np.maximum(1, (4 ** full.descs[:, 1]) * 2 // 8)
    assert np.array_equal(full.descs[:, 0], np.concatenate([[0], np.cumsum(sizes)[:-1]])) and full.array_data.size == int(sizes.sum())
    assert np.unique(full.index[full.index >= 0]).size == len(full.descs)
    assert sum(c for c, l, f in full.array_hist) == len(full.descs) and sum(c for c, l, f in full.index_hist) == int((full.index >= 0).sum())
    seen = set()
    for (off, lvl, fmt), sz in zip(full.descs, sizes):
        key = (int(lvl), xxhash.xxh3_128_digest(full.array_data[int(off):int(off) + int(sz)]))
        assert key not in seen
        seen.add(key)
    # per-triangle parity with the oracle (baking is independent per triangle; dedup changes offsets, never block contents)
    for lo in (0, n - k):
        ob = oracle.create_baker()
        otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
        small = oracle.bake(ob, ot.make_desc(otx, uv[3 * lo:3 * (lo + k)], ix[:3 * k], 10, levels=lv[lo:lo + k], **kw), want_stats=False)
        oracle.destroy_texture(ob, otx)
        oracle.destroy_baker(ob)
        for j in range(k):
            assert block_of_triangle(full, lo + j) == block_of_triangle(small, j), (lo, j)


@pytest.mark.parametrize("offset", [3.0, -7.0, 255.0, 4097.0, 20000.0, -70000.0])
@pytest.mark.parametrize("addr", [ot.WRAP, ot.MIRROR])
def test_large_uv_offsets(product, oracle, offset, addr):
    """far-away UV periods: fp32 texel resolution degrades (ulp(20000) is a texel at 512^2); the hierarchical shortcut must
    stay conservative there (region_state grows its boxes by 64 ulp, and gives up beyond |uv| = 16384)"""
    uv, ix = ot.random_triangles(123, 200, 0.05)
    uv = (uv + np.float32(offset)).astype(np.float32)
    both(product, oracle, [noise_u8()], uv, ix, 5, addr=addr, promo=ot.PROMO_FORCE_OPAQUE)


@pytest.mark.parametrize("offset", [16382.0, 16383.9, 16384.5, -16383.5, -16386.0])
def test_uv_magnitude_threshold_of_the_single_texel_pass(product, oracle, offset):
    """work items on either side of |uv| = 16384: below it the straight-line single-texel pass runs (three-operand min / max, plain float -> int
    conversions under its FINITE precondition), above it the generic path -- the result is the oracle's either way; 4096^2 texture, so the pixel
    coordinates reach 2^26"""
    uv, ix = ot.random_triangles(321, 60, 0.8)
    uv = (uv + np.float32(offset)).astype(np.float32)
    tex = ot.foliage_texture(11, 4096, 4096, feature=64)
    both(product, oracle, [tex], uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)


def test_hierarchy_levels_all_paths(product, oracle):
    """big triangles (many tiles per item, most of them uniform) and tiny ones, every level 0..9, SAT on"""
    tex = ot.foliage_texture(77, 1024, 1024, feature=96)
    for level in range(0, 10):
        for extent in (0.02, 0.3):
            uv, ix = ot.random_triangles(500 + level, 24 if level > 7 else 80, extent)
            both(product, oracle, [tex], uv, ix, level, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
            both(product, oracle, [tex], uv, ix, level, addr=ot.CLAMP, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL)


@pytest.mark.parametrize("level,extent,count", [(10, 0.02, 24), (10, 0.4, 6), (11, 0.03, 10), (11, 0.5, 3), (12, 0.05, 5), (12, 0.7, 2)])
def test_highest_levels(product, oracle, level, extent, count):
    """levels 10..12 (up to 16.7 M micro-triangles and 4096 tiles per work item): tile queue records, split bird decode with 9 high bits,
    tiny and texture-sized triangles, both formats"""
    tex = ot.foliage_texture(91, 2048, 2048, feature=80)
    uv, ix = ot.random_triangles(7000 + level, count, extent)
    both(product, oracle, [tex], uv, ix, level, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    both(product, oracle, [tex], uv, ix, level, fmt=ot.FMT_2STATE, addr=ot.MIRROR, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL | ot.FLAG_NO_DEDUP)


def test_device_resident_entry_point(product, oracle):
    """ommxBakeDevice: inputs and outputs stay in HBM; results must equal ommCpuBake's and the oracle's."""
    hip = ot.Hip()
    tex = noise_u8()
    n = 3000
    uv, ix = ot.random_triangles(321, n, 0.04)
    lv = (ot.hash_u32(np.arange(n) + 9) % 8).astype(np.uint8)
    lv[lv == 7] = 0xF
    for kwargs, levels in ((dict(addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE), None),
                           (dict(addr=ot.CLAMP, promo=ot.PROMO_NEAREST, dyn_scale=1.5, flags=ot.FLAG_THREADS | ot.FLAG_FORCE32), lv)):
        ob = oracle.create_baker()
        otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
        ref = oracle.bake(ob, ot.make_desc(otx, uv, ix, 6, levels=levels, **kwargs))
        oracle.destroy_texture(ob, otx)
        oracle.destroy_baker(ob)
        b = product.create_baker()
        t = product.create_texture(b, [tex], alpha_cutoff=0.5)
        d = ot.make_desc(t, uv, ix, 6, levels=levels, **kwargs)
        host = product.bake(b, d)
        dev = ot.bake_device(product, hip, b, d, uv, ix, levels)
        product.destroy_texture(b, t)
        product.destroy_baker(b)
        assert host.same_as(ref), host.diff(ref)
        assert dev.same_as(ref), dev.diff(ref)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_sharded_bake_equals_single_gpu(product, oracle, world):
    """ommxSharded* (multi-GPU protocol) with all ranks simulated on one GPU: every rank must end with exactly the single-GPU result."""
    hip = ot.Hip()
    tex = ot.foliage_texture(5, 1024, 1024, feature=48)
    n = 4000
    uv, ix = ot.random_triangles(808, n, 0.02)
    uv = uv.reshape(-1, 3, 2).copy(); uv[3000:3500] = uv[0:500]; uv = uv.reshape(-1, 2)     # UV duplicates
    lv = (ot.hash_u32(np.arange(n) + 3) % 8).astype(np.uint8)                                 # levels 0..7 mixed
    for kwargs in (dict(addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE), dict(addr=ot.WRAP, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL, rejection=0.3)):
        ob = oracle.create_baker()
        otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
        ref = oracle.bake(ob, ot.make_desc(otx, uv, ix, 7, levels=lv, **kwargs))
        oracle.destroy_texture(ob, otx)
        oracle.destroy_baker(ob)
        b = product.create_baker()
        t = product.create_texture(b, [tex], alpha_cutoff=0.5)
        d = ot.make_desc(t, uv, ix, 7, levels=lv, **kwargs)
        per_rank = ot.bake_sharded_simulated(product, hip, b, d, uv, ix, world, levels=lv)
        product.destroy_texture(b, t)
        product.destroy_baker(b)
        for r, res in enumerate(per_rank):
            assert res.same_as(ref), "rank %d/%d: %s" % (r, world, res.diff(ref))


def test_sharded_scatter_in_small_chunks(product, oracle):
    """The gathered contributions are scattered chunk by chunk (the RCCL path receives them that way).  With 4 KiB chunks every block of a
    level >= 7 item (4 KiB and more) straddles chunk boundaries and is placed in pieces: 3 simulated ranks must still reproduce the oracle."""
    hip = ot.Hip()
    tex = ot.foliage_texture(9, 1024, 1024, feature=40)
    n = 1500
    uv, ix = ot.random_triangles(4242, n, 0.03)
    lv = (5 + ot.hash_u32(np.arange(n) + 11) % 4).astype(np.uint8)          # levels 5..8: blocks of 256 B .. 16 KiB
    kwargs = dict(addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    ob = oracle.create_baker()
    otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
    ref = oracle.bake(ob, ot.make_desc(otx, uv, ix, 8, levels=lv, **kwargs))
    oracle.destroy_texture(ob, otx)
    oracle.destroy_baker(ob)
    b = product.create_baker()
    product.set_knob(b, ot.KNOB_SHARD_CHUNK_BYTES, 4352)
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, 8, levels=lv, **kwargs)
    per_rank = ot.bake_sharded_simulated(product, hip, b, d, uv, ix, 3, levels=lv)
    product.destroy_texture(b, t)
    product.destroy_baker(b)
    assert len(ref.array_data) > 200000
    for r, res in enumerate(per_rank):
        assert res.same_as(ref), "rank %d: %s" % (r, res.diff(ref))


def test_cpp_user_of_the_rccl_entry_point(tmp_path):
    """examples/sharded_rccl.cpp: a C++ program (no Python, no torch) creates an RCCL communicator through the library, runs
    ommxShardedBakeRccl -- ncclAllReduce / ncclAllGather issued by the library on its own streams -- and compares the merged result
    byte for byte with ommxBakeDevice.  One rank here (one GPU per box); the same binary takes <rank> <world> <id-file> on a node."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "omm_amd", "lib")
    exe = str(tmp_path / "sharded_rccl")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "sharded_rccl.cpp"), "-o", exe,
                        "-L" + lib_dir, "-lomm-lib", "-Wl,-rpath," + lib_dir], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded == single-GPU" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("level,fmt,n", [(5, ot.FMT_4STATE, 300), (6, ot.FMT_2STATE, 300), (7, ot.FMT_4STATE, 200), (9, ot.FMT_4STATE, 40), (8, ot.FMT_2STATE, 60),
                                         (7, ot.FMT_4STATE, 70000)])
def test_item_digests_are_xxh64_of_the_state_bytes(product, oracle, level, fmt, n):
    """CalcDigest (bake_cpu_impl.cpp:374-377, 1038-1040): the digest of a work item is XXH64(seed 42) over one byte per micro-triangle with UT folded
    into UO.  The device has several digest kernels (small items; 256-byte chunks through LDS; for few, long items a form that splits the round into a chain wave and
    three producer waves -- chosen by item size and by how many items a launch has, the last case forces the many-items form); ranks of a sharded bake may pick different ones for the same level, so each must produce
    THE hash, not just a consistent one.  The per-item digests are visible in the metadata words of the four-phase sharded API."""
    import ctypes as C
    import omm_amd.sharded as sh
    hip = ot.Hip()
    oracle.dll.orc_xxh64.restype = C.c_uint64
    oracle.dll.orc_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    tex = ot.foliage_texture(17, 1024, 1024, feature=24)
    uv, ix = ot.random_triangles(900 + level, n, 60.0 / 1024 if n < 1000 else 24.0 / 1024)
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, level, fmt=fmt, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    ref = product.bake(b, d, want_stats=False)
    bits = 2 if fmt == ot.FMT_4STATE else 1
    want = set()
    # the blocks of the result -> state bytes -> XXH64
    M = 4 ** level
    nbytes = max(1, (M * bits) // 8)
    for e in range(len(ref.descs)):
        off = int(ref.descs[e][0])
        blk = ref.array_data[off:off + nbytes]
        if bits == 2:
            st = np.stack([(blk >> (2 * k)) & 3 for k in range(4)], axis=1).reshape(-1)[:M].astype(np.uint8)
            st[st == 2] = 3
        else:
            st = np.unpackbits(blk, bitorder="little")[:M].astype(np.uint8)
        raw = st.tobytes()
        want.add(int(oracle.dll.orc_xxh64(raw, len(raw), 42)))
    assert len(want) > (n // 4 if n < 1000 else 10000), (len(want), len(ref.descs))
    dll = sh.bind(product.dll)
    d_uv, d_ix = hip.upload(uv), hip.upload(ix.astype(np.int32))
    dd = ot.BakeInputDesc.from_buffer_copy(d); dd.texCoords, dd.indexBuffer = d_uv, d_ix
    h = C.c_void_p()
    assert dll.ommxShardedBegin(b, C.byref(dd), 0, 1, C.byref(h)) == ot.SUCCESS
    w, nw = C.c_void_p(), C.c_uint64()
    assert dll.ommxShardedGetMeta(h, C.byref(w), C.byref(nw)) == ot.SUCCESS
    words = hip.download(w, 4 * nw.value, np.uint32).reshape(4, -1)      # [mask | known | digest lo | digest hi] x active items (tail_kernels.hip: shard_pack_meta)
    assert dll.ommxShardedDestroy(h) == ot.SUCCESS
    assert n < 1000 or words.shape[1] > 2 * 256 * 64, words.shape                  # (the last case: more than two workgroups of 64 items per CU)
    mixed = (words[0] & (words[0] - 1)) != 0                                  # items with more than one state: the ones that keep a block
    got = set((words[2][mixed].astype(np.uint64) | (words[3][mixed].astype(np.uint64) << np.uint64(32))).tolist())
    missing = want - got
    assert not missing, "%d of %d block hashes are not among the %d item digests" % (len(missing), len(want), len(got))
    assert got == want or len(got) >= len(want)
    hip.free(d_uv); hip.free(d_ix)
    product.destroy_texture(b, t)
    product.destroy_baker(b)


@pytest.mark.parametrize("world", [2, 3])
def test_one_call_sharded_bake_over_caller_collectives(world):
    """ommxShardedBakeRccl at world_size > 1: real processes sharing GPU 0, the library's own sequence (status agreement, metadata merge, codec
    streams at rank offsets / raw chunks, scatter, host-tail route, idle ranks) over ommxCommFromCollectives + gloo -- everything
    but the two RCCL calls themselves, which one GPU cannot run with two ranks."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "scripts", "ranks_one_call_gloo_gpu.py"), str(world)], cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert r.returncode == 0 and "one-call sharded bake over caller collectives ok" in r.stdout, r.stdout[-4000:]


@pytest.mark.parametrize("ranks", [1, 4, 7])
def test_cpp_user_of_caller_collectives_threads_as_ranks(tmp_path, ranks):
    """examples/sharded_threads.cpp: a C++ program makes a communicator out of two functions of its own (ommxCommFromCollectives: a barrier plus
    device-to-device copies between threads) and runs the one-call sharded bake with several ranks as threads of ONE process on one GPU -- each rank its
    own baker, all bakes in flight at once -- and compares every rank's result byte for byte with ommxBakeDevice."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "omm_amd", "lib")
    exe = str(tmp_path / "sharded_threads")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-Wall", "-Wextra", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "sharded_threads.cpp"), "-o", exe,
                        "-L" + lib_dir, "-lomm-lib", "-Wl,-rpath," + lib_dir, "-lpthread"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr[-3000:]
    r = subprocess.run([exe, str(ranks)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all == single-GPU" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_torch_distributed_plumbing_one_rank_nccl():
    """omm_amd/sharded.py over a real (1-rank) RCCL process group: raw-pointer tensor views, all_reduce, all_gather_into_tensor, and a
    sharded bake compared with ommCpuBake (the multi-rank exchange itself is covered by test_sharded_bake_equals_single_gpu and the gloo test)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "scripts", "one_rank_nccl.py")], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "one-rank nccl plumbing ok" in r.stdout, r.stdout[-3000:]


def test_near_duplicate_merge_and_size_budget(product, oracle):
    """opt-in lossy reducers (bake_cpu_impl.cpp:1068-1430 LSH merge, :1474-1688 Compress): device classification + host tail vs oracle"""
    tex = ot.kat_texture("hexagons", 1024, 1024)
    from kat_cases import hex_grid
    uv, ix = hex_grid()
    both(product, oracle, [tex], uv, ix, 4, addr=ot.CLAMP, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | ot.FLAG_NEAR_DUP, sat=False)
    both(product, oracle, [tex], uv, ix, 3, addr=ot.CLAMP, promo=ot.PROMO_FORCE_OPAQUE, flags=ot.FLAG_THREADS | ot.FLAG_NEAR_DUP | (1 << 10), sat=True)
    # maxArrayDataSize budgets (distinct coverage-per-byte values: the reference's std::sort tie order is not pinned)
    uv2, ix2 = ot.random_triangles(777, 60, 0.2)
    for budget in (20000, 4000, 600):
        out = []
        for lib in (product, oracle):
            b = lib.create_baker()
            t = lib.create_texture(b, [tex], alpha_cutoff=0.5)
            d = ot.make_desc(t, uv2, ix2, 5, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
            d.maxArrayDataSize = budget
            out.append(lib.bake(b, d))
            lib.destroy_texture(b, t)
            lib.destroy_baker(b)
        assert out[0].same_as(out[1]), (budget, out[0].diff(out[1]))
        assert out[0].array_data.size <= budget or len(out[0].descs) == 0 or budget < 100
    # the same reducers through the device-resident entry point (inputs and result in HBM; serial tail on the host in between)
    hip = ot.Hip()
    for flags, budget in ((ot.FLAG_THREADS | ot.FLAG_NEAR_DUP, 0xFFFFFFFF), (ot.FLAG_THREADS, 4000)):
        ob = oracle.create_baker()
        otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
        d = ot.make_desc(otx, uv2, ix2, 5, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=flags)
        d.maxArrayDataSize = budget
        ref = oracle.bake(ob, d)
        oracle.destroy_texture(ob, otx)
        oracle.destroy_baker(ob)
        b = product.create_baker()
        t = product.create_texture(b, [tex], alpha_cutoff=0.5)
        d = ot.make_desc(t, uv2, ix2, 5, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=flags)
        d.maxArrayDataSize = budget
        dev = ot.bake_device(product, hip, b, d, uv2, ix2)
        product.destroy_texture(b, t)
        product.destroy_baker(b)
        assert dev.same_as(ref), (flags, budget, dev.diff(ref))


# ---------------------------------------------------------------------------------------------
# blob (de)serialisation -- SURVEY.md section 8(f) #1
# ---------------------------------------------------------------------------------------------
class BlobDesc(__import__("ctypes").Structure):
    _fields_ = [("data", __import__("ctypes").c_void_p), ("size", __import__("ctypes").c_uint64)]


class DeserializedDesc(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("flags", _C.c_int), ("numInputDescs", _C.c_int), ("inputDescs", _C.POINTER(ot.BakeInputDesc)),
                ("numResultDescs", _C.c_int), ("resultDescs", _C.POINTER(ot.BakeResultDesc))]


def _bind_serialize(dll):
    import ctypes as C
    dll.ommCpuSerialize.argtypes = [C.c_void_p, C.POINTER(DeserializedDesc), C.POINTER(C.c_void_p)]
    dll.ommCpuGetSerializedResultDesc.argtypes = [C.c_void_p, C.POINTER(C.POINTER(BlobDesc))]
    dll.ommCpuDestroySerializedResult.argtypes = [C.c_void_p]
    dll.ommCpuDeserialize.argtypes = [C.c_void_p, C.POINTER(BlobDesc), C.POINTER(C.c_void_p)]
    dll.ommCpuGetDeserializedDesc.argtypes = [C.c_void_p, C.POINTER(C.POINTER(DeserializedDesc))]
    dll.ommCpuDestroyDeserializedResult.argtypes = [C.c_void_p]


def _deserialize(product, baker, blob_bytes, expect=ot.SUCCESS):
    import ctypes as C
    buf = C.create_string_buffer(blob_bytes, len(blob_bytes))
    bd = BlobDesc(C.cast(buf, C.c_void_p), len(blob_bytes))
    h = C.c_void_p()
    r = product.dll.ommCpuDeserialize(baker, C.byref(bd), C.byref(h))
    assert r == expect, r
    if r != ot.SUCCESS:
        return None, None
    pd = C.POINTER(DeserializedDesc)()
    assert product.dll.ommCpuGetDeserializedDesc(h, C.byref(pd)) == ot.SUCCESS
    return h, pd.contents


def test_golden_blobs_through_the_library_deserializer(product):
    """test_omm_bake_cpu.cpp:2034-2304: every embedded blob (v1.4.0 .. v1.7.0, plain and LZ4) decodes; inputs bake to the golden output"""
    _bind_serialize(product.dll)
    b = product.create_baker()
    golden = blobfmt.parse_blob(BLOBS["output_v1_4_0"])["results"][0]
    for name, blob in BLOBS.items():
        h, dd = _deserialize(product, b, blob)
        if name.startswith("input"):
            assert dd.numInputDescs == 1 and dd.numResultDescs == 0
            res = product.bake(b, dd.inputDescs[0])
            check_against_output_blob(res, golden)
            assert res.stats_tuple() == STATS
        else:
            assert dd.numInputDescs == 0 and dd.numResultDescs == 1
            res = ot.BakeResult(dd.resultDescs[0])
            check_against_output_blob(res, golden)
        assert product.dll.ommCpuDestroyDeserializedResult(h) == ot.SUCCESS
    # truncated blob -> digest mismatch -> INVALID_ARGUMENT (test_omm_bake_cpu.cpp:241-251)
    _deserialize(product, b, BLOBS["input_v1_5_0"][:-4], expect=ot.INVALID_ARGUMENT)
    product.destroy_baker(b)


@pytest.mark.parametrize("compress", [0, 1])
@pytest.mark.parametrize("zorder_disabled,sat", [(False, False), (True, True)])
def test_serialize_round_trip(product, oracle, compress, zorder_disabled, sat):
    """Serialize(input + result) -> independent python parser (tests/blobfmt.py) and -> Deserialize -> re-bake == original"""
    import ctypes as C
    _bind_serialize(product.dll)
    tex = np.ascontiguousarray(noise_u8()[:200, :300])     # non-pow2: Morton storage is padded to 512^2
    uv, ix = ot.random_triangles(4242, 50, 0.2)
    lv = (np.arange(50) % 5).astype(np.uint8)
    b = product.create_baker()
    t = product.create_texture(b, [tex, np.ascontiguousarray(tex[::2, ::2])], alpha_cutoff=0.5 if sat else -1.0, disable_zorder=zorder_disabled)
    d = ot.make_desc(t, uv, ix, 5, addr=ot.MIRROR, promo=ot.PROMO_NEAREST, levels=lv)
    r, out = product.bake_raw(b, d)
    assert r == ot.SUCCESS
    prd = C.POINTER(ot.BakeResultDesc)()
    assert product.fn("ommCpuGetBakeResultDesc")(out, C.byref(prd)) == ot.SUCCESS
    ref = ot.BakeResult(prd.contents)
    dd = DeserializedDesc(compress, 1, C.pointer(d), 1, prd)
    sh = C.c_void_p()
    assert product.dll.ommCpuSerialize(b, C.byref(dd), C.byref(sh)) == ot.SUCCESS
    pb = C.POINTER(BlobDesc)()
    assert product.dll.ommCpuGetSerializedResultDesc(sh, C.byref(pb)) == ot.SUCCESS
    blob = C.string_at(pb.contents.data, pb.contents.size)
    assert product.dll.ommCpuDestroySerializedResult(sh) == ot.SUCCESS
    # independent parse
    oracle.dll.orc_xxh64.restype = C.c_uint64
    oracle.dll.orc_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    parsed = blobfmt.parse_blob(blob, xxh64=lambda dat, s: oracle.dll.orc_xxh64(dat, len(dat), s))
    assert parsed["version"] == (1, 9, 0, 5)
    pin = parsed["inputs"][0]
    assert np.array_equal(pin["texture"]["mips"][0], tex) and pin["texture"]["tiling"] == (0 if zorder_disabled else 1)
    assert pin["texCoords"] == uv.tobytes() and pin["indexBuffer"] == ix.tobytes() 
    assert len(pin["subdivisionLevels"]) == 150 and pin["subdivisionLevels"][:50] == lv.tobytes()   # count is indexCount (serialize_impl.cpp:147)
    assert parsed["results"][0]["arrayData"] == ref.array_data.tobytes() and parsed["results"][0]["descArray"] == ref.desc_bytes
    # library deserializer + re-bake
    h, d2 = _deserialize(product, b, blob)
    assert d2.numInputDescs == 1 and d2.numResultDescs == 1
    again = product.bake(b, d2.inputDescs[0])
    assert again.same_as(ref), again.diff(ref)
    assert ot.BakeResult(d2.resultDescs[0]).same_as(ref)
    assert product.dll.ommCpuDestroyDeserializedResult(h) == ot.SUCCESS
    assert product.fn("ommCpuDestroyBakeResult")(out) == ot.SUCCESS
    product.destroy_texture(b, t)
    product.destroy_baker(b)


# ---------------------------------------------------------------------------------------------
# BASELINE.json configurations (configs[0], [1], [4]; configs[2]/[3] are the bench workload, covered above)
# ---------------------------------------------------------------------------------------------
def test_baseline_config0_quad_checker(product, oracle):
    """configs[0]: single quad (2 tris), 256x256 32-px checkerboard (FP32), level 4, 2-state"""
    yy, xx = np.mgrid[0:256, 0:256]
    tex = (((xx // 32) + (yy // 32)) & 1).astype(np.float32)
    uv = np.array([[0, 0], [1, 0], [0, 1], [1, 0], [1, 1], [0, 1]], np.float32)
    ix = np.arange(6, dtype=np.uint32)
    for sat in (True, False):
        r = both(product, oracle, [tex], uv, ix, 4, fmt=ot.FMT_2STATE, addr=ot.WRAP, sat=sat)
        assert r.index.size == 2 and len(r.descs) >= 1 and np.all(r.descs[:, 2] == 1)


def test_baseline_config1_full_parity(product, oracle):
    """configs[1]: 100k random-UV triangles, 2K multi-octave noise alpha, level 6, 4-state -- the WHOLE result, bit for bit"""
    n = 100000
    tex = (ot.value_noise(77, 2048, 2048, octaves=5, base_cell=128) * 255).astype(np.uint8)
    uv, ix = ot.random_triangles(78, n, 10.0 / 2048)
    r = both(product, oracle, [tex], uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    assert r.index.size == n and len(r.descs) > 1000


def c4_workload(n, seed, tex_size):
    """configs[4] shape: per-triangle levels U{4..10} for 75 % of the triangles, 0xF (dynamic heuristic, scale 2, max 10) for the rest"""
    uv, ix = ot.random_triangles(seed, n, 6.0 / tex_size)
    h = ot.hash_u32(np.arange(n) + 1000 * seed)
    lv = (4 + (h >> 8) % 7).astype(np.uint8)
    lv[(h & 3) == 0] = 0xF
    return uv, ix, lv


def omm_by_triangle(res, t):
    v = int(res.index[t])
    if v < 0:
        return v
    off, lvl, fmt = (int(x) for x in res.descs[v])
    return (lvl, fmt, bytes(res.array_data[off:off + max(1, ((4 ** lvl) * fmt) // 8)]))


def test_baseline_config4_mixed_levels_dynamic_8k(product, oracle):
    """configs[4] at reduced triangle count: full parity on 2500 triangles, then a 300k-triangle bake whose per-triangle OMMs must
    equal the small bakes' (dedup on, so equal content may live at different offsets)"""
    tex = ot.foliage_texture(4321, 8192, 8192, feature=96)
    k, n = 2500, 300000
    uv, ix, lv = c4_workload(n, 9, 8192)
    kw = dict(addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=None, dyn_scale=2.0)
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    ob = oracle.create_baker()
    otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
    kw["levels"] = lv[:k]
    small_p = product.bake(b, ot.make_desc(t, uv[:3 * k], ix[:3 * k], 10, **kw))
    small_o = oracle.bake(ob, ot.make_desc(otx, uv[:3 * k], ix[:3 * k], 10, **kw))
    assert small_p.same_as(small_o), small_p.diff(small_o)
    assert len({int(d[1]) for d in small_p.descs}) >= 5           # several subdivision levels really present
    kw["levels"] = lv
    full = product.bake(b, ot.make_desc(t, uv, ix, 10, **kw), want_stats=False)
    assert full.index.size == n
    for tri in range(k):
        assert omm_by_triangle(full, tri) == omm_by_triangle(small_o, tri), tri
    # descriptor array is sorted by (level desc, morton) and offsets are contiguous per-block sizes (bake_cpu_impl.cpp:1742-1800)
    sizes = np.maximum(1, (4 ** full.descs[:, 1]) * full.descs[:, 2] // 8)
    assert np.array_equal(full.descs[:, 0], np.concatenate([[0], np.cumsum(sizes)[:-1]]))
    assert full.array_data.size == int(sizes.sum())
    assert sum(c for c, l, f in full.array_hist) == len(full.descs)
    assert sum(c for c, l, f in full.index_hist) == int((full.index >= 0).sum())
    oracle.destroy_texture(ob, otx); oracle.destroy_baker(ob)
    product.destroy_texture(b, t); product.destroy_baker(b)


def test_array_data_over_4gib_fails_cleanly(product):
    """bake_cpu_impl.cpp:1774-1775: offsets are 32-bit, a bake whose OMM array would exceed UINT32_MAX bytes is a FAILURE, not a wrap-around.
    300k large triangles at level 8 -> ~4.9 GB of mixed 16 KiB blocks."""
    n = 300000
    tex = ot.foliage_texture(77, 4096, 4096, feature=16)
    uv, ix = ot.random_triangles(78, n, 60.0 / 4096)
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP)
    product.bake(b, d, expect=ot.FAILURE, want_stats=False)
    # the baker is still usable afterwards
    d2 = ot.make_desc(t, uv[:300], ix[:300], 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    r = product.bake(b, d2, want_stats=False)
    assert r.index.size == 100
    product.destroy_texture(b, t)
    product.destroy_baker(b)


def test_user_allocator_is_honoured_and_balanced(product):
    """ommMemoryAllocatorInterface (omm.h:243-256): every host block the library hands out comes from the user's callbacks and is
    returned to them; the warm result pool of the default allocator is bypassed (DESIGN.md fences)."""
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    libc.aligned_alloc.restype = C.c_void_p; libc.aligned_alloc.argtypes = [C.c_size_t, C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    live, stats = {}, dict(allocs=0, frees=0, biggest=0)
    ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t)
    REALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)

    def do_alloc(user, size, align):
        a = max(int(align), 16)
        p = libc.aligned_alloc(a, (int(size) + a - 1) // a * a)
        live[p] = size; stats["allocs"] += 1; stats["biggest"] = max(stats["biggest"], int(size))
        return p

    def do_free(user, p):
        if p:
            assert p in live, "free of a block this allocator never returned"
            del live[p]; stats["frees"] += 1; libc.free(p)

    def do_realloc(user, p, size, align):
        q = do_alloc(user, size, align)
        if p:
            C.memmove(q, p, min(int(size), int(live[p]))); do_free(user, p)
        return q

    cbs = (ALLOC(do_alloc), REALLOC(do_realloc), FREE(do_free))
    d = ot.BakerCreationDesc(); d.type = 1
    d.memoryAllocatorInterface.allocate = C.cast(cbs[0], C.c_void_p)
    d.memoryAllocatorInterface.reallocate = C.cast(cbs[1], C.c_void_p)
    d.memoryAllocatorInterface.free = C.cast(cbs[2], C.c_void_p)
    b = C.c_void_p()
    assert product.fn("ommCreateBaker")(C.byref(d), C.byref(b)) == ot.SUCCESS
    n = 4000
    tex = ot.foliage_texture(8, 1024, 1024, feature=24)
    uv, ix = ot.random_triangles(9, n, 10.0 / 1024)
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    desc = ot.make_desc(t, uv, ix, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    r1 = product.bake(b, desc)
    assert r1.array_data.size > (8 << 20) and stats["biggest"] >= r1.array_data.size     # the big block came through the callbacks
    r2 = product.bake(b, desc)
    assert r1.same_as(r2)
    # a streamed result with the user's (pageable) memory: the array is requested up front with the upper bound of the result -- the packed states of
    # every non-uniform item -- and the blocks arrive through hipMemcpyAsync instead of the SDMA path of the pinned default block
    import bench
    product.set_knob(b, ot.KNOB_STREAM_CHUNKS, 3)
    before = stats["biggest"]
    r3 = product.bake(b, desc)
    tm = bench.BakeTimings()
    tm = bench.get_timings(product, b)
    assert r3.same_as(r1) and tm.streamChunks == 3 and tm.streamedBytes == r1.array_data.size
    assert stats["biggest"] >= before and stats["biggest"] >= tm.stateBytes >= r1.array_data.size
    product.destroy_texture(b, t)
    assert product.destroy_baker(b) == ot.SUCCESS
    assert not live and stats["allocs"] == stats["frees"] and stats["allocs"] >= 6, stats
    # and the default-allocator path (warm pool, reused block) yields the same bytes
    b2 = product.create_baker()
    t2 = product.create_texture(b2, [tex], alpha_cutoff=0.5)
    desc2 = ot.make_desc(t2, uv, ix, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    for _ in range(3):
        assert product.bake(b2, desc2).same_as(r1)
    product.destroy_texture(b2, t2); product.destroy_baker(b2)


def test_bench_contract_small():
    """bench.py prints exactly one JSON line with the driver's keys (tiny workload)"""
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c2", "--tris", "3000", "--steps", "2", "--warmup", "1",
                          "--cpu-sample", "300", "--sat-off-sample", "400", "--host-api-steps", "2", "--stream-chunks", "3", "--create-texture", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    # the headline is the SDK entry point itself; the device-resident entry is the named secondary
    assert d["bake_wall_time_entry"].startswith("ommCpuBake") and d["bake_wall_time_ms"] == d["ms_per_step"] > 0 and d["host_api"]["stream"]["ranges"] == 3 and d["value_entry"] == "ommCpuBake"
    assert d["device_resident"]["entry"] == "ommxBakeDevice" and 0 < d["device_resident"]["ms_per_bake"] and d["device_resident"]["micro_triangles_per_s"] > 0
    fpo = d["fine_pass_only"]                       # the same numerator on both sides: the sample's micro-triangles that enter ResampleFine, counted by the oracle
    assert fpo["cpu_micro_triangles_per_s"] > 0 and fpo["gpu_micro_triangles_per_s"] > 0
    cb = d["cpu_baseline"]   # value / cores: the run with one thread per usable CPU; the oversubscribed runs are in the sweep
    assert cb["cores"] == cb["effective_cpus"] >= 1 and str(cb["cores"]) in cb["threads_sweep"] and str(cb["best_threads"]) in cb["threads_sweep"]
    assert cb["value"] == cb["threads_sweep"][str(cb["cores"])]["micro_triangles_per_s"] > 0
    assert d["roofline"]["bound"] == "hbm"          # (the issue-slot roofline needs the PMC summary of the full-size workload)
    for cfgname in ("c1", "c4", "cards"):
        o2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", cfgname, "--tris", "1500", "--steps", "1", "--warmup", "1", "--cpu-sample", "200",
                             "--host-api-steps", "1"], capture_output=True, text=True, timeout=600)
        assert o2.returncode == 0, (cfgname, o2.stderr[-2000:])
        d2 = json.loads([l for l in o2.stdout.splitlines() if l.startswith("{")][0])
        assert d2["config"]["name"] == cfgname and d2["parity_vs_cpu_baseline"].startswith("bit-exact") and d2["value"] > 0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and d["roofline"]["traffic"] is None   # PMC traffic applies to the default workload only
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["parity_vs_cpu_baseline"].startswith("bit-exact") and d["sat_off"]["parity"].startswith("bit-exact")


# ---------------------------------------------------------------------------------------------
# randomized differential test: every knob of the bake desc at once
# ---------------------------------------------------------------------------------------------
def _fuzz_case(seed):
    h = lambda k: int(ot.hash_u32(np.array([seed * 131 + k], dtype=np.int64))[0])
    pick = lambda k, options: options[h(k) % len(options)]
    w, hgt = pick(1, [(256, 256), (512, 128), (300, 200), (64, 96), (1024, 1024)])
    fp32 = pick(2, [False, True, False])
    kind = pick(3, ["noise", "foliage", "checker"])
    if kind == "noise":
        tex = ot.value_noise(seed, w, hgt, octaves=pick(4, [2, 4, 5]), base_cell=pick(5, [8, 32, 64]))
    elif kind == "foliage":
        tex = ot.foliage_texture(seed, w, hgt, feature=pick(4, [6, 16, 48])).astype(np.float32) / 255.0
    else:
        yy, xx = np.mgrid[0:hgt, 0:w]
        c = pick(4, [1, 3, 16])
        tex = (((xx // c) + (yy // c)) & 1).astype(np.float32)
    tex = np.ascontiguousarray(tex.astype(np.float32) if fp32 else (tex * 255).astype(np.uint8))
    mips = [tex]
    if pick(6, [False, False, True]) and min(w, hgt) >= 64:
        mips.append(np.ascontiguousarray(tex[::2, ::2]))
    level = pick(7, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
    n = max(4, min(400, 3000000 // (4 ** level)))
    ext = pick(8, [0.5, 2.0, 8.0, 40.0, 0.05]) / max(w, hgt)
    uv, ix = ot.random_triangles(seed + 7, n, ext * pick(9, [1.0, 1.0, 8.0]))
    uv = (uv + np.float32(pick(10, [0.0, 0.0, -3.0, 17.0]))).astype(np.float32)
    if pick(11, [False, False, True]):   # shared vertices / repeated triangles (UV dedup, digest dedup)
        ix = (ix // 6 * 3 + ix % 3).astype(ix.dtype)
    kw = dict(addr=pick(12, [ot.WRAP, ot.MIRROR, ot.CLAMP, ot.BORDER, ot.MIRROR_ONCE]), filt=pick(13, [ot.LINEAR, ot.LINEAR, ot.NEAREST]),
              fmt=pick(14, [ot.FMT_4STATE, ot.FMT_2STATE]), promo=pick(15, [ot.PROMO_NEAREST, ot.PROMO_FORCE_OPAQUE, ot.PROMO_FORCE_TRANSPARENT]),
              border_alpha=pick(16, [0.0, 1.0, 0.4]), rejection=pick(17, [0.0, 0.0, 0.3]), dyn_scale=pick(18, [0.0, 0.0, 1.5]))
    le, gt = pick(19, [(ot.T, ot.O), (ot.O, ot.T), (ot.UT, ot.UO), (ot.T, ot.UO)])
    if kw["fmt"] == ot.FMT_2STATE:
        le, gt = (le & 1, gt & 1) if (le & 1) != (gt & 1) else (ot.T, ot.O)
    kw["le"], kw["gt"] = le, gt
    flags = ot.FLAG_THREADS
    for k, f in ((20, ot.FLAG_NO_SPECIAL), (21, ot.FLAG_NO_DEDUP), (22, ot.FLAG_FORCE32), (23, ot.FLAG_ALLOW8)):
        if pick(k, [False, False, True]):
            flags |= f
    kw["flags"] = flags
    if pick(24, [False, True]):
        lv = (ot.hash_u32(np.arange(n) + seed) % (level + 1)).astype(np.uint8)
        lv[ot.hash_u32(np.arange(n) + seed + 99) % 7 == 0] = 0xF
        kw["levels"] = lv
    cutoff = pick(25, [0.5, 0.5, 0.25, 0.8])
    sat = pick(26, [True, True, False])
    return mips, uv, ix, level, cutoff, sat, kw


@pytest.mark.parametrize("seed", range(160))
def test_fuzz_all_knobs(product, oracle, seed):
    mips, uv, ix, level, cutoff, sat, kw = _fuzz_case(seed)
    both(product, oracle, mips, uv, ix, level, sat=sat, cutoff=cutoff, **kw)


def _walk_case(seed):
    """a fuzz case of the deferred generic pass: triangles of 4 .. 300 texels at levels 5 .. 8 over _fuzz_case's textures and knobs (also tests/scripts/bigfuzz.py walks)"""
    mips, _, _, _, cutoff, sat, kw = _fuzz_case(seed)
    h = lambda k: int(ot.hash_u32(np.array([seed * 977 + k], dtype=np.int64))[0])
    level = 5 + h(1) % 4
    texels = [4.0, 20.0, 80.0, 300.0][h(2) % 4]
    n = [120, 120, 60, 24][h(2) % 4]
    uv, ix = ot.random_triangles(seed + 3, n, texels / max(mips[0].shape))
    uv = (uv + np.float32([0.0, 0.0, -3.0, 17.0][h(3) % 4])).astype(np.float32)
    kw = dict(kw); kw.pop("levels", None); kw["dyn_scale"] = 0.0
    if h(4) % 2:
        kw["levels"] = (5 + ot.hash_u32(np.arange(n) + seed) % (level - 4)).astype(np.uint8)
    return mips, uv, ix, level, cutoff, sat, kw


@pytest.mark.parametrize("seed", list(range(400, 496)))
def test_fuzz_deferred_texel_walks(product, oracle, seed):
    """The deferred generic pass (bake_kernels.hip: classify_generic / generic_dense -- owners, visit rings, votes summed in LDS) against the oracle over the fuzz
    generator's textures, address modes, filters, promotions, formats, flags and per-triangle levels, with triangles of 4 .. 300 texels at levels 5 .. 8:
    micro-triangles from a fraction of a texel to a dozen texels across, i.e. walks of every length in one wave."""
    mips, uv, ix, level, cutoff, sat, kw = _walk_case(seed)
    both(product, oracle, mips, uv, ix, level, sat=sat, cutoff=cutoff, knobs=[(ot.KNOB_GENERIC_PASS, 2)], **kw)


def test_concurrent_bakes_on_one_baker(product):
    """ommCpuBake is re-entrant (the reference bakes are independent objects, bake.cpp:103-116): four host threads share one baker
    and one texture; a call that finds the baker's device arena busy works in a private one."""
    import threading
    tex = ot.foliage_texture(21, 1024, 1024, feature=20)
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    jobs = []
    for k in range(4):
        uv, ix = ot.random_triangles(300 + k, 1500 + 200 * k, 9.0 / 1024)
        jobs.append((uv, ix, ot.make_desc(t, uv, ix, 6 + (k & 1), addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)))
    expected = [product.bake(b, d, want_stats=False) for (_, _, d) in jobs]
    errors = []

    def worker(k):
        try:
            for _ in range(4):
                r = product.bake(b, jobs[k][2], want_stats=False)
                if not r.same_as(expected[k]):
                    errors.append("thread %d: %s" % (k, r.diff(expected[k])))
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    product.destroy_texture(b, t)
    product.destroy_baker(b)


def test_work_item_ids_collide_like_the_reference(product, oracle):
    """SetupWorkItems keys its triangle -> work item map by a 64-bit hash chain and trusts it (libraries/omm-lib/src/bake_cpu_impl.cpp:626-649): triangles
    whose ids collide are ONE work item, whatever their coordinates.  tests/golden/vmid_collisions.json holds pairs of different UV points with the same
    std::hash<glm::vec2>; triangles that differ only in such a point must share the first one's micro-map -- in the oracle (which restates the hash chain)
    and in the HIP library (omm_amd/csrc/vm_id.h), through ommCpuBake and through the device-resident entry.  Real duplicates and per-triangle levels in
    the mix; with a different level the ids differ again."""
    import json, os
    pairs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vmid_collisions.json")))["pairs"]
    f = lambda h: np.array([int(h, 16)], np.uint32).view(np.float32)[0]
    uv, ix = ot.random_triangles(77, 300, 0.03)
    lv = (3 + ot.hash_u32(np.arange(300) + 5) % 4).astype(np.uint8)
    k = 10
    for pr in pairs:                       # triangle k ends in point a, triangle k + 100 is the same triangle ending in point b; k + 200: b again at another level
        for j, pt in ((k, pr["a"]), (k + 100, pr["b"]), (k + 200, pr["b"])):
            uv[3 * j:3 * j + 2] = uv[3 * k:3 * k + 2]
            uv[3 * j + 2] = (f(pt[0]), f(pt[1]))
        lv[k + 100] = lv[k]; lv[k + 200] = lv[k] + 1
        k += 7
    uv[3 * 50:3 * 53] = uv[3 * 40:3 * 43]; lv[50:53] = lv[40:43]                # real duplicates
    tex = ot.foliage_texture(8, 512, 512, feature=24)
    got = both(product, oracle, [tex], uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv)
    res = got[0] if isinstance(got, (tuple, list)) else got
    if res is not None and hasattr(res, "index"):
        k = 10
        for pr in pairs:
            assert res.index[k] == res.index[k + 100], "colliding ids must share a micro-map"
            k += 7
    both(product, oracle, [tex], uv, ix, 5, addr=ot.CLAMP, promo=ot.PROMO_NEAREST, fmt=ot.FMT_2STATE)
    hip = ot.Hip()
    ob = oracle.create_baker(); otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
    ref = oracle.bake(ob, ot.make_desc(otx, uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv))
    oracle.destroy_texture(ob, otx); oracle.destroy_baker(ob)
    assert all(ref.index[10 + 7 * i] == ref.index[110 + 7 * i] for i in range(len(pairs)))
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    dev = ot.bake_device(product, hip, b, ot.make_desc(t, uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv), uv, ix, levels=lv)
    product.destroy_texture(b, t); product.destroy_baker(b)
    dev.stats2 = None
    assert dev.same_as(ref), dev.diff(ref)


def test_no_memory_growth_over_baker_lifecycles(product):
    """30 complete lifecycles (baker, texture, host-array bake, device-resident bake, sharded bake with one rank, blob round trip, destroy
    everything): the free HBM reported by hipMemGetInfo and the resident set of the process return to where they were after the first
    (warm-up) cycle -- the baker's pools, arenas, streams and events are released with it"""
    import ctypes as C, resource
    hip = ot.Hip()
    hip.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    hip.rt.hipDeviceSynchronize.argtypes = []
    import omm_amd.sharded as sh

    def free_hbm():
        assert hip.rt.hipDeviceSynchronize() == 0
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.rt.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    def rss():
        with open("/proc/self/statm") as fh:
            return int(fh.read().split()[1]) * resource.getpagesize()

    tex = ot.foliage_texture(33, 1024, 1024, feature=24)
    uv, ix = ot.random_triangles(4242, 3000, 10.0 / 1024)
    d_uv, d_ix = hip.upload(uv), hip.upload(ix.astype(np.int32))

    def cycle():
        b = product.create_baker()
        t = product.create_texture(b, [tex], alpha_cutoff=0.5)
        d = ot.make_desc(t, uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
        r = product.bake(b, d, want_stats=False)
        dd = ot.BakeInputDesc.from_buffer_copy(d); dd.texCoords, dd.indexBuffer = d_uv.value, d_ix.value
        out = C.c_void_p()
        product.dll.ommxBakeDevice.argtypes = [C.c_void_p, C.POINTER(ot.BakeInputDesc), C.POINTER(C.c_void_p)]
        assert product.dll.ommxBakeDevice(b, C.byref(dd), C.byref(out)) == ot.SUCCESS
        product.dll.ommxDestroyDeviceBakeResult.argtypes = [C.c_void_p]
        assert product.dll.ommxDestroyDeviceBakeResult(out) == ot.SUCCESS
        out2 = sh.sharded_bake(product.dll, b, C.byref(dd), 0, 1, None, None)
        assert product.dll.ommxDestroyDeviceBakeResult(out2) == ot.SUCCESS
        product.destroy_texture(b, t)
        product.destroy_baker(b)
        return r

    first = cycle()
    cycle()
    f0, r0 = free_hbm(), rss()
    for _ in range(30):
        assert cycle().same_as(first)
    f1, r1 = free_hbm(), rss()
    hip.rt.hipFree(d_uv); hip.rt.hipFree(d_ix)
    assert f0 - f1 < (32 << 20), "free HBM shrank by %d bytes over 30 lifecycles" % (f0 - f1)
    assert r1 - r0 < (96 << 20), "resident set grew by %d bytes over 30 lifecycles" % (r1 - r0)


def test_log_cases(product):
    """support/tests/test_omm_log.cpp:146-209 through the HIP library: same messages, same order, same results"""
    import log_cases
    log_cases.run_log_cases(product)


def test_basic_cases(product):
    """support/tests/test_basic.cpp:28-51,217-279 through the HIP library"""
    import basic_cases
    basic_cases.run_basic_cases(product)


def test_minimal_sample(product, oracle):
    """support/tests/test_minimal_sample.cpp:17-150: triangle fan over a donut, per-triangle levels 2..5, 2-state, shared vertices"""
    j, i = np.mgrid[0:256, 0:256]
    dx = (i.astype(np.float32) / np.float32(256)) - np.float32(0.5)
    dy = (j.astype(np.float32) / np.float32(256)) - np.float32(0.5)
    ln = np.sqrt((dx * dx + dy * dy).astype(np.float32)).astype(np.float32)
    tex = np.where((ln > np.float32(0.2)) & (ln < np.float32(0.3)), 1.0, 0.0).astype(np.float32)
    uv = np.array([[0.05, 0.50], [0.50, 0.05], [0.50, 0.50], [0.95, 0.50], [0.50, 0.95]], np.float32)
    ix = np.array([0, 1, 2, 1, 3, 2, 3, 4, 2, 2, 4, 0], np.uint32)
    r = both(product, oracle, [tex], uv, ix, 8, sat=False, cutoff=0.5, addr=ot.CLAMP, filt=ot.LINEAR, fmt=ot.FMT_2STATE,
             promo=ot.PROMO_FORCE_OPAQUE, flags=0, levels=np.array([2, 3, 4, 5], np.uint8))   # (the sample adds EnableValidation + a log callback)
    assert sorted(int(l) for l in r.descs[:, 1]) == [2, 3, 4, 5] and np.all(r.descs[:, 2] == 1)


def _texture_pitch_and_desc(lib):
    """rowPitch is in BYTES for DisableZOrder textures and in TEXELS otherwise (texture_impl.cpp:141-142,169,179); ommCpuGetTextureDesc
    returns the tight texels with rowPitch = width (texture_impl.cpp:280-325).  Returns the bake results for comparison."""
    import ctypes as C
    w, h, pad = 96, 40, 7
    out = []
    for dtype in (np.uint8, np.float32):
        tight = ot.value_noise(31, w, h, octaves=3, base_cell=16)
        tight = np.ascontiguousarray((tight * 255).astype(np.uint8) if dtype == np.uint8 else tight.astype(np.float32))
        padded = np.zeros((h, w + pad), dtype)
        padded[:, :w] = tight
        uv, ix = ot.random_triangles(32, 60, 0.3)
        for linear in (True, False):
            b = lib.create_baker()
            pitch = (w + pad) * padded.itemsize if linear else (w + pad)
            md = (ot.TextureMipDesc * 1)()
            md[0].width, md[0].height, md[0].rowPitch, md[0].textureData = w, h, pitch, padded.ctypes.data
            td = ot.TextureDesc()
            td.format, td.flags, td.mips, td.mipCount, td.alphaCutoff = (ot.TEX_FP32 if dtype == np.float32 else ot.TEX_UNORM8), (ot.TEXFLAG_DISABLE_ZORDER if linear else 0), md, 1, 0.5
            t = C.c_void_p()
            assert lib.fn("ommCpuCreateTexture")(b, C.byref(td), C.byref(t)) == ot.SUCCESS
            r_pitched = lib.bake(b, ot.make_desc(t, uv, ix, 5, addr=ot.WRAP))
            if hasattr(lib.dll, lib.prefix + "ommCpuGetTextureDesc"):   # (the oracle restates the bake only)
                q = ot.TextureDesc()
                assert lib.fn("ommCpuGetTextureDesc")(t, C.byref(q)) == ot.SUCCESS          # mips == NULL: header only
                assert (q.format, q.flags, q.mipCount) == (td.format, td.flags, 1) and abs(q.alphaCutoff - 0.5) < 1e-7
                back = np.full((h, w), 77, dtype)
                md2 = (ot.TextureMipDesc * 1)(); md2[0].textureData = back.ctypes.data
                q.mips = md2
                assert lib.fn("ommCpuGetTextureDesc")(t, C.byref(q)) == ot.SUCCESS
                assert (md2[0].width, md2[0].height, md2[0].rowPitch) == (w, h, w)
                assert np.array_equal(back, tight)
            lib.destroy_texture(b, t)
            t2 = lib.create_texture(b, [tight], alpha_cutoff=0.5, disable_zorder=linear)
            r_tight = lib.bake(b, ot.make_desc(t2, uv, ix, 5, addr=ot.WRAP))
            assert r_pitched.same_as(r_tight), r_pitched.diff(r_tight)
            lib.destroy_texture(b, t2); lib.destroy_baker(b)
            out.append(r_tight)
    return out


def test_texture_row_pitch_and_get_desc(product, oracle):
    a, b = _texture_pitch_and_desc(product), _texture_pitch_and_desc(oracle)
    for x, y in zip(a, b):
        assert x.same_as(y), x.diff(y)


def test_two_process_sharded_bake_on_one_gpu():
    """two real processes + torch.distributed collectives on device tensors (gloo: RCCL refuses two ranks on one GPU): every rank ends with
    the single-GPU result (tests/scripts/two_rank_gloo_gpu.py)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "scripts", "two_rank_gloo_gpu.py"), "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "two-process sharded bake ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


@pytest.mark.parametrize("collectives,launcher", [("native", "torchrun"), ("torch", "torchrun"), ("native", "plain")])
def test_bench_two_ranks_on_one_gpu(collectives, launcher):
    """bench.py's N > 1 path end to end (barrier + max-over-ranks timing, one JSON line from rank 0), with both ranks on GPU 0 over gloo
    (self-test hooks of bench.py): the one-call entry ommxShardedBakeRccl over the process group's collectives, and the caller-driven
    four-call path -- launched through torch.distributed.run as the driver does, and PLAINLY as `python bench.py --gpus 2`, which
    re-executes itself under the launcher"""
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMM_BENCH_ONE_GPU="1", OMM_BENCH_BACKEND="gloo", OMM_BENCH_COLLECTIVES=collectives)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    wrap = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517"] if launcher == "torchrun" else []
    cmd = [sys.executable] + wrap + [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tris", "20000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["sharding"] != "none"
    assert d["value_entry"] == ("ommxShardedBakeRccl" if collectives == "native" else "ommxSharded* + torch.distributed")
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only


def test_c_example_program(product, tmp_path):
    """examples/minimal_sample.c (plain C against include/omm_mi355x.h) prints the same micro-maps as the same bake through ctypes"""
    import subprocess, re
    from test_abi_exports import _build_example
    exe = str(tmp_path / "minimal_sample")
    _build_example(["gcc", "-std=c99", "-pedantic"], exe)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [tuple(int(x) for x in m.groups()) for m in re.finditer(r"omm \d+: level (\d+) offset (\d+) opaque (\d+) / (\d+)", out.stdout)]
    j, i = np.mgrid[0:256, 0:256]
    dx = (i.astype(np.float32) / np.float32(256)) - np.float32(0.5)
    dy = (j.astype(np.float32) / np.float32(256)) - np.float32(0.5)
    ln = np.sqrt((dx * dx + dy * dy).astype(np.float32)).astype(np.float32)
    tex = np.where((ln > np.float32(0.2)) & (ln < np.float32(0.3)), 1.0, 0.0).astype(np.float32)
    uv = np.array([[0.05, 0.50], [0.50, 0.05], [0.50, 0.50], [0.95, 0.50], [0.50, 0.95]], np.float32)
    ix = np.array([0, 1, 2, 1, 3, 2, 3, 4, 2, 2, 4, 0], np.uint32)
    b = product.create_baker(callback=lambda s, m, u: None)
    t = product.create_texture(b, [tex], alpha_cutoff=-1.0)
    d = ot.make_desc(t, uv, ix, 8, addr=ot.CLAMP, filt=ot.LINEAR, fmt=ot.FMT_2STATE, promo=ot.PROMO_FORCE_OPAQUE, flags=ot.FLAG_VALIDATION,
                     levels=np.array([2, 3, 4, 5], np.uint8), dyn_scale=2.0)
    r = product.bake(b, d)
    want = []
    for off, lvl, fmt in r.descs:
        n = 4 ** int(lvl)
        bits = np.unpackbits(r.array_data[int(off):int(off) + max(1, n // 8)], bitorder="little")[:n]
        want.append((int(lvl), int(off), int(bits.sum()), n))
    assert got == want and len(got) == 4, (got, want)
    product.destroy_texture(b, t); product.destroy_baker(b)


def test_internal_flags_no_fine_pass_and_edge_heuristic(product, oracle):
    """Internal bake flags (bake_cpu_impl.cpp:43-48).  Bit 9 (DisableFineClassification) skips ResampleFine: what the summed-area pass leaves
    unresolved keeps the initial UnknownOpaque -- the reference's own "everything but the fine pass" timing switch; bit 11 (EnableEdgeHeuristic)
    selects the edge-length level heuristic for every triangle (glibc log2f on the host).  Bits 7 / 8 (the alternative conservative-bilinear
    kernel): test_internal_flags_conservative_bilinear_kernel."""
    tex = ot.foliage_texture(3, 512, 512, feature=24)
    uv, ix = ot.random_triangles(31, 600, 0.05)
    lv = (2 + ot.hash_u32(np.arange(600) + 3) % 7).astype(np.uint8)          # levels 2..8: both tile sizes and the small-item launches
    for sat in (True, False):
        for filt in (ot.LINEAR, ot.NEAREST):
            both(product, oracle, [tex], uv, ix, 8, sat=sat, addr=ot.WRAP, filt=filt, promo=ot.PROMO_FORCE_OPAQUE, levels=lv, flags=ot.FLAG_THREADS | (1 << 9))
    both(product, oracle, [noise_u8()], uv, ix, 7, addr=ot.CLAMP, promo=ot.PROMO_NEAREST, flags=ot.FLAG_THREADS | (1 << 9))
    # edge heuristic: dynamic levels for every triangle, with and without per-triangle overrides
    lv2 = np.where(ot.hash_u32(np.arange(600) + 9) % 3 == 0, 0xF, lv).astype(np.uint8)
    both(product, oracle, [tex], uv, ix, 9, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, dyn_scale=2.0, flags=ot.FLAG_THREADS | (1 << 11))
    both(product, oracle, [tex], uv, ix, 9, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, dyn_scale=0.7, levels=lv2, flags=ot.FLAG_THREADS | (1 << 11) | (1 << 9))
    # bit 9 with the 2-state format: unresolved micro-triangles keep UnknownOpaque (3), which the reference digests unpacked and ORs into the 1-bit packing as
    # `3 << (i & 7)` (bake_cpu_impl.cpp:1811) -- every level (blocks of less than a byte included), special indices on and off, dedup on and off
    lv3 = (ot.hash_u32(np.arange(600) + 5) % 9).astype(np.uint8)             # levels 0..8
    for sat in (True, False):
        for extra in (0, ot.FLAG_NO_SPECIAL, ot.FLAG_NO_DEDUP):
            both(product, oracle, [tex], uv, ix, 8, sat=sat, addr=ot.WRAP, fmt=ot.FMT_2STATE, promo=ot.PROMO_FORCE_OPAQUE, levels=lv3, flags=ot.FLAG_THREADS | (1 << 9) | extra)
    both(product, oracle, [noise_u8()], uv, ix, 6, addr=ot.CLAMP, fmt=ot.FMT_2STATE, promo=ot.PROMO_NEAREST, rejection=0.4, flags=ot.FLAG_THREADS | (1 << 9))


def test_internal_flags_conservative_bilinear_kernel(product, oracle):
    """Internal bake flags bit 8 (DisableLevelLineIntersection: ConservativeBilinearKernel over the micro-triangle's raster, bake_kernels_cpu.h:404-452,
    bake_cpu_impl.cpp:941-966) and bits 7 + 8 (EnableAABBTesting: the same kernel over the two triangles of the micro-triangle's bounding box, :915-940): no
    centre vote, mip 0 only; the summed-area pass in front of it as usual.  Bit 7 alone is an invalid argument (:718-719)."""
    tex = ot.foliage_texture(3, 512, 512, feature=24)
    texf = ot.value_noise(8, 300, 200, octaves=3, base_cell=16).astype(np.float32)
    uv, ix = ot.random_triangles(41, 500, 0.05)
    uv[3 * 40:3 * 44, :] = uv[3 * 40:3 * 44, :1] * np.float32(0.5) + np.float32(0.25)          # degenerate triangles (all three vertices on a line x == y)
    lv = (ot.hash_u32(np.arange(500) + 13) % 9).astype(np.uint8)                               # levels 0..8
    for bits in ((1 << 8), (1 << 7) | (1 << 8)):
        for sat in (True, False):
            both(product, oracle, [tex], uv, ix, 8, sat=sat, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv, flags=ot.FLAG_THREADS | bits)
        both(product, oracle, [tex], uv - np.float32(0.3), ix, 6, addr=ot.MIRROR, promo=ot.PROMO_NEAREST, fmt=ot.FMT_2STATE, flags=ot.FLAG_THREADS | bits)   # negative pixels: int(x + 0.5) truncates towards zero
        both(product, oracle, [texf], uv * np.float32(1.7) - np.float32(0.2), ix, 5, sat=False, addr=ot.BORDER, border_alpha=0.7, promo=ot.PROMO_FORCE_TRANSPARENT, flags=ot.FLAG_THREADS | bits)
        both(product, oracle, [texf], uv, ix, 7, addr=ot.CLAMP, promo=ot.PROMO_NEAREST, levels=lv, flags=ot.FLAG_THREADS | bits, knobs=[(ot.KNOB_GENERIC_PASS, 2)])
        both(product, oracle, [tex], uv, ix, 6, addr=ot.WRAP, filt=ot.NEAREST, promo=ot.PROMO_FORCE_OPAQUE, flags=ot.FLAG_THREADS | bits)      # (the Nearest filter ignores both bits)
        both(product, oracle, [tex], uv, ix, 7, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=ot.FLAG_THREADS | bits | (1 << 9))            # (bit 9 wins: no fine pass at all)
    both(product, oracle, [tex], uv, ix, 5, addr=ot.WRAP, flags=ot.FLAG_THREADS | (1 << 7), expect=ot.INVALID_ARGUMENT)


def test_baker_memory_retention_knob_and_trim(product, oracle):
    """ommxBakerKnob_RetainMemory = 1: the baker keeps no idle working set, device result block or pinned host block between bakes; ommxTrimBaker gives the
    idle ones back at once.  Results stay valid (they own their blocks) and later bakes allocate again: same bytes as the oracle throughout."""
    import ctypes
    tex = ot.foliage_texture(9, 512, 512, feature=24)
    uv, ix = ot.random_triangles(77, 1500, 0.03)
    product.dll.ommxTrimBaker.argtypes = [ctypes.c_void_p]
    ref = both(product, oracle, [tex], uv, ix, 7, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    b = product.create_baker()
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, 7, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    assert product.bake(b, d).same_as(ref)
    assert product.dll.ommxTrimBaker(b) == ot.SUCCESS
    assert product.bake(b, d).same_as(ref)
    product.set_knob(b, ot.KNOB_RETAIN_MEMORY, 1)
    for _ in range(3):
        assert product.bake(b, d).same_as(ref)
    product.set_knob(b, ot.KNOB_STREAM_CHUNKS, 3)          # (a streamed result takes its pinned block from the pool and hands it straight back)
    assert product.bake(b, d).same_as(ref)
    product.set_knob(b, ot.KNOB_RETAIN_MEMORY, 0)
    assert product.bake(b, d).same_as(ref)
    product.destroy_texture(b, t)
    product.destroy_baker(b)


@pytest.mark.parametrize("chunks", [1, 3, 7])
def test_streamed_result_of_ommCpuBake(product, oracle, chunks):
    """ommCpuBake sends finished OMM blocks to the host while the classification is still running: the active items are sorted into the order of
    the final result, the levels >= 6 are classified in `chunks` launches over consecutive ranges of it, the blocks of each range are packed behind
    the earlier ones and copied to their final arrayData offsets on a second stream; the placement is verified against the ordinary tail at the end
    (omm_host.cpp: StreamOut; tail_kernels.hip: "Streamed result").  Large bakes do that on their own; ommxBakerKnob_StreamChunks forces it
    here on small ones: every level (small-item launches, 1024-tiles, 4096-tiles), duplicates, both formats, the rejection threshold -- all
    byte-identical to the oracle."""
    tex = ot.foliage_texture(21, 1024, 1024, feature=48)
    n = 2500
    uv, ix = ot.random_triangles(515, n, 0.04)
    uv[3 * 700:3 * 730] = uv[3 * 20:3 * 50]                                   # exact duplicates (UV dedup)
    lv = (1 + ot.hash_u32(np.arange(n) + 17) % 9).astype(np.uint8); lv[700:730] = lv[20:50]          # levels 1..9
    knobs = [(ot.KNOB_STREAM_CHUNKS, chunks)]
    both(product, oracle, [tex], uv, ix, 9, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv, knobs=knobs)
    both(product, oracle, [tex], uv, ix, 7, addr=ot.CLAMP, promo=ot.PROMO_NEAREST, fmt=ot.FMT_2STATE, knobs=knobs)
    both(product, oracle, [tex], uv, ix, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv, rejection=0.7, knobs=knobs)
    both(product, oracle, [tex], uv, ix, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv, flags=0, knobs=knobs)
    both(product, oracle, [tex], uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL, knobs=knobs)   # (falls back to the plain copy)
    both(product, oracle, [tex], uv, ix, 6, sat=False, addr=ot.MIRROR, promo=ot.PROMO_FORCE_TRANSPARENT, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP, knobs=knobs)
    # nothing to stream: every item uniform
    flat = np.full((64, 64), 255, np.uint8)
    both(product, oracle, [flat], uv, ix, 6, addr=ot.WRAP, knobs=knobs)


def test_streamed_result_repeated_and_concurrent(product, oracle):
    """the working sets (staging buffer included) are reused across bakes, and concurrent bakes on one baker get their own working sets"""
    import threading
    tex = ot.foliage_texture(5, 1024, 1024, feature=32)
    uv, ix = ot.random_triangles(99, 1500, 0.03)
    ob = oracle.create_baker(); otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
    ref = oracle.bake(ob, ot.make_desc(otx, uv, ix, 7, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE))
    oracle.destroy_texture(ob, otx); oracle.destroy_baker(ob)
    b = product.create_baker(); product.set_knob(b, ot.KNOB_STREAM_CHUNKS, 4)
    t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, 7, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    for _ in range(3):
        assert product.bake(b, d).same_as(ref)
    errors = []
    def worker():
        try:
            for _ in range(3):
                r = product.bake(b, d)
                if not r.same_as(ref):
                    errors.append(r.diff(ref))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
    ths = [threading.Thread(target=worker) for _ in range(3)]
    for x in ths: x.start()
    for x in ths: x.join()
    assert not errors, errors
    product.destroy_texture(b, t); product.destroy_baker(b)


def test_streamed_result_falls_back_when_a_later_range_owns_the_block(product, oracle):
    """The streamed placement is speculative in one respect (tail_kernels.hip "Streamed result"): a block is emitted for the lowest work-item index
    with its digest SO FAR.  Copies of one triangle shifted by whole periods of a periodic texture classify identically, so their digests collide
    across ranges in both index orders.  For ordinary triangles the level-5 preview sees that coming (the copies share their preview, are classified
    early, and the bake streams); asset-sized triangles are not previewed, so there a later range brings the lower index, the bake must notice and
    fall back to the ordinary gather + copy -- and the result is the oracle's either way."""
    import bench
    import ctypes
    fell_back = []
    for size, period, extent, lo, hi, n_base, level in ((512, 64, 0.02, 0.03, 0.09, 60, 7), (2048, 64, 0.7, 0.45, 0.55, 5, 7)):
        yy, xx = np.mgrid[0:size, 0:size]
        tile = ((((xx % period) - period // 2) ** 2 + ((yy % period) - period // 2) ** 2) < (period * 5 // 16) ** 2).astype(np.uint8) * 255   # a disc per period
        step = np.float32(period / size)
        base_uv, _ = ot.random_triangles(4, n_base, extent, lo=lo, hi=hi)
        rng_k = ot.hash_u32(np.arange(n_base * 12) + 5)
        shift = np.stack([(rng_k % 8).astype(np.float32) * step, ((rng_k >> 3) % 8).astype(np.float32) * step], 1).reshape(12, n_base, 1, 2)
        uv = (base_uv.reshape(1, n_base, 3, 2) + shift).astype(np.float32).reshape(-1, 2)                 # 12 copies of every triangle, each in a period of its own
        n = uv.shape[0] // 3
        ix = np.arange(3 * n, dtype=np.uint32)
        ob = oracle.create_baker(); otx = oracle.create_texture(ob, [tile], alpha_cutoff=0.5)
        ref = oracle.bake(ob, ot.make_desc(otx, uv, ix, level, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE))
        oracle.destroy_texture(ob, otx); oracle.destroy_baker(ob)
        assert 0 < len(ref.descs) < n // 3        # the workload really is full of duplicate blocks
        b = product.create_baker(); product.set_knob(b, ot.KNOB_STREAM_CHUNKS, 6)
        t = product.create_texture(b, [tile], alpha_cutoff=0.5)
        res = product.bake(b, ot.make_desc(t, uv, ix, level, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE))
        tm = bench.BakeTimings()
        tm = bench.get_timings(product, b)
        product.destroy_texture(b, t); product.destroy_baker(b)
        assert res.same_as(ref), res.diff(ref)
        assert tm.streamedBytes > 0
        fell_back.append(tm.streamChunks == 0)
    assert fell_back == [False, True], fell_back      # previewed copies stream; the asset-sized ones stream, notice, and take the ordinary path


def test_both_formats_and_both_generic_passes_at_scale(product):
    """Size-independent cross checks of this round's machinery at sizes the oracle cannot follow: (a) a 2-state bake large enough to stream its result on its
    own (512-byte digest chunks, one-bit blocks) equals the device-resident bake, which assembles its result after the classification; (b) the asset-shaped
    workload gives the same bytes with the generic texel-loop path inside the persistent kernel and as the deferred pass."""
    import workloads as wl, bench, ctypes
    hip = ot.Hip()
    # (a)
    tex, uv, ix, lv, kw = wl.workload("c2", 200000, fmt=ot.FMT_2STATE)
    kw = dict(kw, fmt=ot.FMT_2STATE); lvl = kw.pop("level")
    b = product.create_baker(); t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, lvl, **kw)
    product.set_knob(b, ot.KNOB_RESULT_TRANSFER, ot.TRANSFER_STREAMED)
    host = product.bake(b, d, want_stats=False)
    tm = bench.get_timings(product, b)
    assert tm.streamChunks > 1 and tm.streamedBytes == host.array_data.size > (64 << 20), (tm.streamChunks, tm.streamedBytes, host.array_data.size)
    dev = ot.bake_device(product, hip, b, d, uv, ix)
    assert dev.same_as(host), dev.diff(host)
    product.set_knob(b, ot.KNOB_RESULT_TRANSFER, ot.TRANSFER_COMPRESSED)   # ... and as a codec stream expanded by the helper threads (one-bit blocks)
    comp = product.bake(b, d, want_stats=False)
    tm = bench.get_timings(product, b)
    assert tm.resultTransfer == ot.TRANSFER_COMPRESSED and 0 < tm.compressedBytes < host.array_data.size // 2, (tm.resultTransfer, tm.compressedBytes)
    assert comp.same_as(host), comp.diff(host)
    product.destroy_texture(b, t); product.destroy_baker(b)
    # (b)
    tex, uv, ix, lv, kw = wl.workload("cards", 8000)
    kw = dict(kw); lvl = kw.pop("level")
    results = []
    for mode in (1, 2):
        b = product.create_baker(); product.set_knob(b, ot.KNOB_GENERIC_PASS, mode)
        t = product.create_texture(b, [tex], alpha_cutoff=0.5)
        results.append(ot.bake_device(product, hip, b, ot.make_desc(t, uv, ix, lvl, levels=lv, **kw), uv, ix, levels=lv))
        tm = bench.get_timings(product, b)
        assert (tm.genericMicroTriangles > 1000000) == (mode == 2), (mode, tm.genericMicroTriangles)
        product.destroy_texture(b, t); product.destroy_baker(b)
    assert results[0].same_as(results[1]), results[0].diff(results[1])


@pytest.mark.parametrize("generic_pass", [1, 2])
def test_micro_triangles_of_several_texels(product, oracle, generic_pass):
    """Micro-triangles that span several texels (asset-sized triangles: the shape of the reference's Leaflet KATs at production size): the generic
    texel-loop path (conservative raster + level-line kernel per texel) with every promotion.  Quads of 40 .. 700 texels at levels 6 .. 9,
    Clamp / Wrap / Mirror / Border, UNORM8 and FP32, SAT on and off, 2-state.  Both homes of that path (ommxBakerKnob_GenericPass): inside the
    persistent classification launch, one lane per micro-triangle, and the deferred pass (bake_kernels.hip: classify_generic), whose lanes pull walks from a queue."""
    import workloads as wl
    knobs = [(ot.KNOB_GENERIC_PASS, generic_pass)]
    tex8 = ot.foliage_texture(77, 1024, 1024, feature=48)
    texf = ot.value_noise(13, 700, 500, octaves=4, base_cell=40).astype(np.float32)
    uv, ix, lv = wl.card_quads(3, 40, 1024, lo_texels=40.0, hi_texels=700.0)
    lv9 = lv.copy(); lv9[::5] = 9
    for promo in (ot.PROMO_FORCE_OPAQUE, ot.PROMO_FORCE_TRANSPARENT, ot.PROMO_NEAREST):
        both(product, oracle, [tex8], uv, ix, 8, addr=ot.CLAMP, promo=promo, levels=lv, knobs=knobs)
    both(product, oracle, [tex8], uv, ix, 9, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv9, knobs=knobs)
    both(product, oracle, [tex8], uv, ix, 8, sat=False, addr=ot.MIRROR, promo=ot.PROMO_FORCE_OPAQUE, levels=lv, knobs=knobs)
    both(product, oracle, [texf], uv, ix, 7, addr=ot.CLAMP, promo=ot.PROMO_FORCE_TRANSPARENT, fmt=ot.FMT_2STATE, knobs=knobs)
    both(product, oracle, [texf], uv * np.float32(1.7) - np.float32(0.3), ix, 6, addr=ot.BORDER, promo=ot.PROMO_FORCE_OPAQUE, border_alpha=0.7, knobs=knobs)
    # random triangles of 20 .. 60 texels (no axis-aligned edges), level 6
    uv2, ix2 = ot.random_triangles(61, 300, 0.05)
    both(product, oracle, [tex8], uv2, ix2, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, knobs=knobs)
    # Nearest filter (every covered texel votes with its sample; the counts decide under the Nearest promotion), a mip chain (the walk of a micro-triangle
    # ends at the first mip that leaves it unknown), degenerate triangles among the cards (line walks)
    both(product, oracle, [tex8], uv, ix, 7, filt=ot.NEAREST, addr=ot.WRAP, promo=ot.PROMO_NEAREST, levels=lv, knobs=knobs)
    both(product, oracle, [tex8], uv, ix, 7, sat=False, filt=ot.NEAREST, addr=ot.CLAMP, promo=ot.PROMO_FORCE_TRANSPARENT, knobs=knobs)
    mips = [texf]
    while min(mips[-1].shape) > 40:
        t = mips[-1][:mips[-1].shape[0] // 2 * 2, :mips[-1].shape[1] // 2 * 2]
        mips.append(((t[0::2, 0::2] + t[1::2, 0::2]) + t[0::2, 1::2] + t[1::2, 1::2]) * np.float32(0.25))
    both(product, oracle, mips[:3], uv, ix, 6, sat=False, addr=ot.WRAP, promo=ot.PROMO_NEAREST, knobs=knobs)
    both(product, oracle, mips[:3], uv, ix, 7, sat=False, addr=ot.CLAMP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv, knobs=knobs)
    uvd = uv.copy().reshape(-1, 2); uvd[ix[3::18]] = uvd[ix[4::18]]          # every sixth triangle collapses onto an edge
    both(product, oracle, [tex8], uvd, ix, 7, addr=ot.CLAMP, promo=ot.PROMO_FORCE_OPAQUE, knobs=knobs)


def test_texel_walk_shapes(product, oracle):
    """The walks of the deferred pass leave a row at the first texel that is not under the triangle after one that was, and pass cells that cannot change
    the state: shapes that stress both -- thin slivers in every direction (rows with a single covered texel, rows with none), triangles whose boxes are
    hundreds of texels wide and a few high, UV offsets far from the origin and below zero (Wrap / Mirror), a non-power-of-two FP32 texture, 0 / 1 noise
    (no flat cells next to each other), a constant texture (only flat cells) -- against the oracle, every promotion."""
    knobs = [(ot.KNOB_GENERIC_PASS, 2)]
    rng = np.random.RandomState(5)
    n = 60
    c = rng.rand(n, 2).astype(np.float32) * np.float32(0.8) + np.float32(0.1)
    ang = rng.rand(n).astype(np.float32) * np.float32(2 * np.pi)
    ln = (np.float32(0.05) + rng.rand(n).astype(np.float32) * np.float32(0.25)); th = ln * np.float32(0.002) * (1 + (np.arange(n) % 7)).astype(np.float32)
    d = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32); o = np.stack([-d[:, 1], d[:, 0]], 1)
    tri = np.stack([c - d * ln[:, None], c + d * ln[:, None], c + o * th[:, None]], 1).astype(np.float32)
    tri[::9, 1, 1] = tri[::9, 0, 1]                                   # exactly horizontal long edge
    tri[1::9, 1, 0] = tri[1::9, 0, 0]                                 # exactly vertical long edge
    uv = np.ascontiguousarray(tri.reshape(-1, 2)); ix = np.arange(3 * n, dtype=np.uint32)
    tex8 = ot.foliage_texture(21, 512, 512, feature=24)
    texf = ot.value_noise(3, 300, 200, octaves=3, base_cell=20).astype(np.float32)
    noise = (rng.rand(256, 256) > 0.5).astype(np.float32)
    const = np.full((128, 128), 0.75, np.float32)
    for promo in (ot.PROMO_FORCE_OPAQUE, ot.PROMO_FORCE_TRANSPARENT, ot.PROMO_NEAREST):
        both(product, oracle, [tex8], uv, ix, 6, addr=ot.CLAMP, promo=promo, knobs=knobs)
    both(product, oracle, [tex8], uv + np.float32(37.0), ix, 7, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, knobs=knobs)
    both(product, oracle, [tex8], uv - np.float32(2.5), ix, 6, addr=ot.MIRROR, promo=ot.PROMO_NEAREST, knobs=knobs)
    both(product, oracle, [texf], uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_TRANSPARENT, knobs=knobs)
    both(product, oracle, [texf], uv * np.float32(2.0) - np.float32(0.5), ix, 5, addr=ot.BORDER, promo=ot.PROMO_FORCE_OPAQUE, border_alpha=0.2, sat=False, knobs=knobs)
    both(product, oracle, [noise], uv, ix, 6, addr=ot.CLAMP, promo=ot.PROMO_FORCE_OPAQUE, knobs=knobs)
    both(product, oracle, [noise], uv, ix, 5, addr=ot.WRAP, promo=ot.PROMO_NEAREST, fmt=ot.FMT_2STATE, sat=False, knobs=knobs)
    both(product, oracle, [const], uv, ix, 6, addr=ot.MIRROR_ONCE, promo=ot.PROMO_FORCE_OPAQUE, sat=False, knobs=knobs)


def test_texel_walks_of_boxes_wider_than_the_visit_rings_offsets(product, oracle):
    """The deferred pass keeps a visit as (owner, x offset, y offset) in one word, 13 bits per offset; a micro-triangle whose raster box is wider or higher than
    8192 texels is offered to the wave a piece of a row at a time instead (bake_kernels.hip: generic_dense, `big`).  Level-5 triangles 300 texture periods long
    and two texels thin, along x and along y (micro-triangle boxes of ~9600 x 2 texels), next to ordinary ones; counts decide under the Nearest promotion,
    so every covered texel must vote exactly once."""
    knobs = [(ot.KNOB_GENERIC_PASS, 2)]
    tex8 = ot.foliage_texture(9, 1024, 1024, feature=32)
    tri = np.array([[[0.1, 0.5], [300.1, 0.501], [150.0, 0.5025]],
                    [[0.5, 0.2], [0.502, 300.2], [0.5005, 150.0]],
                    [[0.2, 0.2], [0.45, 0.22], [0.3, 0.4]],
                    [[3.1, 0.7], [303.1, 0.7], [150.0, 0.7021]]], np.float32)
    uv = np.ascontiguousarray(tri.reshape(-1, 2)); ix = np.arange(12, dtype=np.uint32)
    for promo in (ot.PROMO_NEAREST, ot.PROMO_FORCE_OPAQUE):
        both(product, oracle, [tex8], uv, ix, 5, addr=ot.WRAP, promo=promo, knobs=knobs)
    both(product, oracle, [tex8], uv, ix, 5, filt=ot.NEAREST, addr=ot.MIRROR, promo=ot.PROMO_NEAREST, sat=False, knobs=knobs)


def test_zeroing_ahead_never_leaves_stale_bytes(product):
    """Compressed transfer, round 6: while the device bakes, the baker's helper threads zero the idle result block the previous bake left, and the expansion leaves
    codec blocks of zeros in the zeroed pieces alone.  Bakes of DIFFERENT workloads alternate on one baker, so the block a result lands in holds another
    result's bytes before it is zeroed: every result must equal the plain copy of the same bake (a fresh baker), whether the block was zeroed completely,
    partly (a larger result than the one before), or not at all (first bake, results kept alive: no idle block)."""
    import bench, workloads as wl
    texA, uvA, ixA, lvA, kwA = wl.workload("c2", 60000)
    texB = ot.foliage_texture(123, 2048, 2048, feature=96)
    uvB, ixB = ot.random_triangles(9, 45000, 0.004)
    kwA = dict(kwA); lvlA = kwA.pop("level")
    plain = {}
    b0 = product.create_baker(); product.set_knob(b0, ot.KNOB_RESULT_TRANSFER, ot.TRANSFER_PLAIN)
    tA0 = product.create_texture(b0, [texA], alpha_cutoff=0.5); tB0 = product.create_texture(b0, [texB], alpha_cutoff=0.5)
    descs0 = {"A": ot.make_desc(tA0, uvA, ixA, lvlA, **kwA), "B": ot.make_desc(tB0, uvB, ixB, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=ot.FLAG_THREADS),
              "A2": ot.make_desc(tA0, uvA[:3 * 30000], ixA[:3 * 30000], lvlA, **kwA)}
    for k, d in descs0.items():
        plain[k] = product.bake(b0, d, want_stats=False)
    assert min(plain[k].array_data.size for k in plain) >= (32 << 20)
    product.destroy_texture(b0, tA0); product.destroy_texture(b0, tB0); product.destroy_baker(b0)
    b = product.create_baker(); product.set_knob(b, ot.KNOB_RESULT_TRANSFER, ot.TRANSFER_COMPRESSED)
    tA = product.create_texture(b, [texA], alpha_cutoff=0.5); tB = product.create_texture(b, [texB], alpha_cutoff=0.5)
    descs = {"A": ot.make_desc(tA, uvA, ixA, lvlA, **kwA), "B": ot.make_desc(tB, uvB, ixB, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, flags=ot.FLAG_THREADS),
             "A2": ot.make_desc(tA, uvA[:3 * 30000], ixA[:3 * 30000], lvlA, **kwA)}
    zeroed = skipped = 0
    for k in ("A", "B", "A", "A2", "B", "A", "B", "A2", "A2", "A"):
        r = product.bake(b, descs[k], want_stats=False)
        tm = bench.get_timings(product, b)
        assert tm.resultTransfer == ot.TRANSFER_COMPRESSED
        assert r.same_as(plain[k]), (k, r.diff(plain[k]))
        assert tm.expandSkippedBytes <= tm.prefilledBytes
        zeroed += tm.prefilledBytes; skipped += tm.expandSkippedBytes
    assert zeroed > 0 and skipped > 0, (zeroed, skipped)   # (the mechanism did run: the sequence above re-uses the pool's block from the second bake on)
    product.destroy_texture(b, tA); product.destroy_texture(b, tB); product.destroy_baker(b)


def test_near_duplicate_merge_and_budget_at_scale(product, oracle):
    """The serial reducers (near-duplicate LSH merge, maxArrayDataSize budget) work on the device's 2-bit packed states, uniform work items as
    (state, level): a level-8 bake of 20 000 triangles needs 40 MB on the host for them, not the 2.6 GB of one byte x 2 per micro-triangle of every
    item that the reference (and the oracle) hold.  Result = the oracle's."""
    import workloads as wl
    tex, uv, ix, lv, kw = wl.workload("c2", 20000)
    kw = dict(kw); level = kw.pop("level")
    for flags, extra in ((ot.FLAG_THREADS | ot.FLAG_NEAR_DUP, {}), (ot.FLAG_THREADS, {"max_array": 12 << 20})):
        out = []
        for lib in (product, oracle):
            b = lib.create_baker()
            t = lib.create_texture(b, [tex], alpha_cutoff=0.5)
            d = ot.make_desc(t, uv, ix, level, flags=flags, **kw)
            if "max_array" in extra:
                d.maxArrayDataSize = extra["max_array"]
            out.append(lib.bake(b, d))
            lib.destroy_texture(b, t); lib.destroy_baker(b)
        assert out[0].same_as(out[1]), out[0].diff(out[1])
        assert len(out[0].descs) > 50


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 101])
def test_curve_free_regions_device_against_oracle_on_adversarial_inputs(product, oracle, seed):
    """The curve-free-region test (region_curve.h) is audited on the HOST compilation of the header (tests/test_region_curve_audit.py); this bakes the audit's
    adversarial cases -- thin work items (edge-free verdict), textures without a summed-area table, alpha within ulps of the cutoff, FP32 values of +-1000,
    non-power-of-two sizes, every address mode / promotion, both formats -- through the DEVICE code (items, 4096-tiles and 64-groups all culled) and compares
    every result array with the oracle: a divergence between the device and the host evaluation of the header (an FMA contraction, a fast-math flag) shows here."""
    import region_cases
    want = []
    region_cases.run(oracle, seed=seed, tri_seed_base=2000 if seed == 11 else 7000 + seed, each=lambda k, r: want.append(r))
    bad = []
    region_cases.run(product, seed=seed, tri_seed_base=2000 if seed == 11 else 7000 + seed, each=lambda k, r: (None if r.same_as(want[k]) else bad.append((k, r.diff(want[k])))))
    assert len(want) == 84 and not bad, bad[:3]




@pytest.mark.gpu
def test_result_transfer_modes_give_the_same_bytes(product, oracle):
    """ommxBakerKnob_ResultTransfer: plain copy, streamed placement and the compressed form (codec stream + helper threads) of ommCpuBake, on results
    from a few hundred kilobytes (always the plain copy) to well above the 32 MiB where the compressed form starts; noise-like states (a 0 / 1 random texture:
    the stream does not shrink below half and the array crosses the link as it is); a bake without ommCpuBakeFlags_EnableInternalThreads keeps to the streamed
    form by default; the smallest against the oracle."""
    import bench, workloads as wl
    rng = np.random.RandomState(3)
    # (the compressed form has the gather produce the codec's unit codes when every OMM is a multiple of 16 bytes in the bake's own packing: the "small" case mixes
    #  levels 0 - 2 into a large bake, "uniform" emits the uniform work items instead of special indices (pattern fills), "2state" packs 2-bit states into 1 bit)
    cases = [("c2", 2000, None), ("c2", 60000, None), ("noise", 600, (rng.rand(1024, 1024) > 0.5).astype(np.float32)),
             ("small", 110000, None), ("uniform", 60000, None), ("2state", 120000, None)]
    for kind, n, tex_override in cases:
        if kind in ("small", "uniform", "2state"):
            tex, uv, ix, lv, kw = wl.workload("c2", n); kw = dict(kw)
            if kind == "small":
                lv = np.full(n, kw["level"], np.uint8); lv[::7] = 0; lv[1::7] = 1; lv[2::7] = 2; lv[3::7] = 3; kw["levels"] = lv
            elif kind == "uniform":
                kw["flags"] = kw.get("flags", ot.FLAG_THREADS) | ot.FLAG_NO_SPECIAL
            else:
                kw["fmt"] = ot.FMT_2STATE
        elif kind == "noise":   # micro-triangles of a texel each on 0 / 1 noise, promotion by majority: UO / UT at random, units of 64 states rarely repeat one
            tex = tex_override; uv, ix = ot.random_triangles(77, n, 0.5); lv = None
            kw = dict(level=9, addr=ot.WRAP, promo=ot.PROMO_NEAREST)
        else:
            tex, uv, ix, lv, kw = wl.workload(kind, n)
        kw = dict(kw); lvl = kw.pop("level")
        b = product.create_baker(); t = product.create_texture(b, [tex], alpha_cutoff=0.5)
        d = ot.make_desc(t, uv, ix, lvl, **kw)
        got = {}
        for mode in (ot.TRANSFER_PLAIN, ot.TRANSFER_STREAMED, ot.TRANSFER_COMPRESSED, ot.TRANSFER_AUTO):
            product.set_knob(b, ot.KNOB_RESULT_TRANSFER, mode)
            got[mode] = (product.bake(b, d, want_stats=False), bench.get_timings(product, b))
        ref = got[ot.TRANSFER_PLAIN][0]
        assert got[ot.TRANSFER_PLAIN][1].resultTransfer == ot.TRANSFER_PLAIN
        for mode, (r, tm) in got.items():
            assert r.same_as(ref), (kind, n, mode, r.diff(ref))
        big = ref.array_data.size >= (32 << 20)
        tmc = got[ot.TRANSFER_COMPRESSED][1]
        if not big:
            assert tmc.resultTransfer == ot.TRANSFER_PLAIN
        elif tex_override is None:
            assert tmc.resultTransfer == ot.TRANSFER_COMPRESSED and 0 < tmc.compressedBytes < ref.array_data.size // 2 and tmc.expandThreads >= 1, (tmc.resultTransfer, tmc.compressedBytes)
        else:
            assert tmc.resultTransfer == ot.TRANSFER_PLAIN and tmc.compressMs > 0, (tmc.resultTransfer, tmc.compressedBytes, ref.array_data.size)   # (tried, incompressible)
        # without the caller's permission to use threads the default is never the compressed form
        product.set_knob(b, ot.KNOB_RESULT_TRANSFER, ot.TRANSFER_AUTO)
        r0 = product.bake(b, ot.make_desc(t, uv, ix, lvl, **dict(kw, flags=kw.get("flags", ot.FLAG_THREADS) & ~ot.FLAG_THREADS)), want_stats=False)
        assert bench.get_timings(product, b).resultTransfer != ot.TRANSFER_COMPRESSED and r0.same_as(ref)
        if n <= 2000:
            ob = oracle.create_baker(); otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
            want = oracle.bake(ob, ot.make_desc(otx, uv, ix, lvl, **kw), want_stats=False)
            oracle.destroy_texture(ob, otx); oracle.destroy_baker(ob)
            assert ref.same_as(want), ref.diff(want)
        product.destroy_texture(b, t); product.destroy_baker(b)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [2, 3, 8])
def test_multi_device_ommCpuBake_equals_one_device(product, oracle, devices):
    """ommxBakerKnob_Devices: ommCpuBake spread over N devices of the process (one host thread per device; on a one-GPU box the ranks share the device, which
    exercises everything but the peer copies of the texture): replicated set-up, every rank's share of the classification, the item metadata summed through
    host memory, the replicated tail, every rank's own blocks as a codec stream, the host writing each block from its owner's stream.  Same bytes as the
    one-device bake -- mixed levels incl. blocks below a codec block (levels 3 - 5), 2-state and 4-state, uniform items with special indices disabled,
    a noise texture (raw contributions), dynamic levels -- and the smallest against the oracle."""
    import bench, workloads as wl
    rng = np.random.RandomState(5)
    cases = []
    tex, uv, ix, lv, kw = wl.workload("c2", 3000); cases.append(("c2", tex, uv, ix, lv, dict(kw)))
    tex, uv, ix, lv, kw = wl.workload("c4", 6000); cases.append(("c4", tex, uv, ix, lv, dict(kw)))
    tex, uv, ix, lv, kw = wl.workload("cards", 300); cases.append(("cards", tex, uv, ix, lv, dict(kw)))
    tex, uv, ix, lv, kw = wl.workload("c1", 5000); cases.append(("c1-2state-nospecial", tex, uv, ix, lv, dict(kw, fmt=ot.FMT_2STATE, flags=ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL)))
    uvn, ixn = ot.random_triangles(78, 120, 0.4)
    cases.append(("noise", (rng.rand(512, 512) > 0.5).astype(np.float32), uvn, ixn, (3 + (np.arange(120) % 6)).astype(np.uint8), dict(level=8, addr=ot.WRAP, promo=ot.PROMO_NEAREST)))
    for name, tex, uv, ix, lv, kw in cases:
        lvl = kw.pop("level")
        b = product.create_baker(); t = product.create_texture(b, [tex], alpha_cutoff=0.5)
        d = ot.make_desc(t, uv, ix, lvl, levels=lv, **kw)
        one = product.bake(b, d)
        product.set_knob(b, ot.KNOB_DEVICES, devices)
        many = product.bake(b, d)
        tm = bench.get_timings(product, b)
        assert tm.devices == devices, (name, tm.devices)
        assert many.same_as(one), (name, devices, many.diff(one))
        again = product.bake(b, d)   # (shadow bakers, texture copies and pools are reused)
        assert again.same_as(one), (name, devices, "second bake", again.diff(one))
        product.set_knob(b, ot.KNOB_DEVICES, 0)
        assert product.bake(b, d).same_as(one)
        if name == "c2":
            ob = oracle.create_baker(); otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
            want = oracle.bake(ob, ot.make_desc(otx, uv, ix, lvl, levels=lv, **kw))
            oracle.destroy_texture(ob, otx); oracle.destroy_baker(ob)
            assert many.same_as(want), many.diff(want)
        product.destroy_texture(b, t); product.destroy_baker(b)


@pytest.mark.gpu
def test_compressed_transfer_from_concurrent_callers(product):
    """Several caller threads bake large results on ONE baker at the same time (docs/integration_guide.md:434): each call has its own working set and pinned
    staging, the baker's helper threads expand one result at a time; and a multi-device bake next to them.  Same bytes as a bake on its own."""
    import threading, bench, workloads as wl
    tex, uv, ix, lv, kw = wl.workload("c2", 60000)
    kw = dict(kw); lvl = kw.pop("level")
    b = product.create_baker(); t = product.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, lvl, **kw)
    product.set_knob(b, ot.KNOB_RESULT_TRANSFER, ot.TRANSFER_COMPRESSED)
    ref = product.bake(b, d, want_stats=False)
    assert bench.get_timings(product, b).resultTransfer == ot.TRANSFER_COMPRESSED
    out, errs = {}, []
    def work(k):
        try:
            for rep in range(3):
                out[(k, rep)] = product.bake(b, d, want_stats=False)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for th in ths: th.start()
    for th in ths: th.join()
    assert not errs, errs
    assert len(out) == 12 and all(r.same_as(ref) for r in out.values())
    product.destroy_texture(b, t); product.destroy_baker(b)

