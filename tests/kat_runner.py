"""Runs the reference's known-answer tests (tests/kat_cases.py) through one implementation of the C ABI."""
import json
import os
import numpy as np
import ommtest as ot
from kat_cases import CONFIGS, hex_grid

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_tex_cache = {}


def texture_array(kind, size, param):
    key = (kind, size, param)
    if key not in _tex_cache:
        _tex_cache[key] = ot.kat_texture(kind, size[0], size[1], param)
    return _tex_cache[key]


def pack_uv(uv, fmt):
    """glm::packHalf2x16 / packUnorm2x16 of the reference tests (test_omm_bake_cpu.cpp:116-141)."""
    uv = np.asarray(uv, np.float32).reshape(-1, 2)
    if fmt == "fp16":
        h = uv.astype(np.float16).view(np.uint16).astype(np.uint32)
        return (h[:, 0] | (h[:, 1] << 16)).astype(np.uint32), ot.UV16_FLOAT
    if fmt == "unorm16":
        q = np.round(np.clip(uv, 0, 1) * np.float32(65535.0)).astype(np.uint32)
        return (q[:, 0] | (q[:, 1] << 16)).astype(np.uint32), ot.UV16_UNORM
    return uv, ot.UV32_FLOAT


def run_case(lib, baker, c, config="Default"):
    zo, f32, sat = CONFIGS[config]
    opt = c["opt"]
    cutoff = opt.get("alpha_cutoff", 0.5)
    tex_arr = texture_array(c["tex"], c["size"], c["param"])
    tex = lib.create_texture(baker, [tex_arr], alpha_cutoff=cutoff if sat else -1.0, disable_zorder=zo)
    if isinstance(c["uv"], str):
        uv, ix = hex_grid()
    else:
        uv, ix = np.array(c["uv"], np.float32), np.array(c["ix"], np.uint32)
    uvp, uvfmt = pack_uv(uv, opt.get("uv_format"))
    flags = ot.FLAG_THREADS | (ot.FLAG_FORCE32 if f32 else 0) | (ot.FLAG_NEAR_DUP if opt.get("merge_similar") else 0)
    d = ot.make_desc(tex, uvp, ix, c["level"], alpha_cutoff=cutoff, fmt=opt.get("fmt", ot.FMT_4STATE),
                     addr=opt.get("addr", ot.CLAMP), promo=ot.PROMO_NEAREST, flags=flags,
                     le=opt.get("le", ot.T), gt=opt.get("gt", ot.O), dyn_scale=opt.get("dyn_scale", 0.0),
                     unresolved=opt.get("unresolved", ot.SPECIAL_FUO), uv_format=uvfmt)
    res = lib.bake(baker, d)
    lib.destroy_texture(baker, tex)
    return res


# ---- leaflet ----
_leaf = None


def leaflet_mips(n):
    """mips[0] = blue/255; mips[k] = 2x2 box filter, summation order of test_omm_bake_cpu.cpp:671-695."""
    global _leaf
    if _leaf is None:
        meta = json.load(open(os.path.join(GOLDEN, "leaflet.json")))
        b = np.fromfile(os.path.join(GOLDEN, "leaflet_b.bin"), np.uint8).reshape(meta["height"], meta["width"])
        _leaf = [b.astype(np.float32) / np.float32(255.0)]
    while len(_leaf) < n:
        t = _leaf[-1]
        hh, hw = t.shape[0] // 2, t.shape[1] // 2
        p0 = t[0:2 * hh:2, 0:2 * hw:2]
        p1 = t[1:2 * hh:2, 0:2 * hw:2]
        p2 = t[0:2 * hh:2, 1:2 * hw:2]
        p3 = t[1:2 * hh:2, 1:2 * hw:2]
        _leaf.append((((p0 + p1) + p2) + p3) * np.float32(0.25))
    return _leaf[:n]


def run_leaflet_mip(lib, baker, mip_start, num_mip, cutoff, config="Default"):
    zo, f32, sat = CONFIGS[config]
    mips = leaflet_mips(mip_start + num_mip)
    w0, h0 = mips[mip_start].shape[1], mips[mip_start].shape[0]
    chain = []
    for k in range(num_mip):  # vmtest::TextureImpl: mip k is (w0 >> k, h0 >> k), util/omm.h:35-50
        mw, mh = w0 // (1 << k), h0 // (1 << k)
        src = mips[mip_start + k]
        chain.append(np.ascontiguousarray(np.float32(1.0) - src[:mh, :mw]))
    tex = lib.create_texture(baker, chain, alpha_cutoff=cutoff if sat else -1.0, disable_zorder=zo)
    uv = np.array([0.05, 0.1, 0.1, 0.9, 0.9, 0.9], np.float32)
    flags = ot.FLAG_THREADS | (ot.FLAG_FORCE32 if f32 else 0)
    d = ot.make_desc(tex, uv, np.array([0, 1, 2], np.uint32), 6, alpha_cutoff=cutoff, flags=flags)
    res = lib.bake(baker, d)
    lib.destroy_texture(baker, tex)
    return res


def run_leaflet_level(lib, baker, level, config="Default", max_workload=0xFFFFFFFFFFFFFFFF, expect=ot.SUCCESS):
    zo, f32, sat = CONFIGS[config]
    tex = lib.create_texture(baker, [np.float32(1.0) - leaflet_mips(1)[0]], alpha_cutoff=0.5 if sat else -1.0, disable_zorder=zo)
    uv = np.array([0.35, 0.1, 0.1, 0.9, 0.9, 0.8], np.float32)
    flags = ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL | (ot.FLAG_FORCE32 if f32 else 0)
    d = ot.make_desc(tex, uv, np.array([0, 1, 2], np.uint32), level, flags=flags, max_workload=max_workload)
    res = lib.bake(baker, d, expect=expect)
    lib.destroy_texture(baker, tex)
    return res
