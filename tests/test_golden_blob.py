"""Bit-level golden vectors: the serialized 8x8 StandardCircle / level-4 bake embedded in the reference's
test-suite (test_omm_bake_cpu.cpp:2034-2304).  The input blob is baked and every output array is compared
byte-for-byte with the matching output blob."""
import json
import os
import numpy as np
import pytest
import blobfmt
import ommtest as ot

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLOBS = {k: bytes.fromhex(v) for k, v in json.load(open(os.path.join(GOLDEN, "blobs.json"))).items()}
PAIRS = [("input_v1_4_0", "output_v1_4_0"), ("input_v1_5_0", "output_v1_5_0"), ("input_compress_v1_5_0", "output_compress_v1_5_0")]
STATS = dict(O=152, T=232, UT=70, UO=58, FO=0, FT=0, FUO=0, FUT=0)  # :2068-2073


def bake_input_blob(lib, inp):
    b = lib.create_baker()
    t = inp["texture"]
    tex = lib.create_texture(b, t["mips"], alpha_cutoff=t["alphaCutoff"], disable_zorder=bool(t["flags"] & 1))
    d = ot.default_bake_desc()
    keep = [np.frombuffer(inp["texCoords"], np.uint8).copy(), np.frombuffer(inp["indexBuffer"], np.uint8).copy()]
    d.bakeFlags = inp["bakeFlags"]
    d.texture = tex
    d.runtimeSamplerDesc = ot.SamplerDesc(inp["addressingMode"], inp["filter"], inp["borderAlpha"])
    d.alphaMode = inp["alphaMode"]
    d.texCoordFormat = inp["texCoordFormat"]
    d.texCoords = keep[0].ctypes.data
    d.texCoordStrideInBytes = inp["texCoordStrideInBytes"]
    d.indexFormat = inp["indexFormat"]
    d.indexBuffer = keep[1].ctypes.data
    d.indexCount = inp["indexCount"]
    for k in ("dynamicSubdivisionScale", "rejectionThreshold", "alphaCutoff", "alphaCutoffLessEqual", "alphaCutoffGreater",
              "format", "unknownStatePromotion", "unresolvedTriState", "maxSubdivisionLevel", "maxArrayDataSize", "maxWorkloadSize"):
        setattr(d, k, inp[k])
    res = lib.bake(b, d)
    lib.destroy_texture(b, tex)
    lib.destroy_baker(b)
    return res


def check_against_output_blob(res, out):
    assert res.array_data.tobytes() == out["arrayData"]
    assert res.desc_bytes == out["descArray"]
    assert res.index_format == out["indexFormat"]
    assert res.index.tobytes() == out["indexBuffer"]
    pack = lambda h: b"".join(int(c).to_bytes(4, "little") + int(l).to_bytes(2, "little") + int(f).to_bytes(2, "little") for c, l, f in h)
    assert pack(res.array_hist) == out["descArrayHistogram"]
    assert pack(res.index_hist) == out["indexHistogram"]


def test_blob_digests(oracle):
    import ctypes as C
    oracle.dll.orc_xxh64.restype = C.c_uint64
    oracle.dll.orc_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    for name, blob in BLOBS.items():
        blobfmt.parse_blob(blob, xxh64=lambda d, s: oracle.dll.orc_xxh64(d, len(d), s))


@pytest.mark.parametrize("iname,oname", PAIRS)
def test_oracle_reproduces_golden_blob(oracle, iname, oname):
    inp = blobfmt.parse_blob(BLOBS[iname])["inputs"][0]
    out = blobfmt.parse_blob(BLOBS[oname])["results"][0]
    res = bake_input_blob(oracle, inp)
    check_against_output_blob(res, out)
    assert res.stats_tuple() == STATS


def test_all_output_blobs_agree():
    outs = [blobfmt.parse_blob(b)["results"][0] for k, b in BLOBS.items() if k.startswith("output")]
    for o in outs[1:]:
        assert o["arrayData"] == outs[0]["arrayData"] and o["indexBuffer"] == outs[0]["indexBuffer"]
