"""Reader for the reference's serialized bake blobs (test-side only).

Layout restated from /root/reference/libraries/omm-lib/src/serialize_impl.cpp:81-276,351-582 and
texture_impl.h:232-336.  Used to turn the golden blobs embedded in the reference's
test-suite (tests/golden/blobs.json) into bake inputs / expected outputs.
"""
import struct
import numpy as np


def lz4_block_decompress(src, out_size):
    """Plain LZ4 block format (lz4.org block spec)."""
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        tok = src[i]; i += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = src[i]; i += 1; ll += b
                if b != 255:
                    break
        out += src[i:i + ll]; i += ll
        if i >= n:
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = src[i]; i += 1; ml += b
                if b != 255:
                    break
        ml += 4
        start = len(out) - off
        for k in range(ml):
            out.append(out[start + k])
    assert len(out) == out_size, (len(out), out_size)
    return bytes(out)


class Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def take(self, n):
        b = self.d[self.o:self.o + n]
        assert len(b) == n, "blob truncated"
        self.o += n
        return b

    def fmt(self, f):
        v = struct.unpack_from("<" + f, self.d, self.o)
        self.o += struct.calcsize("<" + f)
        return v[0] if len(v) == 1 else v


def morton_to_xy(i):
    def compact(x):
        x &= 0x55555555
        x = (x | (x >> 1)) & 0x33333333
        x = (x | (x >> 2)) & 0x0F0F0F0F
        x = (x | (x >> 4)) & 0x00FF00FF
        x = (x | (x >> 8)) & 0x0000FFFF
        return x
    return compact(i), compact(i >> 1)


def xy_to_morton(x, y):
    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(x) | (spread(y) << 1)


def parse_blob(blob, xxh64=None):
    """Returns dict(version=..., inputs=[...], results=[...]).  xxh64(data, seed)->int optionally verifies the digest."""
    r = Reader(blob)
    digest = r.fmt("Q")
    if xxh64 is not None:
        assert xxh64(blob[8:], 42) == digest, "blob digest mismatch"
    major, minor, patch, ver, flags = r.fmt("iiiii")
    decomp = r.fmt("i") if ver >= 2 else 0
    body = blob[r.o:]
    if decomp:
        body = lz4_block_decompress(body, decomp)
    r = Reader(body)
    out = dict(version=(major, minor, patch, ver), flags=flags, inputs=[], results=[])
    for _ in range(r.fmt("i")):
        d = {}
        d["bakeFlags"] = r.fmt("I")
        nm = r.fmt("i")
        mips = [r.fmt("iiffQQQ") for _ in range(nm)]
        tiling = r.fmt("i")
        if ver >= 3:
            texflags, tex_cutoff = r.fmt("I"), r.fmt("f")
        else:
            texflags, tex_cutoff = (0 if tiling == 1 else 1), -1.0
        texfmt = r.fmt("i")
        data = r.take(r.fmt("Q"))
        sat = r.take(r.fmt("Q"))
        px = 4 if texfmt == 1 else 1
        arrays = []
        for (w, h, rw, rh, off, ne, offsat) in mips:
            raw = np.frombuffer(data, np.float32 if texfmt == 1 else np.uint8, count=ne, offset=off)
            a = np.empty((h, w), raw.dtype)
            if tiling == 1:  # Morton-Z internal order
                for j in range(h):
                    for i in range(w):
                        a[j, i] = raw[xy_to_morton(i, j)]
            else:
                a[:] = raw.reshape(h, w)
            arrays.append(a)
        d["texture"] = dict(mips=arrays, tiling=tiling, flags=texflags, alphaCutoff=tex_cutoff, format=texfmt, has_sat=len(sat) != 0)
        d["addressingMode"], d["filter"], d["borderAlpha"], d["alphaMode"] = r.fmt("iifi")
        d["texCoordFormat"] = r.fmt("i")
        d["texCoords"] = r.take(r.fmt("Q"))
        d["texCoordStrideInBytes"] = r.fmt("I")
        d["indexFormat"], d["indexCount"] = r.fmt("iI")
        isz = {2: 1, 0: 2, 1: 4}[d["indexFormat"]]
        d["indexBuffer"] = r.take(isz * d["indexCount"])
        d["dynamicSubdivisionScale"], d["rejectionThreshold"], d["alphaCutoff"] = r.fmt("fff")
        d["alphaCutoffLessEqual"], d["alphaCutoffGreater"], d["format"] = r.fmt("iii")
        nf = r.fmt("Q")
        d["formats"] = r.take(4 * nf)
        d["unknownStatePromotion"] = r.fmt("i")
        d["unresolvedTriState"] = r.fmt("i") if ver >= 2 else -4
        d["maxSubdivisionLevel"] = r.fmt("B")
        d["maxArrayDataSize"] = r.fmt("I") if ver >= 4 else 0xFFFFFFFF
        ns = r.fmt("Q")
        d["subdivisionLevels"] = r.take(ns)
        d["maxWorkloadSize"] = r.fmt("Q")
        if d["texture"]["has_sat"] and ver < 3:
            d["texture"]["alphaCutoff"] = d["alphaCutoff"]
        out["inputs"].append(d)
    for _ in range(r.fmt("i")):
        res = {}
        res["arrayData"] = r.take(r.fmt("I"))
        n = r.fmt("I"); res["descArray"] = r.take(8 * n)
        n = r.fmt("I"); res["descArrayHistogram"] = r.take(8 * n)
        res["indexFormat"] = r.fmt("i")
        n = r.fmt("I"); res["indexBuffer"] = r.take({2: 1, 0: 2, 1: 4}[res["indexFormat"]] * n)
        res["indexCount"] = n
        n = r.fmt("I"); res["indexHistogram"] = r.take(8 * n)
        out["results"].append(res)
    assert r.o == len(body), "trailing bytes in blob: %d" % (len(body) - r.o)
    return out
