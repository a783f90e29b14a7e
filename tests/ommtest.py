"""ctypes test driver for the omm C ABI (include/omm_mi355x.h).

The same driver runs a bake through the CPU oracle (oracle/libomm_oracle.so, symbols prefixed
``oracle_``) or through the product HIP library (omm_amd/lib/libomm-lib.so, unprefixed
symbols), so parity tests read like the reference's own tests
(/root/reference/support/tests/test_omm_bake_cpu.cpp:168-207).
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- enums (include/omm_mi355x.h) ----
SUCCESS, FAILURE, INVALID_ARGUMENT, INSUFFICIENT_SCRATCH, NOT_IMPLEMENTED, WORKLOAD_TOO_BIG = range(6)
T, O, UT, UO = 0, 1, 2, 3
FMT_2STATE, FMT_4STATE = 1, 2
PROMO_NEAREST, PROMO_FORCE_OPAQUE, PROMO_FORCE_TRANSPARENT = 0, 1, 2
UV16_UNORM, UV16_FLOAT, UV32_FLOAT = 0, 1, 2
IDX_U16, IDX_U32, IDX_U8 = 0, 1, 2
WRAP, MIRROR, CLAMP, BORDER, MIRROR_ONCE = range(5)
NEAREST, LINEAR = 0, 1
TEX_UNORM8, TEX_FP32 = 0, 1
FLAG_THREADS, FLAG_NO_SPECIAL, FLAG_FORCE32, FLAG_NO_DEDUP, FLAG_NEAR_DUP, FLAG_VALIDATION, FLAG_ALLOW8 = (1 << i for i in range(7))
TEXFLAG_DISABLE_ZORDER = 1
SPECIAL_FT, SPECIAL_FO, SPECIAL_FUT, SPECIAL_FUO = -1, -2, -3, -4
KNOB_RESERVED0, KNOB_SHARD_CHUNK_BYTES, KNOB_STREAM_CHUNKS, KNOB_GENERIC_PASS, KNOB_RETAIN_MEMORY, KNOB_RESULT_TRANSFER, KNOB_EXPAND_THREADS, KNOB_DEVICES, KNOB_HELPER_AFFINITY, KNOB_ZERO_AHEAD = range(10)   # ommxBakerKnob
TRANSFER_AUTO, TRANSFER_PLAIN, TRANSFER_STREAMED, TRANSFER_COMPRESSED = range(4)   # ommxResultTransfer


class SamplerDesc(C.Structure):
    _fields_ = [("addressingMode", C.c_int), ("filter", C.c_int), ("borderAlpha", C.c_float)]


class AllocatorInterface(C.Structure):
    _fields_ = [("allocate", C.c_void_p), ("reallocate", C.c_void_p), ("free", C.c_void_p), ("userArg", C.c_void_p)]


MESSAGE_CB = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)


class MessageInterface(C.Structure):
    _fields_ = [("messageCallback", MESSAGE_CB), ("userArg", C.c_void_p)]


class BakerCreationDesc(C.Structure):
    _fields_ = [("type", C.c_int), ("memoryAllocatorInterface", AllocatorInterface), ("messageInterface", MessageInterface)]


class TextureMipDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rowPitch", C.c_uint32), ("textureData", C.c_void_p)]


class TextureDesc(C.Structure):
    _fields_ = [("format", C.c_int), ("flags", C.c_int), ("mips", C.POINTER(TextureMipDesc)), ("mipCount", C.c_uint32),
                ("alphaCutoff", C.c_float)]


class BakeInputDesc(C.Structure):
    _fields_ = [
        ("bakeFlags", C.c_uint32), ("texture", C.c_void_p), ("runtimeSamplerDesc", SamplerDesc), ("alphaMode", C.c_int),
        ("texCoordFormat", C.c_int), ("texCoords", C.c_void_p), ("texCoordStrideInBytes", C.c_uint32),
        ("indexFormat", C.c_int), ("indexBuffer", C.c_void_p), ("indexCount", C.c_uint32),
        ("dynamicSubdivisionScale", C.c_float), ("rejectionThreshold", C.c_float), ("alphaCutoff", C.c_float),
        ("nearDuplicateDeduplicationFactor", C.c_float), ("alphaCutoffLessEqual", C.c_int), ("alphaCutoffGreater", C.c_int),
        ("format", C.c_int), ("formats", C.c_void_p), ("unknownStatePromotion", C.c_int), ("unresolvedTriState", C.c_int),
        ("maxSubdivisionLevel", C.c_uint8), ("maxArrayDataSize", C.c_uint32), ("subdivisionLevels", C.c_void_p),
        ("maxWorkloadSize", C.c_uint64)]


assert C.sizeof(BakeInputDesc) == 136  # /root/reference/libraries/omm-lib/src/serialize_impl.cpp:86


class MicromapDesc(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("subdivisionLevel", C.c_uint16), ("format", C.c_uint16)]


class UsageCount(C.Structure):
    _fields_ = [("count", C.c_uint32), ("subdivisionLevel", C.c_uint16), ("format", C.c_uint16)]


class BakeResultDesc(C.Structure):
    _fields_ = [("arrayData", C.c_void_p), ("arrayDataSize", C.c_uint32), ("descArray", C.POINTER(MicromapDesc)),
                ("descArrayCount", C.c_uint32), ("descArrayHistogram", C.POINTER(UsageCount)),
                ("descArrayHistogramCount", C.c_uint32), ("indexBuffer", C.c_void_p), ("indexCount", C.c_uint32),
                ("indexFormat", C.c_int), ("indexHistogram", C.POINTER(UsageCount)), ("indexHistogramCount", C.c_uint32)]


class DebugStats(C.Structure):
    _fields_ = [("totalOpaque", C.c_uint64), ("totalTransparent", C.c_uint64), ("totalUnknownTransparent", C.c_uint64),
                ("totalUnknownOpaque", C.c_uint64), ("totalFullyOpaque", C.c_uint32), ("totalFullyTransparent", C.c_uint32),
                ("totalFullyUnknownOpaque", C.c_uint32), ("totalFullyUnknownTransparent", C.c_uint32),
                ("knownAreaMetric", C.c_float)]


def default_bake_desc():
    """ommCpuBakeInputDescDefault (include/omm_mi355x.h, reference omm.h:462-490)."""
    d = BakeInputDesc()
    d.bakeFlags = 0
    d.texture = None
    d.runtimeSamplerDesc = SamplerDesc(5, 2, 0.0)
    d.alphaMode = 2
    d.texCoordFormat = 3
    d.indexFormat = 3
    d.dynamicSubdivisionScale = 2.0
    d.rejectionThreshold = 0.0
    d.alphaCutoff = 0.5
    d.nearDuplicateDeduplicationFactor = 0.15
    d.alphaCutoffLessEqual = T
    d.alphaCutoffGreater = O
    d.format = FMT_4STATE
    d.unknownStatePromotion = PROMO_FORCE_OPAQUE
    d.unresolvedTriState = SPECIAL_FUO
    d.maxSubdivisionLevel = 8
    d.maxArrayDataSize = 0xFFFFFFFF
    d.maxWorkloadSize = 0xFFFFFFFFFFFFFFFF
    return d


# ---- library build/load helpers ----
def _run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))


def oracle_path():
    p = os.path.join(ROOT, "oracle", "libomm_oracle.so")
    src = os.path.join(ROOT, "oracle", "omm_oracle.c")
    if not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src):
        _run(["make", "-C", os.path.join(ROOT, "oracle")], ROOT)
    return p


def product_path():
    # OMM_AMD_LIBRARY: A/B builds of the same sources (kernel tuning experiments); default = the in-tree build
    if os.environ.get("OMM_AMD_LIBRARY"):
        return os.environ["OMM_AMD_LIBRARY"]
    p = os.path.join(ROOT, "omm_amd", "lib", "libomm-lib.so")
    if not os.path.exists(p):   # clean checkout: same as __graft_entry__.build() (hipcc cross-compiles gfx950 without a GPU)
        _run(["make", "-C", os.path.join(ROOT, "omm_amd", "csrc"), "-j4"], ROOT)
    return p


_KAT = None


def kat_lib():
    global _KAT
    if _KAT is None:
        src = os.path.join(ROOT, "tests", "native", "kat_textures.c")
        out = os.path.join(ROOT, "tests", "native", "libkat_textures.so")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            _run(["gcc", "-O2", "-msse4.1", "-ffp-contract=off", "-shared", "-fPIC", "-o", out, src, "-lm"], ROOT)
        _KAT = C.CDLL(out)
        _KAT.kat_fill_f32.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
        _KAT.kat_fill_u8.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
        _KAT.wl_noise_f32.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _KAT.wl_foliage_u8.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return _KAT


KAT_KINDS = dict(const=0, circle=1, diag8=2, corner=3, sine=4, mandelbrot=5, julia=6, uniform4=7, hexagons=8,
                 checker2=9, sine_u8=10, julia_u8=11)


def kat_texture(kind, w, h, param=0.0):
    """Returns a (h, w) float32 or uint8 array of the named reference-test texture."""
    k = KAT_KINDS[kind]
    if kind.endswith("_u8"):
        a = np.empty((h, w), np.uint8)
        kat_lib().kat_fill_u8(k, param, w, h, a.ctypes.data)
    else:
        a = np.empty((h, w), np.float32)
        kat_lib().kat_fill_f32(k, param, w, h, a.ctypes.data)
    return a


class BakeResult:
    """Host copy of an ommCpuBakeResultDesc."""

    def __init__(self, desc):
        self.array_data = np.ctypeslib.as_array(C.cast(desc.arrayData, C.POINTER(C.c_uint8)), (desc.arrayDataSize,)).copy() \
            if desc.arrayDataSize else np.zeros(0, np.uint8)
        n = desc.descArrayCount
        self.descs = np.array([(desc.descArray[i].offset, desc.descArray[i].subdivisionLevel, desc.descArray[i].format)
                               for i in range(n)], dtype=np.int64).reshape(n, 3) if n < 4096 else \
            np.frombuffer(C.string_at(desc.descArray, 8 * n), dtype=[("o", "<u4"), ("l", "<u2"), ("f", "<u2")])
        if n >= 4096:
            self.descs = np.stack([self.descs["o"].astype(np.int64), self.descs["l"].astype(np.int64),
                                   self.descs["f"].astype(np.int64)], axis=1)
        self.desc_bytes = C.string_at(desc.descArray, 8 * n) if n else b""
        self.index_format = desc.indexFormat
        isz = {IDX_U8: 1, IDX_U16: 2, IDX_U32: 4}[desc.indexFormat]
        idt = {IDX_U8: np.int8, IDX_U16: np.int16, IDX_U32: np.int32}[desc.indexFormat]
        self.index = np.frombuffer(C.string_at(desc.indexBuffer, isz * desc.indexCount), dtype=idt).copy() \
            if desc.indexCount else np.zeros(0, idt)
        self.array_hist = [(desc.descArrayHistogram[i].count, desc.descArrayHistogram[i].subdivisionLevel,
                            desc.descArrayHistogram[i].format) for i in range(desc.descArrayHistogramCount)]
        self.index_hist = [(desc.indexHistogram[i].count, desc.indexHistogram[i].subdivisionLevel,
                            desc.indexHistogram[i].format) for i in range(desc.indexHistogramCount)]
        self.stats = None
        self.stats2 = None   # ommDebugGetStats2 on the result object (adds knownAreaMetric)

    def stats2_key(self):
        """every field of ommDebugGetStats2's answer, the float as its bit pattern (it may be NaN: 0 / 0 for zero total area)"""
        if self.stats2 is None:
            return None
        s = self.stats2
        return (s.totalOpaque, s.totalTransparent, s.totalUnknownTransparent, s.totalUnknownOpaque, s.totalFullyOpaque, s.totalFullyTransparent,
                s.totalFullyUnknownOpaque, s.totalFullyUnknownTransparent, np.float32(s.knownAreaMetric).view(np.uint32).item())

    def same_as(self, other):
        k0, k1 = self.stats2_key(), other.stats2_key()
        return (np.array_equal(self.array_data, other.array_data) and self.desc_bytes == other.desc_bytes
                and self.index_format == other.index_format and np.array_equal(self.index, other.index)
                and self.array_hist == other.array_hist and self.index_hist == other.index_hist
                and (k0 is None or k1 is None or k0 == k1))

    def diff(self, other):
        out = []
        if not np.array_equal(self.array_data, other.array_data):
            if self.array_data.shape != other.array_data.shape:
                out.append("arrayData size %d vs %d" % (self.array_data.size, other.array_data.size))
            else:
                bad = np.nonzero(self.array_data != other.array_data)[0]
                out.append("arrayData differs in %d bytes, first at %d" % (bad.size, bad[0]))
        if self.desc_bytes != other.desc_bytes:
            out.append("descArray differs (%d vs %d descs)" % (len(self.desc_bytes) // 8, len(other.desc_bytes) // 8))
        if self.index_format != other.index_format:
            out.append("indexFormat %d vs %d" % (self.index_format, other.index_format))
        elif not np.array_equal(self.index, other.index):
            bad = np.nonzero(self.index != other.index)[0] if self.index.shape == other.index.shape else []
            out.append("indexBuffer differs in %d entries" % len(bad))
        if self.array_hist != other.array_hist:
            out.append("descArrayHistogram %r vs %r" % (self.array_hist, other.array_hist))
        if self.index_hist != other.index_hist:
            out.append("indexHistogram %r vs %r" % (self.index_hist, other.index_hist))
        if self.stats2_key() is not None and other.stats2_key() is not None and self.stats2_key() != other.stats2_key():
            out.append("ommDebugGetStats2 %r vs %r" % (self.stats2_key(), other.stats2_key()))
        return "; ".join(out)

    def stats_tuple(self):
        s = self.stats
        return dict(O=s.totalOpaque, T=s.totalTransparent, UT=s.totalUnknownTransparent, UO=s.totalUnknownOpaque,
                    FO=s.totalFullyOpaque, FT=s.totalFullyTransparent, FUO=s.totalFullyUnknownOpaque,
                    FUT=s.totalFullyUnknownTransparent)


class Lib:
    """One of the two implementations of the C ABI."""

    def __init__(self, which):
        if which == "oracle":
            self.dll = C.CDLL(oracle_path())
            self.prefix = "oracle_"
        elif which == "product":
            p = product_path()
            if not os.path.exists(p):
                raise RuntimeError("product library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
            self.dll = C.CDLL(p)
            self.prefix = ""
        else:
            raise ValueError(which)
        self.which = which
        f = self.fn
        f("ommCreateBaker").argtypes = [C.POINTER(BakerCreationDesc), C.POINTER(C.c_void_p)]
        f("ommDestroyBaker").argtypes = [C.c_void_p]
        f("ommCpuCreateTexture").argtypes = [C.c_void_p, C.POINTER(TextureDesc), C.POINTER(C.c_void_p)]
        f("ommCpuDestroyTexture").argtypes = [C.c_void_p, C.c_void_p]
        f("ommCpuBake").argtypes = [C.c_void_p, C.POINTER(BakeInputDesc), C.POINTER(C.c_void_p)]
        f("ommCpuDestroyBakeResult").argtypes = [C.c_void_p]
        f("ommCpuGetBakeResultDesc").argtypes = [C.c_void_p, C.POINTER(C.POINTER(BakeResultDesc))]
        f("ommDebugGetStats").argtypes = [C.c_void_p, C.POINTER(BakeResultDesc), C.POINTER(DebugStats)]
        f("ommDebugGetStats2").argtypes = [C.c_void_p, C.c_void_p, C.POINTER(DebugStats)]

    def fn(self, name):
        return getattr(self.dll, self.prefix + name)

    def create_baker(self, baker_type=1, callback=None):
        d = BakerCreationDesc()
        d.type = baker_type
        self._cb = MESSAGE_CB(callback) if callback else MESSAGE_CB()
        d.messageInterface.messageCallback = self._cb
        out = C.c_void_p()
        r = self.fn("ommCreateBaker")(C.byref(d), C.byref(out))
        assert r == SUCCESS, r
        return out

    def destroy_baker(self, baker):
        return self.fn("ommDestroyBaker")(baker)

    def set_knob(self, baker, knob, value):
        """ommxSetBakerKnob (include/omm_mi355x_ext.h): product library only"""
        self.dll.ommxSetBakerKnob.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        r = self.dll.ommxSetBakerKnob(baker, knob, value)
        assert r == SUCCESS, r

    def create_texture(self, baker, mips, alpha_cutoff=-1.0, disable_zorder=False, expect=SUCCESS, row_pitch=0):
        """mips: list of 2-D float32 / uint8 arrays (mip 0 first)."""
        mips = [np.ascontiguousarray(m) for m in mips]
        fmt = TEX_FP32 if mips[0].dtype == np.float32 else TEX_UNORM8
        md = (TextureMipDesc * len(mips))()
        for i, m in enumerate(mips):
            md[i].width, md[i].height, md[i].rowPitch = m.shape[1], m.shape[0], row_pitch
            md[i].textureData = m.ctypes.data
        td = TextureDesc()
        td.format, td.flags, td.mips, td.mipCount, td.alphaCutoff = fmt, (TEXFLAG_DISABLE_ZORDER if disable_zorder else 0), md, len(mips), alpha_cutoff
        out = C.c_void_p()
        r = self.fn("ommCpuCreateTexture")(baker, C.byref(td), C.byref(out))
        assert r == expect, (r, expect)
        return out if r == SUCCESS else None

    def destroy_texture(self, baker, tex):
        return self.fn("ommCpuDestroyTexture")(baker, tex)

    def bake_raw(self, baker, desc):
        out = C.c_void_p()
        r = self.fn("ommCpuBake")(baker, C.byref(desc), C.byref(out))
        return r, out

    def bake(self, baker, desc, expect=SUCCESS, want_stats=True):
        r, out = self.bake_raw(baker, desc)
        assert r == expect, "ommCpuBake returned %d, expected %d" % (r, expect)
        if r != SUCCESS:
            assert not out.value, "outBakeResult must stay untouched on failure"
            return None
        pd = C.POINTER(BakeResultDesc)()
        assert self.fn("ommCpuGetBakeResultDesc")(out, C.byref(pd)) == SUCCESS
        res = BakeResult(pd.contents)
        if want_stats:
            st = DebugStats()
            assert self.fn("ommDebugGetStats")(baker, pd, C.byref(st)) == SUCCESS
            res.stats = st
            st2 = DebugStats()
            assert self.fn("ommDebugGetStats2")(baker, out, C.byref(st2)) == SUCCESS
            res.stats2 = st2
            assert (st2.totalOpaque, st2.totalTransparent, st2.totalFullyOpaque, st2.totalFullyUnknownTransparent) == \
                   (st.totalOpaque, st.totalTransparent, st.totalFullyOpaque, st.totalFullyUnknownTransparent)
        assert self.fn("ommCpuDestroyBakeResult")(out) == SUCCESS
        return res


class Hip:
    """Minimal ctypes view of the HIP runtime: device buffers for the device-resident entry point (no torch needed)."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(arr.nbytes, 16)) == 0
        assert self.rt.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p

    def download(self, ptr, nbytes, dtype=np.uint8):
        out = np.empty(nbytes, np.uint8)
        if nbytes:
            assert self.rt.hipMemcpy(out.ctypes.data, ptr, nbytes, 2) == 0
        return out.view(dtype)

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(int(nbytes), 16)) == 0
        return p

    def copy_dtod(self, dst, src, nbytes):
        if nbytes:
            assert self.rt.hipMemcpy(dst, src, nbytes, 3) == 0
            # a device-to-device hipMemcpy returns before the copy has finished, and the library's streams are non-blocking (they do
            # not order against the null stream): wait, as a real caller's collective does before it hands the buffer over
            assert self.rt.hipDeviceSynchronize() == 0

    def copy_htod(self, dst, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            assert self.rt.hipMemcpy(dst, arr.ctypes.data, arr.nbytes, 1) == 0

    def free(self, p):
        self.rt.hipFree(p)


def device_result_to_host(lib, hip, out):
    """host copy (BakeResult) of an ommxDeviceBakeResult; destroys the device result"""
    lib.dll.ommxGetDeviceBakeResultDesc.argtypes = [C.c_void_p, C.POINTER(C.POINTER(BakeResultDesc))]
    lib.dll.ommxDestroyDeviceBakeResult.argtypes = [C.c_void_p]
    pd = C.POINTER(BakeResultDesc)()
    assert lib.dll.ommxGetDeviceBakeResultDesc(out, C.byref(pd)) == SUCCESS
    dev = pd.contents
    isz = {IDX_U8: 1, IDX_U16: 2, IDX_U32: 4}[dev.indexFormat]
    host_arrays = [hip.download(dev.arrayData, dev.arrayDataSize), hip.download(dev.descArray, 8 * dev.descArrayCount),
                   hip.download(dev.indexBuffer, isz * dev.indexCount)]
    hd = BakeResultDesc.from_buffer_copy(dev)
    hd.arrayData = host_arrays[0].ctypes.data
    hd.descArray = C.cast(host_arrays[1].ctypes.data, C.POINTER(MicromapDesc))
    hd.indexBuffer = host_arrays[2].ctypes.data
    res = BakeResult(hd)
    assert lib.dll.ommxDestroyDeviceBakeResult(out) == SUCCESS
    return res


def bake_sharded_simulated(lib, hip, baker, desc, uv, ix, world, levels=None):
    """All `world` ranks of a sharded bake inside ONE process on one GPU; the two collectives are emulated with host arithmetic
    (sum of the metadata words, concatenation of the padded contributions).  Returns one BakeResult per rank."""
    import omm_amd.sharded as sh
    dll = sh.bind(lib.dll)
    d_uv, d_ix = hip.upload(uv), hip.upload(ix)
    d_lv = hip.upload(np.ascontiguousarray(levels, dtype=np.uint8)) if levels is not None else None
    dd = BakeInputDesc.from_buffer_copy(desc)
    dd.texCoords, dd.indexBuffer, dd.subdivisionLevels = d_uv, d_ix, d_lv
    handles = []
    for r in range(world):
        h = C.c_void_p()
        rc = dll.ommxShardedBegin(baker, C.byref(dd), r, world, C.byref(h))
        assert rc == SUCCESS, rc
        handles.append(h)
    # all-reduce(SUM) of the metadata words
    metas = []
    for h in handles:
        w, n = C.c_void_p(), C.c_uint64()
        assert dll.ommxShardedGetMeta(h, C.byref(w), C.byref(n)) == SUCCESS
        metas.append((w, n.value))
    if metas[0][1]:
        total = np.zeros(metas[0][1], np.uint32)
        for w, n in metas:
            total += hip.download(w, 4 * n, np.uint32)
        for w, n in metas:
            hip.copy_htod(w, total)
    # tail + all-gather of the padded contributions
    contribs = []
    for h in handles:
        c, nb, st = C.c_void_p(), C.c_uint64(), C.c_uint64()
        rc = dll.ommxShardedTail(h, C.byref(c), C.byref(nb), C.byref(st))
        assert rc == SUCCESS, rc
        contribs.append((c, nb.value, st.value))
    stride = contribs[0][2]
    assert all(c[2] == stride for c in contribs)
    gathered = hip.alloc(stride * world)
    for r, (c, nb, st) in enumerate(contribs):
        hip.copy_dtod(C.c_void_p(gathered.value + r * stride), c, stride)
    results = []
    for h in handles:
        out = C.c_void_p()
        rc = dll.ommxShardedFinish(h, gathered, C.byref(out))
        assert rc == SUCCESS, rc
        results.append(device_result_to_host(lib, hip, out))
        assert dll.ommxShardedDestroy(h) == SUCCESS
    for p in (d_uv, d_ix, d_lv, gathered):
        if p is not None:
            hip.free(p)
    return results


def bake_device(lib, hip, baker, desc, uv, ix, levels=None):
    """ommxBakeDevice (include/omm_mi355x_ext.h): same desc, bulk arrays in HBM; returns a host copy as BakeResult."""
    lib.dll.ommxBakeDevice.argtypes = [C.c_void_p, C.POINTER(BakeInputDesc), C.POINTER(C.c_void_p)]
    lib.dll.ommxGetDeviceBakeResultDesc.argtypes = [C.c_void_p, C.POINTER(C.POINTER(BakeResultDesc))]
    lib.dll.ommxDestroyDeviceBakeResult.argtypes = [C.c_void_p]
    d_uv, d_ix = hip.upload(uv), hip.upload(ix)
    d_lv = hip.upload(np.ascontiguousarray(levels, dtype=np.uint8)) if levels is not None else None
    dd = BakeInputDesc.from_buffer_copy(desc)
    dd.texCoords, dd.indexBuffer = d_uv, d_ix
    dd.subdivisionLevels = d_lv
    out = C.c_void_p()
    r = lib.dll.ommxBakeDevice(baker, C.byref(dd), C.byref(out))
    assert r == SUCCESS, r
    pd = C.POINTER(BakeResultDesc)()
    assert lib.dll.ommxGetDeviceBakeResultDesc(out, C.byref(pd)) == SUCCESS
    dev = pd.contents
    isz = {IDX_U8: 1, IDX_U16: 2, IDX_U32: 4}[dev.indexFormat]
    host_arrays = [hip.download(dev.arrayData, dev.arrayDataSize), hip.download(dev.descArray, 8 * dev.descArrayCount),
                   hip.download(dev.indexBuffer, isz * dev.indexCount)]
    hd = BakeResultDesc.from_buffer_copy(dev)
    hd.arrayData = host_arrays[0].ctypes.data
    hd.descArray = C.cast(host_arrays[1].ctypes.data, C.POINTER(MicromapDesc))
    hd.indexBuffer = host_arrays[2].ctypes.data
    res = BakeResult(hd)
    assert lib.dll.ommxDestroyDeviceBakeResult(out) == SUCCESS
    for p in (d_uv, d_ix, d_lv):
        if p is not None:
            hip.free(p)
    return res


def make_desc(tex, tex_coords, indices, level, *, alpha_cutoff=0.5, fmt=FMT_4STATE, addr=CLAMP, filt=LINEAR,
              promo=PROMO_NEAREST, flags=FLAG_THREADS, le=T, gt=O, dyn_scale=0.0, unresolved=SPECIAL_FUO,
              uv_format=UV32_FLOAT, max_workload=0xFFFFFFFFFFFFFFFF, levels=None, border_alpha=0.0,
              rejection=0.0, keep=None):
    """Bake desc with the reference KAT defaults (test_omm_bake_cpu.cpp:181-207).  `keep` collects the numpy buffers
    the desc points into so they outlive the call."""
    d = default_bake_desc()
    tc = np.ascontiguousarray(tex_coords)
    ix = np.ascontiguousarray(indices)
    d.texture = tex
    d.format = fmt
    d.alphaMode = 0
    d.runtimeSamplerDesc = SamplerDesc(addr, filt, border_alpha)
    d.indexFormat = {np.dtype(np.uint32): IDX_U32, np.dtype(np.uint16): IDX_U16, np.dtype(np.uint8): IDX_U8}[ix.dtype]
    d.indexBuffer = ix.ctypes.data
    d.texCoords = tc.ctypes.data
    d.texCoordFormat = uv_format
    d.indexCount = ix.size
    d.maxSubdivisionLevel = level
    d.alphaCutoff = alpha_cutoff
    d.alphaCutoffLessEqual = le
    d.alphaCutoffGreater = gt
    d.unknownStatePromotion = promo
    d.bakeFlags = flags
    d.maxWorkloadSize = max_workload
    d.unresolvedTriState = unresolved
    d.dynamicSubdivisionScale = dyn_scale
    d.rejectionThreshold = rejection
    bufs = [tc, ix]
    if levels is not None:
        lv = np.ascontiguousarray(levels, dtype=np.uint8)
        d.subdivisionLevels = lv.ctypes.data
        bufs.append(lv)
    d._bufs = bufs
    if keep is not None:
        keep.extend(bufs)
    return d


# ---- seeded synthetic workloads shared by tests and bench.py (no std::uniform_real_distribution: counter hash) ----
def hash_u32(x):
    x = np.asarray(x, dtype=np.uint64)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7feb352d) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846ca68b) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(16))
    return x.astype(np.uint32)


def uniform01(seed, n, stream):
    idx = np.arange(n, dtype=np.uint64)
    h = hash_u32(idx * np.uint64(0x9E3779B1) + np.uint64(seed) * np.uint64(0x85EBCA6B) + np.uint64(stream) * np.uint64(0xC2B2AE35))
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def random_triangles(seed, n, extent, lo=0.0, hi=1.0):
    """n triangles: centre ~ U[lo,hi)^2, vertices = centre + U(-extent/2, extent/2)^2.  Unshared vertices."""
    cx = uniform01(seed, n, 0) * np.float32(hi - lo) + np.float32(lo)
    cy = uniform01(seed, n, 1) * np.float32(hi - lo) + np.float32(lo)
    uv = np.empty((n, 3, 2), np.float32)
    for v in range(3):
        uv[:, v, 0] = cx + (uniform01(seed, n, 2 + 2 * v) - np.float32(0.5)) * np.float32(extent)
        uv[:, v, 1] = cy + (uniform01(seed, n, 3 + 2 * v) - np.float32(0.5)) * np.float32(extent)
    idx = np.arange(3 * n, dtype=np.uint32)
    return uv.reshape(-1, 2), idx


def value_noise(seed, w, h, octaves=4, base_cell=64):
    """Multi-octave value noise on an integer-hash lattice -> float32 (h, w) in [0,1] (tests/native/kat_textures.c)."""
    out = np.empty((h, w), np.float32)
    kat_lib().wl_noise_f32(seed, w, h, octaves, base_cell, out.ctypes.data)
    return out


def foliage_texture(seed, w, h, feature=64):
    """'Foliage-style' alpha: thresholded low-frequency noise blobs with a soft edge -> uint8 (h, w)."""
    out = np.empty((h, w), np.uint8)
    kat_lib().wl_foliage_u8(seed, w, h, feature, out.ctypes.data)
    return out
