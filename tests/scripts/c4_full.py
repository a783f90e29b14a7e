"""BASELINE configs[4] at FULL size on one GPU: 4 M triangles, per-triangle levels U{4..10} (75 %) / dynamic heuristic (25 %, scale 2, max 10),
8192^2 alpha, dedup on.  Prints timings and the result sizes; `python tests/scripts/c4_full.py [extent_texels]`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
import ommtest as ot

n = int(os.environ.get("C4_TRIS", "4000000")); ext = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
t0 = time.time()
tex = ot.foliage_texture(4321, 8192, 8192, feature=96)
uv, ix = ot.random_triangles(9, n, ext / 8192)
h = ot.hash_u32(np.arange(n) + 9000)
lv = (4 + (h >> 8) % 7).astype(np.uint8); lv[(h & 3) == 0] = 0xF
print("workload %.1f s" % (time.time() - t0))
lib = ot.Lib("product")
b = lib.create_baker()
t = lib.create_texture(b, [tex], alpha_cutoff=0.5)
d = ot.make_desc(t, uv, ix, 10, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv, dyn_scale=2.0)
for it in range(2):
    t1 = time.time()
    r, out = lib.bake_raw(b, d)
    dt = time.time() - t1
    print("bake %d: result %d in %.1f ms" % (it, r, dt * 1e3))
    if r != ot.SUCCESS:
        break
    pd = C.POINTER(ot.BakeResultDesc)()
    lib.fn("ommCpuGetBakeResultDesc")(out, C.byref(pd))
    rd = pd.contents
    print("  arrayData %.2f GB, %d descs, index format %d, hist %s" % (rd.arrayDataSize / 1e9, rd.descArrayCount, rd.indexFormat,
          [(rd.descArrayHistogram[k].subdivisionLevel, rd.descArrayHistogram[k].count) for k in range(rd.descArrayHistogramCount)]))
    lib.fn("ommCpuDestroyBakeResult")(out)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import bench
    tm = bench.get_timings(lib, b)   # (ommxGetLastBakeTimingsSized: the unsized symbol only fills the round-3 prefix)
    print("  micro-triangles %.3e, active %d / %d items, level-line %.3e; ms: setup %.1f triage %.1f classify %.1f digest %.1f tail %.1f gather %.1f d2h %.1f"
          % (tm.microTriangles, tm.activeItems, tm.uniqueItems, tm.fineMicroTriangles, tm.setupMs, tm.triageMs, tm.classifyMs, tm.digestMs, tm.tailMs, tm.gatherMs, tm.downloadMs))
