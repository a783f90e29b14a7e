"""How many work items of the metric workload share their coarse classification with another item, by preview level?  (The streamed ommCpuBake classifies such
'families' early; round 4 asks whether a second preview stage at level 6 would shrink the early class.)  Bakes the workload at levels 4..7 with dedup and
special indices off and counts the non-uniform blocks that occur more than once."""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ommtest as ot, workloads as wl
tex, uv, ix, lv, kw = wl.workload("c2", int(sys.argv[1]) if len(sys.argv) > 1 else 1000000)
prod = ot.Lib("product"); b = prod.create_baker(); t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
k = dict(kw); k.pop("level")
ref = None
for level in (4, 5, 6, 7):
    r = prod.bake(b, ot.make_desc(t, uv, ix, level, flags=ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL | ot.FLAG_NO_DEDUP, **k), want_stats=False)
    n = len(r.descs); size = max(1, (4 ** level) // 4)
    order = np.argsort(r.descs[:, 0]); idx = np.asarray(r.index)
    blocks = r.array_data.reshape(n, size)
    uniform = (blocks == blocks[:, :1]).all(1) & np.isin(blocks[:, 0], [0x00, 0x55, 0xAA, 0xFF])
    # per triangle signature
    h = np.zeros(n, np.uint64)
    v = blocks.view(np.uint8).astype(np.uint64)
    mult = np.uint64(0x9E3779B97F4A7C15)
    for c in range(0, size, 1):
        h = (h ^ v[:, c]) * mult; h ^= h >> np.uint64(29)
    sig = h[idx]                      # descriptor of every triangle (no dedup: one per triangle)
    mixed = ~uniform[idx]
    s = sig[mixed]
    uniq, cnt = np.unique(s, return_counts=True)
    shared = cnt[cnt > 1].sum()
    print("level %d: triangles with a non-uniform coarse block %d, of which share it with another triangle %d (%d families)" % (level, mixed.sum(), shared, (cnt > 1).sum()), flush=True)
