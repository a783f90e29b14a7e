"""Would a key made of the EXACT final states of K sampled micro-triangles per work item be as sharp a duplicate preview as the level-5 classification
(41 425 'early' items of ~125 k at the metric configuration) -- at a fraction of its 2.7 ms?  Bakes the workload once at level 8 without duplicate
detection (one block per non-uniform item) and counts, for K evenly spaced bird-curve indices, the items whose key is shared with another item."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ommtest as ot, workloads as wl
tex, uv, ix, lv, kw = wl.workload("c2", int(sys.argv[1]) if len(sys.argv) > 1 else 1000000)
prod = ot.Lib("product"); b = prod.create_baker(); t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
k = dict(kw); level = k.pop("level")
r = prod.bake(b, ot.make_desc(t, uv, ix, level, flags=ot.FLAG_THREADS | ot.FLAG_NO_DEDUP, **k), want_stats=False)
n = len(r.descs); size = (4 ** level) // 4
blocks = np.asarray(r.array_data).reshape(n, size)
print("non-uniform items (blocks):", n, flush=True)
def count_shared(keys):
    _, inv, cnt = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
    return int((cnt[inv.reshape(-1)] > 1).sum()), int((cnt > 1).sum())
full = np.zeros(n, np.uint64); mult = np.uint64(0x9E3779B97F4A7C15)
v = blocks.view(np.uint64)
for c in range(v.shape[1]):
    full = (full ^ v[:, c]) * mult; full ^= full >> np.uint64(29)
s, f = count_shared(full.reshape(-1, 1)); print("true duplicates (whole block equal): %d items in %d families" % (s, f), flush=True)
total = 4 ** level
for K in (32, 64, 128, 256, 512, 1024, 4096):
    idx = (np.arange(K, dtype=np.int64) * (total // K)) + (total // K) // 2
    st = (blocks[:, idx // 4] >> ((idx % 4) * 2).astype(np.uint8)) & 3
    key = np.zeros(n, np.uint64)
    for c in range(K):
        key = (key ^ st[:, c].astype(np.uint64)) * mult; key ^= key >> np.uint64(31)
    s, f = count_shared(key.reshape(-1, 1))
    print("K = %4d sampled micro-triangles: %d items share their key with another item (%d families)" % (K, s, f), flush=True)
