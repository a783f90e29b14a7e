"""One-off consistency check at FULL bench size (1 M triangles, 4K alpha, level 8): the sharded path with WORLD simulated ranks on one GPU
must reproduce the single-GPU result byte for byte (1.27 GB arrayData, descs, indices) on every rank.
usage: python tests/scripts/full_size_sharded_check.py [world]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import ommtest as ot

world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 1000000
tex = ot.foliage_texture(1234, 4096, 4096, feature=64)
uv, ix = ot.random_triangles(1235, n, 8.0 / 4096)
prod = ot.Lib("product"); hip = ot.Hip()
b = prod.create_baker(); t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
d = ot.make_desc(t, uv, ix, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
ref = prod.bake(b, d, want_stats=False)
h = lambda r: hashlib.sha256(r.array_data.tobytes() + r.desc_bytes + r.index.tobytes()).hexdigest()[:16]
print("single GPU: %d descs, %.2f GB, sha %s" % (len(ref.descs), ref.array_data.size / 1e9, h(ref)))
t0 = time.time()
res = ot.bake_sharded_simulated(prod, hip, b, d, uv, ix.astype(np.int32), world)
print("sharded x%d: %.1f s" % (world, time.time() - t0))
for r, x in enumerate(res):
    ok = x.same_as(ref)
    print(" rank %d: %s sha %s" % (r, "identical" if ok else "DIFFERENT " + x.diff(ref), h(x)))
    assert ok
print("ok")
