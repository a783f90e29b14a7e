"""Worst cases of the two hash builds (not part of the suite): `python tests/scripts/hot_keys.py`
  A  500 000 copies of ONE triangle                      -> one hot key in the UV-dedup table
  B  500 000 different triangles with identical content  -> one hot key in the digest table (periodic texture, shifts by whole periods)
Prints the phase times; both results are checked against the obvious expectation (1 OMM block)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, ommtest as ot, bench
lib = ot.Lib("product"); b = lib.create_baker()
period = 64
tile = np.zeros((period, period), np.uint8); tile[:, period // 2:] = 255; tile[:, period // 2 - 2:period // 2 + 2] = np.array([40, 100, 160, 220], np.uint8)[None, :]   # a soft vertical edge
tex = np.tile(tile, (2048 // period, 2048 // period))
t = lib.create_texture(b, [tex], alpha_cutoff=0.5)
n = 500000
base = np.array([[0.0146, 0.0113], [0.0166, 0.0109], [0.0154, 0.0138]], np.float32)   # straddles the edge at u = 32 / 2048
def run(name, uv):
    ix = np.arange(3 * n, dtype=np.uint32)
    d = ot.make_desc(t, uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
    for it in range(2):
        t0 = time.time(); r = lib.bake(b, d, want_stats=False); dt = time.time() - t0
    tm = bench.get_timings(lib, b)
    print("%s: %d OMM blocks, %d unique items, bake %.1f ms; setup %.2f triage %.2f classify %.2f digest %.2f tail %.2f gather %.2f" %
          (name, len(r.descs), tm.uniqueItems, dt * 1e3, tm.setupMs, tm.triageMs, tm.classifyMs, tm.digestMs, tm.tailMs, tm.gatherMs))
    return r
uvA = np.tile(base, (n, 1)).astype(np.float32)
rA = run("A (one triangle x 500000)", uvA)
assert len(rA.descs) == 1
k = np.arange(n)
shift = np.stack([(k % 32) * (period / 2048.0), ((k // 32) % 32) * (period / 2048.0)], 1).astype(np.float32)   # 1024 distinct whole-period shifts
shift += np.stack([(k // 1024) * 1.0, np.zeros(n)], 1).astype(np.float32)                                      # ... times whole wraps: 500 000 distinct UVs
uvB = (base[None, :, :] + shift[:, None, :]).reshape(-1, 2).astype(np.float32)
rB = run("B (500000 shifted copies)", uvB)
assert len(rB.descs) <= 64, len(rB.descs)   # (nearly) identical content everywhere: a few dozen blocks, 500 000 references
