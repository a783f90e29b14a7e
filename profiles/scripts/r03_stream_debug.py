"""does a forced streamed ommCpuBake of a workload stream or fall back?  usage: r03_stream_debug.py <config> <tris> <ranges> [level override]"""
import sys, os, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, ommtest as ot, workloads as wl
cfg, tris, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = ot.Lib("product")
msgs = []
b = lib.create_baker(callback=lambda s, m, u: msgs.append(m.decode()))
tex_a, uv, ix, lv, kw = wl.workload(cfg, tris)
kw = dict(kw); lvl = kw.pop("level")
if len(sys.argv) > 4:
    mode = sys.argv[4]
    if mode == "uniform": lv = None
    elif mode == "hi": lv = np.where(lv == 0xF, 9, np.maximum(lv, 6)).astype(np.uint8)
    elif mode == "nodyn": lv = np.where(lv == 0xF, 7, lv).astype(np.uint8)
t = lib.create_texture(b, [tex_a], alpha_cutoff=0.5)
d = ot.make_desc(t, uv, ix, lvl, flags=ot.FLAG_THREADS | (1 << 5), levels=lv, **kw)
lib.set_knob(b, ot.KNOB_STREAM_CHUNKS, K)
r, out = lib.bake_raw(b, d); print(cfg, tris, K, "->", r); lib.fn("ommCpuDestroyBakeResult")(out)
for m in msgs: print("MSG:", m)
