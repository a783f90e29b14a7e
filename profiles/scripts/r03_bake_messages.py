import sys, os, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, ommtest as ot, workloads as wl
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
lib = ot.Lib("product")
msgs = []
b = lib.create_baker(callback=lambda s, m, u: msgs.append(m.decode()))
tex_a, uv, ix, lv, kw = wl.workload(cfg)
kw = dict(kw)
t = lib.create_texture(b, [tex_a], alpha_cutoff=0.5)
lvl = kw.pop("level")
d = ot.make_desc(t, uv, ix, lvl, flags=ot.FLAG_THREADS | (1 << 5), levels=lv, **kw)
for i in range(2):
    t0 = time.perf_counter(); r, out = lib.bake_raw(b, d); dt = time.perf_counter() - t0
    print(cfg, "bake", i, r, "%.1f ms" % (dt * 1e3)); lib.fn("ommCpuDestroyBakeResult")(out)
for m in msgs: print("MSG:", m)
