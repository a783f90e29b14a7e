"""how compressible are the OMM blocks of a bake for the multi-GPU block exchange?  counts the 16-byte units of arrayData (= 64 micro-triangles in 4-state) that hold one
repeated state pattern (0x00 / 0x55 / 0xAA / 0xFF).  usage: r03_block_compressibility.py <config> [tris]"""
import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, ommtest as ot, workloads as wl
cfg = sys.argv[1]; tris = int(sys.argv[2]) if len(sys.argv) > 2 else None
lib = ot.Lib("product")
b = lib.create_baker()
tex, uv, ix, lv, kw = wl.workload(cfg, tris)
kw = dict(kw); lvl = kw.pop("level")
t = lib.create_texture(b, [tex], alpha_cutoff=0.5)
res = lib.bake(b, ot.make_desc(t, uv, ix, lvl, levels=lv, **kw), want_stats=False)
a = res.array_data
n = a.size // 16 * 16
u = a[:n].reshape(-1, 16)
first = u[:, :1]
uniform = (u == first).all(axis=1) & np.isin(first[:, 0], [0x00, 0x55, 0xAA, 0xFF])
raw = int((~uniform).sum())
comp = u.shape[0] * 0.5 + raw * 16          # 4-bit code per unit + raw units
print("%s: arrayData %.1f MB, %d units of 16 B, uniform %.1f %%, compressed (4-bit code per unit + raw units) %.1f MB = %.1f %% -> x%.1f" %
      (cfg, a.size / 1e6, u.shape[0], 100.0 * uniform.mean(), comp / 1e6, 100.0 * comp / a.size, a.size / comp))
