"""Which phase of ommCpuBake is long in the slow bakes?  N bakes of the metric configuration through the C ABI, per-bake phase clocks (ommxBakeTimings).
usage (GPU box): python profiles/scripts/r06_outliers.py [N]"""
import os, sys, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ommtest as ot, workloads as wl, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tex, uv, ix, lv, kw = wl.workload("c2", 1000000)
prod = ot.Lib("product"); b = prod.create_baker(); t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
d = bench.desc_for(t, uv, ix, lv, kw)
def cpu_stat():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}
st0 = cpu_stat()
rows = []
for i in range(n + 3):
    t0 = time.perf_counter(); r, out = prod.bake_raw(b, d); wall = (time.perf_counter() - t0) * 1e3
    tm = bench.get_timings(prod, b); prod.fn("ommCpuDestroyBakeResult")(out)
    if i >= 3:
        dev = tm.setupMs + tm.triageMs + tm.classifyMs + tm.digestMs + tm.tailMs + tm.gatherMs
        rows.append((wall, tm.uploadMs, dev, tm.compressMs, tm.expandMs, tm.totalMs - tm.uploadMs - dev - tm.compressMs - tm.expandMs))
st1 = cpu_stat()
print("cgroup cpu.stat over the run:", {k: st1[k] - st0.get(k, 0) for k in st1 if k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec")})
a = np.array(rows); med = np.median(a, axis=0)
print("median: wall %.2f = upload %.2f + device %.2f + codec %.2f + expand %.2f + rest %.2f" % tuple(med))
slow = a[a[:, 0] > med[0] + 1.0]
print("%d of %d bakes more than 1 ms above the median; their excess by phase (ms):" % (len(slow), n))
for row in slow[:12]:
    print("  wall %+.2f: upload %+.2f device %+.2f codec %+.2f expand %+.2f rest %+.2f" % tuple(row - med))
