"""the metric configuration through ommxShardedBakeRccl with a ONE-rank communicator: what the block exchange codec costs (compress, expand, scatter) and how much it
would put on the wire per rank; compared byte for byte with ommxBakeDevice.  usage: r03_rccl_one_rank_c2.py [config]"""
import os, sys, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
import ommtest as ot, omm_amd.sharded as sh, workloads as wl, bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1); torch.cuda.set_device(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
prod = ot.Lib("product"); b = prod.create_baker()
tex, uv, ix, lv, kw = wl.workload(cfg); kw = dict(kw); lvl = kw.pop("level")
t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
d = ot.make_desc(t, uv, ix, lvl, levels=lv, **kw)
hip = ot.Hip()
ref = ot.bake_device(prod, hip, b, d, uv, ix, levels=lv)
duv = torch.from_numpy(uv).cuda(); dix = torch.from_numpy(ix.astype(np.int32)).cuda()
dd = ot.BakeInputDesc.from_buffer_copy(d); dd.texCoords, dd.indexBuffer = duv.data_ptr(), dix.data_ptr()
if lv is not None:
    dlv = torch.from_numpy(lv).cuda(); dd.subdivisionLevels = dlv.data_ptr()
comm = sh.rccl_comm(prod.dll, torch, dist, 0, 1)
for i in range(3):
    t0 = time.perf_counter(); out = sh.sharded_bake_rccl(prod.dll, b, C.byref(dd), comm); dt = time.perf_counter() - t0
    tm = bench.get_timings(prod, b)
    print("bake %d: %.2f ms wall; classify %.2f tail %.2f exchange+scatter %.2f ms; contribution %.1f MB -> %.2f MB on the wire (%.1f %%)" %
          (i, dt * 1e3, tm.classifyMs, tm.tailMs, tm.gatherMs, tm.contributionBytes / 1e6, tm.exchangeBytes / 1e6, 100.0 * tm.exchangeBytes / max(1, tm.contributionBytes)))
    res = ot.device_result_to_host(prod, hip, out) if i == 2 else None
    if res is None: prod.dll.ommxDestroyDeviceBakeResult.argtypes = [C.c_void_p]; prod.dll.ommxDestroyDeviceBakeResult(out)
assert res.same_as(ref), res.diff(ref)
print("identical to ommxBakeDevice")
prod.dll.ommxRcclCommDestroy(comm); dist.destroy_process_group()
