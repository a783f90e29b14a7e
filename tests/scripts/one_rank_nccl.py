# exercises the torch.distributed plumbing (device_tensor over raw pointers, nccl all_reduce / all_gather) with world_size 1
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.distributed as dist
import ommtest as ot, omm_amd.sharded as sh
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
a = torch.arange(16, dtype=torch.int32, device="cuda")
v = sh.device_tensor(torch, a.data_ptr(), 16, torch.int32); v += 1
assert int(a.sum()) == 136, a
dist.all_reduce(v); torch.cuda.synchronize()
g = torch.empty(16, dtype=torch.uint8, device="cuda"); dist.all_gather_into_tensor(g, torch.arange(16, dtype=torch.uint8, device="cuda"))
prod = ot.Lib("product"); b = prod.create_baker()
tex = ot.foliage_texture(5, 1024, 1024, feature=48); t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
uv, ix = ot.random_triangles(808, 5000, 0.02)
d = ot.make_desc(t, uv, ix, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
ref = prod.bake(b, d)
duv = torch.from_numpy(uv).cuda(); dix = torch.from_numpy(ix.astype(np.int32)).cuda()
dd = ot.BakeInputDesc.from_buffer_copy(d); dd.texCoords, dd.indexBuffer = duv.data_ptr(), dix.data_ptr()
out = sh.sharded_bake(prod.dll, b, C.byref(dd), 0, 1, torch, dist)
res = ot.device_result_to_host(prod, ot.Hip(), out)
assert res.same_as(ref), res.diff(ref)
# the native path next to torch's own RCCL process group: communicator bootstrap through torch.distributed, collectives inside the library
comm = sh.rccl_comm(prod.dll, torch, dist, 0, 1)
out2 = sh.sharded_bake_rccl(prod.dll, b, C.byref(dd), comm)
res2 = ot.device_result_to_host(prod, ot.Hip(), out2)
assert res2.same_as(ref), res2.diff(ref)
# ... and with the opt-in size budget: states merged by ncclAllReduce, serial tail on the host, result uploaded (== ommCpuBake with the same budget)
d3 = ot.BakeInputDesc.from_buffer_copy(d); d3.maxArrayDataSize = 300000
ref3 = prod.bake(b, d3)
dd3 = ot.BakeInputDesc.from_buffer_copy(dd); dd3.maxArrayDataSize = 300000
out3 = sh.sharded_bake_rccl(prod.dll, b, C.byref(dd3), comm)
res3 = ot.device_result_to_host(prod, ot.Hip(), out3)
assert res3.same_as(ref3) and res3.array_data.size <= 300000 < ref.array_data.size, (res3.diff(ref3), res3.array_data.size, ref.array_data.size)
# the block exchange codec (tail_kernels.hip): the contributions cross the links as unit codes + raw units.  A level-8 bake of the metric workload's kind
# shrinks to a few per cent and must arrive bit for bit; blocks of noise do not shrink below half and travel as they are
import bench
def timings():
    return bench.get_timings(prod, b)
res2b = ot.device_result_to_host(prod, ot.Hip(), sh.sharded_bake_rccl(prod.dll, b, C.byref(dd), comm))
tm2 = timings()
assert res2b.same_as(ref)
assert 0 < tm2.exchangeBytes < tm2.contributionBytes // 2, (tm2.exchangeBytes, tm2.contributionBytes)
uv8, ix8 = ot.random_triangles(811, 30000, 8.0 / 1024)
d8 = ot.make_desc(t, uv8, ix8, 8, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
ref8 = prod.bake(b, d8, want_stats=False)
duv8 = torch.from_numpy(uv8).cuda(); dix8 = torch.from_numpy(ix8.astype(np.int32)).cuda()
dd8 = ot.BakeInputDesc.from_buffer_copy(d8); dd8.texCoords, dd8.indexBuffer = duv8.data_ptr(), dix8.data_ptr()
res8 = ot.device_result_to_host(prod, ot.Hip(), sh.sharded_bake_rccl(prod.dll, b, C.byref(dd8), comm))
tm8 = timings()
assert res8.same_as(ref8), res8.diff(ref8)
assert ref8.array_data.size > (20 << 20) and tm8.exchangeBytes * 8 < tm8.contributionBytes, (ref8.array_data.size, tm8.exchangeBytes, tm8.contributionBytes)
noise = (np.random.RandomState(5).rand(512, 512) * 255).astype(np.uint8)
tn = prod.create_texture(b, [noise], alpha_cutoff=0.5)
uvn, ixn = ot.random_triangles(812, 400, 0.3)
dn = ot.make_desc(tn, uvn, ixn, 6, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE)
refn = prod.bake(b, dn, want_stats=False)
duvn = torch.from_numpy(uvn).cuda(); dixn = torch.from_numpy(ixn.astype(np.int32)).cuda()
ddn = ot.BakeInputDesc.from_buffer_copy(dn); ddn.texCoords, ddn.indexBuffer = duvn.data_ptr(), dixn.data_ptr()
resn = ot.device_result_to_host(prod, ot.Hip(), sh.sharded_bake_rccl(prod.dll, b, C.byref(ddn), comm))
tmn = timings()
assert resn.same_as(refn), resn.diff(refn)
assert tmn.exchangeBytes == tmn.contributionBytes > 0, (tmn.exchangeBytes, tmn.contributionBytes)      # (sent raw: the chunked all-gather of round 2)
print("codec: %d -> %d bytes (foliage, level 8), %d -> %d (noise)" % (tm8.contributionBytes, tm8.exchangeBytes, tmn.contributionBytes, tmn.exchangeBytes))
prod.dll.ommxRcclCommDestroy(comm)
print("one-rank nccl plumbing ok", len(res.descs))
dist.destroy_process_group()
