"""Two REAL processes (torch.multiprocessing), both on GPU 0, run the sharded bake of omm_amd/sharded.py with torch.distributed
collectives on device tensors (backend gloo: RCCL refuses two ranks on one GPU).  Every rank must end with the single-GPU result.
usage: python tests/scripts/two_rank_gloo_gpu.py [world]"""
import os, sys, ctypes as C, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def worker(rank, world, port, outdir):
    import torch, torch.distributed as dist
    import ommtest as ot, omm_amd.sharded as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prod = ot.Lib("product"); b = prod.create_baker()
    tex = ot.foliage_texture(5, 1024, 1024, feature=48); t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
    n = 6000
    uv, ix = ot.random_triangles(808, n, 0.02)
    lv = (5 + ot.hash_u32(np.arange(n) + 3) % 3).astype(np.uint8)          # levels 5..7: three level groups to partition
    d = ot.make_desc(t, uv, ix, 7, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=lv)
    ref = prod.bake(b, d)
    duv = torch.from_numpy(uv).cuda(); dix = torch.from_numpy(ix.astype(np.int32)).cuda(); dlv = torch.from_numpy(lv).cuda()
    dd = ot.BakeInputDesc.from_buffer_copy(d); dd.texCoords, dd.indexBuffer, dd.subdivisionLevels = duv.data_ptr(), dix.data_ptr(), dlv.data_ptr()
    ok = True
    for _ in range(2):
        out = sh.sharded_bake(prod.dll, b, C.byref(dd), rank, world, torch, dist)
        res = ot.device_result_to_host(prod, ot.Hip(), out)
        ok = ok and res.same_as(ref)
        if not ok:
            print("rank", rank, res.diff(ref), flush=True)
    open(os.path.join(outdir, "rank%d" % rank), "w").write("%d %d" % (ok, len(ref.descs)))
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile, torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(worker, args=(world, port, td), nprocs=world, join=True)
        res = [open(os.path.join(td, "rank%d" % r)).read().split() for r in range(world)]
    assert all(r[0] == "1" for r in res), res
    print("two-process sharded bake ok:", res)
