"""Several REAL processes, all on GPU 0, run the ONE-CALL sharded bake (ommxShardedBakeRccl) over a communicator made of torch.distributed
collectives (ommxCommFromCollectives; backend gloo: RCCL refuses two ranks on one GPU).  Everything the library does between its collectives
at world_size > 1 -- status agreement, metadata merge, codec streams at their rank offsets, decoding into the result, the raw chunked exchange,
the host-tail route, ranks without a share -- runs exactly as under RCCL; only the two transport calls differ.
usage: python tests/scripts/ranks_one_call_gloo_gpu.py [world] [full]     (full: the metric workload at its full size only -- 1 M triangles, 1.27 GB of blocks)
       python tests/scripts/ranks_one_call_gloo_gpu.py world fuzz LO HI  (the randomized cases LO..HI-1 of test_gpu_parity._fuzz_case -- formats, filters, address modes, mip chains,
                                                                          per-triangle levels, flags -- sharded, against the single-GPU ommCpuBake of the same library)"""
import os, sys, ctypes as C, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def worker(rank, world, port, outdir, full):
    import torch, torch.distributed as dist
    import ommtest as ot, omm_amd.sharded as sh, bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prod = ot.Lib("product"); b = prod.create_baker(); hip = ot.Hip()
    comm = sh.CollectivesComm(prod.dll, torch, dist, rank, world)
    log = []

    def case(name, tex, uv, ix, level, levels=None, budget=None, flags=None, chunk=0, expect=None):
        t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
        kw = {} if flags is None else {"flags": flags}
        d = ot.make_desc(t, uv, ix, level, addr=ot.WRAP, promo=ot.PROMO_FORCE_OPAQUE, levels=levels, **kw)
        if budget is not None:
            d.maxArrayDataSize = budget
        ref = prod.bake(b, d, want_stats=False)
        keep = [torch.from_numpy(uv).cuda(), torch.from_numpy(ix.astype(np.int32)).cuda()]
        dd = ot.BakeInputDesc.from_buffer_copy(d); dd.texCoords, dd.indexBuffer = keep[0].data_ptr(), keep[1].data_ptr()
        if levels is not None:
            keep.append(torch.from_numpy(levels).cuda()); dd.subdivisionLevels = keep[2].data_ptr()
        prod.set_knob(b, ot.KNOB_SHARD_CHUNK_BYTES, chunk)
        ok = True
        for _ in range(2):   # (the second bake re-uses the pooled working set)
            res = ot.device_result_to_host(prod, hip, sh.sharded_bake_rccl(prod.dll, b, C.byref(dd), comm.handle))
            same = res.same_as(ref)
            if not same:
                print("rank", rank, name, res.diff(ref), flush=True)
            ok = ok and same
        tm = bench.get_timings(prod, b)
        if expect == "codec":
            ok = ok and 0 < tm.exchangeBytes < tm.contributionBytes // 2
        elif expect == "raw":
            ok = ok and tm.exchangeBytes == tm.contributionBytes > 0
        log.append("%s:%d:%d/%d:%d" % (name, ok, tm.exchangeBytes, tm.contributionBytes, len(ref.descs)))
        prod.set_knob(b, ot.KNOB_SHARD_CHUNK_BYTES, 0)
        prod.destroy_texture(b, t)
        return ok

    if isinstance(full, tuple):
        import test_gpu_parity as T
        bad = 0
        for seed in range(full[0], full[1]):
            mips, uv, ix, level, cutoff, sat, kw = T._fuzz_case(seed)
            t = prod.create_texture(b, mips, alpha_cutoff=cutoff if sat else -1.0)
            d = ot.make_desc(t, uv, ix, level, alpha_cutoff=cutoff, **kw)
            ref = prod.bake(b, d, want_stats=False)
            keep = [torch.from_numpy(uv).cuda(), torch.from_numpy(ix.astype(np.int32)).cuda()]
            dd = ot.BakeInputDesc.from_buffer_copy(d); dd.texCoords, dd.indexBuffer = keep[0].data_ptr(), keep[1].data_ptr()
            if kw.get("levels") is not None:
                keep.append(torch.from_numpy(kw["levels"]).cuda()); dd.subdivisionLevels = keep[2].data_ptr()
            prod.set_knob(b, ot.KNOB_SHARD_CHUNK_BYTES, (0, 256, 4096)[seed % 3])
            prod.set_knob(b, ot.KNOB_GENERIC_PASS, (seed // 3) % 3)      # (where the micro-triangles of several texels are classified: automatic / inline / deferred)
            res = ot.device_result_to_host(prod, hip, sh.sharded_bake_rccl(prod.dll, b, C.byref(dd), comm.handle))
            if not res.same_as(ref):
                bad += 1
                print("MISMATCH rank", rank, "seed", seed, res.diff(ref)[:300], flush=True)
            prod.destroy_texture(b, t)
        open(os.path.join(outdir, "rank%d" % rank), "w").write("%d\nseeds %d..%d: %d mismatches" % (bad == 0, full[0], full[1], bad))
        comm.destroy()
        dist.barrier(); dist.destroy_process_group()
        return
    if full:
        texf = ot.foliage_texture(1234, 4096, 4096, feature=64)
        uvf, ixf = ot.random_triangles(1235, 1000000, 8.0 / 4096)
        ok = case("metric workload, full size", texf, uvf, ixf, 8, expect="codec")
        open(os.path.join(outdir, "rank%d" % rank), "w").write("%d\n%s" % (ok, "\n".join(log)))
        comm.destroy()
        dist.barrier(); dist.destroy_process_group()
        return
    foliage = ot.foliage_texture(5, 1024, 1024, feature=48)
    noise = (np.random.RandomState(5).rand(512, 512) * 255).astype(np.uint8)
    n = 6000
    uv, ix = ot.random_triangles(808, n, 0.02)
    lv = (5 + ot.hash_u32(np.arange(n) + 3) % 3).astype(np.uint8)
    ok = case("levels 5..7, codec streams", foliage, uv, ix, 7, levels=lv, expect="codec")
    uv8, ix8 = ot.random_triangles(811, 8000, 8.0 / 1024)
    ok = case("level 8, codec streams", foliage, uv8, ix8, 8, expect="codec") and ok
    # few, large blocks (256 KiB each): the codec's count / scan scratch outgrows the bake's own scratch block, which is sized from the triangle count
    uv10, ix10 = ot.random_triangles(813, 300, 0.02)
    ok = case("level 10, few large blocks, codec streams", foliage, uv10, ix10, 10, expect="codec") and ok
    uvn, ixn = ot.random_triangles(812, 400, 0.3)
    ok = case("noise, raw exchange in many chunks", noise, uvn, ixn, 6, chunk=4096, expect="raw") and ok
    ok = case("noise, raw exchange in one chunk", noise, uvn, ixn, 6, expect="raw") and ok
    ok = case("size budget: states merged, serial tail on the host", foliage, uv[:6 * 1500], ix[:3 * 1500], 6, budget=200000) and ok
    ok = case("near-duplicate merge: the same route", foliage, uv[:6 * 600], ix[:3 * 600], 4, flags=ot.FLAG_THREADS | ot.FLAG_NEAR_DUP) and ok
    ok = case("fewer work items than ranks", foliage, uv[:6 * 2], ix[:3 * 2], 5) and ok
    ok = case("one work item", foliage, uv[:6], ix[:3], 6) and ok
    ok = case("no blocks at all", np.full((64, 64), 255, np.uint8), uv[:6 * 50], ix[:3 * 50], 4) and ok
    open(os.path.join(outdir, "rank%d" % rank), "w").write("%d\n%s" % (ok, "\n".join(log)))
    comm.destroy()
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile, torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        mode = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 and sys.argv[2] == "fuzz" else (len(sys.argv) > 2 and sys.argv[2] == "full")
        mp.spawn(worker, args=(world, port, td, mode), nprocs=world, join=True)
        res = [open(os.path.join(td, "rank%d" % r)).read().split("\n") for r in range(world)]
    for r in res:
        print(r)
    assert all(r[0] == "1" for r in res), res
    print("one-call sharded bake over caller collectives ok: world", world)
