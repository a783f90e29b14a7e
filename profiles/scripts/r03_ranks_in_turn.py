"""The compute side of an N-GPU sharded bake, MEASURED on one GPU: N real processes run the one-call entry ommxShardedBakeRccl over a communicator of
caller collectives (ommxCommFromCollectives, gloo) whose callbacks make the ranks take TURNS -- between two collectives only one rank works on the GPU at a
time, the others wait in the callback; each rank clocks the time from getting its turn to reaching the next collective with its stream drained.  A rank of a
real N-GPU job spends exactly that (host work + kernels of the phase) on its own GPU, and every collective is a rendezvous, so

    step(N) = sum over the phases of  max over the ranks of the phase time   +   the wire time of the collectives (not measured here: DESIGN.md section 7)

usage: python profiles/scripts/r03_ranks_in_turn.py WORLD [config] [bakes]"""
import os, sys, time, ctypes as C, socket, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def worker(rank, world, port, outdir, cfg, bakes):
    import torch, torch.distributed as dist
    import ommtest as ot, omm_amd.sharded as sh, workloads as wl, bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prod = ot.Lib("product"); b = prod.create_baker(); hip = ot.Hip()
    tex, uv, ix, lv, kw = wl.workload(cfg); kw = dict(kw); lvl = kw.pop("level")
    t = prod.create_texture(b, [tex], alpha_cutoff=0.5)
    d = ot.make_desc(t, uv, ix, lvl, levels=lv, **kw)
    keep = [torch.from_numpy(uv).cuda(), torch.from_numpy(ix.astype(np.int32)).cuda()]
    dd = ot.BakeInputDesc.from_buffer_copy(d); dd.texCoords, dd.indexBuffer = keep[0].data_ptr(), keep[1].data_ptr()
    if lv is not None:
        keep.append(torch.from_numpy(lv).cuda()); dd.subdivisionLevels = keep[2].data_ptr()

    token = torch.zeros(1, dtype=torch.int32)
    clock = {"t0": 0.0, "phases": []}

    def acquire():
        if rank > 0:
            dist.recv(token, src=rank - 1)
        clock["t0"] = time.perf_counter()

    def release(stream):
        if stream:
            torch.cuda.ExternalStream(int(stream)).synchronize()
        else:
            torch.cuda.synchronize()
        clock["phases"].append(time.perf_counter() - clock["t0"])
        if rank < world - 1:
            dist.send(token, dst=rank + 1)

    ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

    class Collectives(C.Structure):
        _fields_ = [("allReduceU32", ALLREDUCE), ("allGatherBytes", ALLGATHER), ("user", C.c_void_p)]

    def all_reduce(_u, send, recv, count, op, stream):
        try:
            release(stream)
            src = sh.device_tensor(torch, send, count, torch.int32)
            dst = src if recv == send else sh.device_tensor(torch, recv, count, torch.int32)
            if op == 0:
                tt = src if recv == send else src.clone()
                dist.all_reduce(tt, op=dist.ReduceOp.SUM)
                if recv != send:
                    dst.copy_(tt)
            else:
                tt = src.to(torch.int64) & 0xFFFFFFFF
                dist.all_reduce(tt, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.MIN)
                dst.copy_(torch.where(tt >= 2 ** 31, tt - 2 ** 32, tt).to(torch.int32))
            torch.cuda.current_stream().synchronize()
            dist.barrier()      # (every rank has its result: the GPU is idle when rank 0 takes the next turn)
            acquire()
            return 0
        except Exception:
            import traceback; traceback.print_exc(); return 1

    def all_gather(_u, send, recv, nbytes, stream):
        try:
            release(stream)
            src = sh.device_tensor(torch, send, nbytes, torch.uint8); out = sh.device_tensor(torch, recv, nbytes * world, torch.uint8)
            dist.all_gather(list(out.view(world, -1).unbind(0)), src)
            torch.cuda.current_stream().synchronize()
            dist.barrier()
            acquire()
            return 0
        except Exception:
            import traceback; traceback.print_exc(); return 1

    cbs = (ALLREDUCE(all_reduce), ALLGATHER(all_gather)); table = Collectives(cbs[0], cbs[1], None)
    prod.dll.ommxCommFromCollectives.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    comm = C.c_void_p(); assert prod.dll.ommxCommFromCollectives(C.byref(table), rank, world, C.byref(comm)) == 0
    prod.dll.ommxDestroyDeviceBakeResult.argtypes = [C.c_void_p]

    # single-GPU reference time of the same bake, alone on the GPU (rank 0, the others wait)
    single = None
    if rank == 0:
        out = C.c_void_p(); ts = []
        prod.dll.ommxBakeDevice.argtypes = [C.c_void_p, C.POINTER(ot.BakeInputDesc), C.POINTER(C.c_void_p)]
        for i in range(bakes + 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            assert prod.dll.ommxBakeDevice(b, C.byref(dd), C.byref(out)) == 0
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            prod.dll.ommxDestroyDeviceBakeResult(out)
        single = float(np.median(ts[2:]))
    dist.barrier()
    runs = []
    for i in range(bakes + 2):
        dist.barrier()
        clock["phases"] = []
        acquire()
        out = sh.sharded_bake_rccl(prod.dll, b, C.byref(dd), comm)
        release(None)
        tm = bench.get_timings(prod, b)
        prod.dll.ommxDestroyDeviceBakeResult(out)
        if i >= 2:
            runs.append(list(clock["phases"]))
    json.dump({"single": single, "runs": runs, "exchange": int(tm.exchangeBytes), "contribution": int(tm.contributionBytes),
               "events_ms": {"setup": tm.setupMs, "triage": tm.triageMs, "classify": tm.classifyMs, "digest": tm.digestMs, "tail": tm.tailMs, "gather": tm.gatherMs}}, open(os.path.join(outdir, "rank%d.json" % rank), "w"))
    dist.barrier()
    prod.dll.ommxRcclCommDestroy(comm)
    dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile, torch.multiprocessing as mp
    # N processes on ONE GPU: with the HIP runtime's default of up to four hardware queues per process, four or more processes oversubscribe the GPU's
    # hardware-queue scheduler and every phase of every rank waits a scheduler quantum (10 ms at 4 processes, 33 ms at 8: profiles/r06_ranks_in_turn_c2.jsonl).
    # One queue per TEST process removes the artefact (an environment variable of the HIP runtime; the library itself reads none).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
    world = int(sys.argv[1]); cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"; bakes = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(worker, args=(world, port, td, cfg, bakes), nprocs=world, join=True)
        res = [json.load(open(os.path.join(td, "rank%d.json" % r))) for r in range(world)]
    nph = min(len(run) for r in res for run in r["runs"])
    # per phase: median over the bakes of (max over the ranks)
    per_phase = []
    for k in range(nph):
        per_phase.append(float(np.median([max(res[r]["runs"][i][k] for r in range(world)) for i in range(len(res[0]["runs"]))])) * 1e3)
    mean_phase = [float(np.median([np.mean([res[r]["runs"][i][k] for r in range(world)]) for i in range(len(res[0]["runs"]))])) * 1e3 for k in range(nph)]
    single = res[0]["single"] * 1e3
    total = sum(per_phase)
    print(json.dumps({"config": cfg, "world": world, "single_gpu_ms": round(single, 2), "phases_ms_max_over_ranks": [round(x, 3) for x in per_phase],
                      "phases_ms_mean_over_ranks": [round(x, 3) for x in mean_phase], "compute_side_step_ms": round(total, 2),
                      "compute_side_speedup": round(single / total, 2), "rank0_device_events_ms_last_bake": {k: round(v, 3) for k, v in res[0]["events_ms"].items()}, "exchange_bytes_per_rank": res[0]["exchange"], "contribution_bytes_per_rank": res[0]["contribution"],
                      "phases": "classification of the share | (agreement) metadata merge | tail | (agreement) codec | (all-gather) expansion + scatter + result"}))
