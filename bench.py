#!/usr/bin/env python3
"""bench.py -- micro-triangles classified per second + bake wall time of the MI355X opacity-micromap baker.

A "step" is one complete bake (the whole hot path: work-item setup, SAT coarse pass, level-line fine pass, special-index promotion,
XXH64 dedup, spatial sort, pack, index buffer).  Default workload = BASELINE.json's metric configuration (configs[2] / [3]):
    1 M random-UV triangles, 4096^2 foliage-style UNORM8 alpha (texture alphaCutoff = 0.5 => SAT on), level 8, 4-state, Wrap / Linear
`--config c1 | c2 | c4 | cards` selects another of tests/workloads.py (configs[1], configs[4] on one GPU, and an asset-shaped bake whose
micro-triangles span several texels); the driver's line is the default, c2.

Prints ONE JSON line (rank 0):
  value / ms_per_step   micro-triangles of all unique work items / wall time of a step through ommCpuBake, the SDK entry point: host arrays in, host
  = bake_wall_time_ms   arrays out, PCIe inclusive (SURVEY.md section 8d defines both metrics on it); K timed steps after W warm-up steps; details: host_api
  device_resident       the SAME bake through ommxBakeDevice (UV / index inputs and result arrays resident in HBM), same steps and warm-up.
                        (N > 1 GPUs: the sharded device-resident entry is the headline; value_same_entry_as_n_gt_1 = the same entry at every N.)
  host_api              details of the ommCpuBake steps: how arrayData reached the host (result_transfer: plain / streamed / compressed, --result-transfer;
                        bytes over PCIe, codec and expansion times, helper threads), --devices N = the same call spread over N devices of this process
  roofline              what limits the dominant kernel (classify_tiles): VALU issue slots; roofline_hbm = the HBM view on the units the
                        launch really processes; cpu_baseline = the oracle (port of the reference CPU baker) on the host cores, same run
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ommtest as ot  # noqa: E402
import workloads as wl  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_PEAK = 0.5      # one wave64 VALU instruction per SIMD every 2 cycles (SIMD-32 lanes), the plain-fp32 issue limit of gfx950
NUM_SIMDS = 256 * 4

# per config: triangles, CPU-baseline sample (triangles), description
CONFIGS = {
    "c1": dict(tris=100000, cpu_sample=20000, what="BASELINE configs[1]: %d random-UV triangles (10 texels), 2048x2048 value-noise UNORM8 alpha + SAT, subdiv level 6, 4-state, Wrap/Linear"),
    "c2": dict(tris=1000000, cpu_sample=20000, what="%d random-UV triangles (8.0 texels), 4096x4096 foliage-style UNORM8 alpha + SAT, subdiv level 8, 4-state, Wrap/Linear"),
    "c4": dict(tris=4000000, cpu_sample=3000, what="BASELINE configs[4] on ONE GPU: %d random-UV triangles (3 texels), per-triangle levels 4..10 (75 %%) + dynamic heuristic (25 %%, scale 2, max 10), 8192x8192 foliage UNORM8 alpha + SAT, 4-state, Wrap/Linear, dedup on"),
    "cards": dict(tris=40000, cpu_sample=600, what="asset-shaped: %d triangles = axis-aligned quads covering 256..1024 texels of a 4096x4096 foliage UNORM8 alpha + SAT, per-quad levels 6..8 (micro-triangles of 1..16 texels: the generic texel-loop path), 4-state, Clamp/Linear"),
}


class BakeTimings(C.Structure):
    _fields_ = [("hostSetupMs", C.c_float), ("uploadMs", C.c_float), ("classifyMs", C.c_float), ("digestMs", C.c_float),
                ("tailMs", C.c_float), ("gatherMs", C.c_float), ("downloadMs", C.c_float), ("totalMs", C.c_float),
                ("microTriangles", C.c_uint64), ("uniqueItems", C.c_uint32), ("classifyLaunches", C.c_uint32),
                ("stateBytes", C.c_uint64), ("triageMs", C.c_float), ("activeItems", C.c_uint32), ("fineMicroTriangles", C.c_uint64), ("setupMs", C.c_float),
                ("streamChunks", C.c_uint32), ("streamedBytes", C.c_uint64), ("streamTailMs", C.c_float),
                ("openTiles", C.c_uint32), ("openTileMicroTriangles", C.c_uint64), ("streamEarlyItems", C.c_uint32), ("persistentMs", C.c_float),
                ("genericMs", C.c_float), ("genericMicroTriangles", C.c_uint64), ("exchangeBytes", C.c_uint64), ("contributionBytes", C.c_uint64),
                ("streamPreviewMs", C.c_float), ("streamFirstCopyMs", C.c_float), ("streamLastCopyMs", C.c_float), ("streamRangeReadyMs", C.c_float * 32),
                ("resultTransfer", C.c_uint32), ("expandThreads", C.c_uint32), ("compressedBytes", C.c_uint64), ("compressMs", C.c_float), ("expandMs", C.c_float), ("devices", C.c_uint32),
                ("prefilledBytes", C.c_uint64), ("expandSkippedBytes", C.c_uint64)]


def get_timings(lib, baker):
    """ommxGetLastBakeTimingsSized into this file's mirror of ommxBakeTimings (the unsized symbol only fills the round-3 prefix of the struct)"""
    tm = BakeTimings()
    fn = lib.dll.ommxGetLastBakeTimingsSized
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    fn(baker, C.byref(tm), C.sizeof(tm), None)
    return tm


def source_hash():
    """sha256 (16 hex digits) over omm_amd/csrc, the same recipe as profiles/summarize_pmc.py: ties a committed PMC summary to the sources"""
    import hashlib
    d = os.path.join(ROOT, "omm_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp", ".inc")) or f == "Makefile":
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def create_texture_ms(prod, baker, size, seed):
    """ommCpuCreateTexture of a size x size UNORM8 texture with alphaCutoff (upload + summed-area table on the device), best of two;
    the reference builds the same object serially on one host thread (texture_impl.cpp:77-224)"""
    tex = ot.foliage_texture(seed, size, size, feature=64)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        th = prod.create_texture(baker, [tex], alpha_cutoff=0.5)
        dt = (time.perf_counter() - t0) * 1e3
        prod.destroy_texture(baker, th)
        best = dt if best is None else min(best, dt)
    return best


def host_info():
    import multiprocessing
    model, sockets = "unknown", set()
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                sockets.add(ln.split(":", 1)[1].strip())
    except OSError:
        pass
    cores = multiprocessing.cpu_count()
    eff, why = effective_cpus(cores)
    return cores, eff, "%s, %d socket(s), %d hardware threads; %d CPUs usable by this process (%s)" % (model, max(1, len(sockets)), cores, eff, why)


def effective_cpus(hardware_threads):
    """CPUs this process can really run on at once: the scheduler affinity mask, capped by the cgroup CPU quota (cpu.max of cgroup v2, cfs quota of v1).
    A container with a quota of 16 CPUs on a 256-thread host gets 16 CPUs' worth of time however many threads it starts."""
    n, why = hardware_threads, "no affinity mask or cgroup quota below the hardware threads"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, why = max(1, int(quota + 0.5)), "cgroup cpu quota %.1f" % quota
    return n, why


def desc_for(tex_handle, uv, ix, lv, kw, extra_flags=0):
    k = dict(kw)
    level = k.pop("level")
    return ot.make_desc(tex_handle, uv, ix, level, levels=lv, filt=ot.LINEAR, flags=ot.FLAG_THREADS | extra_flags, **k)


def micro_triangles_of(lib, baker, desc):
    """sum of 4^level over the unique work items of a bake (product library: from its timings)"""
    return int(get_timings(lib, baker).microTriangles)


def oracle_timings(orc):
    tm = (C.c_double * 8)()
    orc.dll.oracle_ommxGetLastBakeTimings(tm)
    return {"setup_s": tm[0], "coarse_s": tm[1], "fine_s": tm[2], "tail_s": tm[3], "sum_s": tm[4], "fine_micro_triangles": tm[5], "micro_triangles": tm[6], "threads": int(tm[7])}


def cpu_baseline(tex, uv, ix, lv, kw, sample, sat=True, grow=True, sweep=True):
    """The oracle (bit-exact restatement of the reference CPU baker, OpenMP over work items like the reference) timed on a bounded sample
    of the same triangle stream: the reference needs 2 * 4^N bytes per work item (131 GB at the full metric configuration).
    The main run uses as many threads as this process has CPUs (affinity mask and cgroup quota: `effective_cpus`, printed next to the hardware threads) on a
    sample that grows until it is >= 8 s of wall time, with the oracle's own phase clocks (set-up, ResampleCoarse, ResampleFine, serial tail:
    oracle_ommxGetLastBakeTimings); a quarter of the sample is baked again at twice that many threads and at all hardware threads (oversubscribed when
    the quota is the limit): they are reported in threads_sweep; `value` is the rate of the main run, `cores` its thread count."""
    hw, eff, host = host_info()
    orc = ot.Lib("oracle")
    orc.dll.oracle_ommxSetThreads(eff)
    b = orc.create_baker()
    t = orc.create_texture(b, [tex], alpha_cutoff=0.5 if sat else -1.0)
    n = min(sample, ix.size // 3)
    while True:   # the sample grows until it is >= 8 s of CPU work (once or twice: the time per triangle is flat), or the whole workload
        suv, six, slv = wl.subset(uv, ix, lv, 0, n)
        d = desc_for(t, suv, six, slv, kw)
        t0 = time.time()
        res = orc.bake(b, d, want_stats=False)
        dt = time.time() - t0
        if dt >= 8.0 or n >= ix.size // 3 or not grow: break
        n = min(ix.size // 3, int(n * min(16.0, 12.0 / max(dt, 1e-3))) + 1)
    ph = oracle_timings(orc)
    threads = {str(ph["threads"]): {"micro_triangles_per_s": ph["micro_triangles"] / dt, "sample_triangles": n, "seconds": dt}}
    if sweep:
        nq = max(1, n // 4)
        quv, qix, qlv = wl.subset(uv, ix, lv, 0, nq)
        for th in sorted(set((min(hw, 2 * eff), hw)) - {eff}):
            orc.dll.oracle_ommxSetThreads(th)
            t0 = time.time()
            orc.bake(b, desc_for(t, quv, qix, qlv, kw), want_stats=False)
            dq = time.time() - t0
            threads[str(th)] = {"micro_triangles_per_s": oracle_timings(orc)["micro_triangles"] / dq, "sample_triangles": nq, "seconds": dq}
        orc.dll.oracle_ommxSetThreads(0)
    orc.destroy_texture(b, t)
    orc.destroy_baker(b)
    best = max(threads, key=lambda k: threads[k]["micro_triangles_per_s"])
    main = str(ph["threads"])   # `value` / `cores`: the run with one thread per usable CPU (the sweep's other entries are oversubscribed; best_threads names the fastest)
    out = {"unit": "micro-triangles/s", "value": threads[main]["micro_triangles_per_s"], "cores": int(main), "kind": "port", "host": host,
           "host_threads": hw, "effective_cpus": eff, "threads_sweep": threads, "best_threads": int(best),
           "sample_triangles": n, "seconds": dt, "phases_at_effective_cpus_s": {k: ph[k] for k in ("setup_s", "coarse_s", "fine_s", "tail_s")},
           "note": "same loop structure as the reference (static-schedule OpenMP over work items, serial set-up and tail); `value` / `cores` = the run with one thread per usable CPU; "
                   "`effective_cpus` = what the affinity mask / cgroup quota let this process use at once (thread counts above it are oversubscribed: they say "
                   "nothing about how the loops scale on more cores); the phases are the oracle's own wall clocks of the run at effective_cpus threads"}
    return out, res, (suv, six, slv), dt, ph


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: re-exec the same command line under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1 at a free port).  os.execv: no wrapper process stays between the driver's clock and the ranks."""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--tris", type=int, default=0, help="override the configuration's triangle count")
    ap.add_argument("--host-api-steps", type=int, default=-1, help="timed bakes through ommCpuBake (host arrays in / out); -1 = --steps, 0 = skip")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="triangles baked by the CPU baseline (-1 = the configuration's default, 0 = skip)")
    ap.add_argument("--create-texture", type=int, default=1, help="time ommCpuCreateTexture at 4K and 8K (0 = skip)")
    ap.add_argument("--sat-off-sample", type=int, default=50000, help="c2 only: triangles of the SAT-off (no coarse pass) GPU measurement; CPU uses 1/20 of it (0 = skip)")
    ap.add_argument("--generic-pass", type=int, default=0, help="ommxBakerKnob_GenericPass (0 = library default, 1 = inside the persistent launch, 2 = deferred pass)")
    ap.add_argument("--stream-chunks", type=int, default=0, help="ommxBakerKnob_StreamChunks for the ommCpuBake measurement (0 = library default)")
    ap.add_argument("--result-transfer", type=int, default=0, help="ommxBakerKnob_ResultTransfer for the ommCpuBake measurement (0 = library default, 1 = plain copy, 2 = streamed placement, 3 = compressed)")
    ap.add_argument("--devices", type=int, default=0, help="ommxBakerKnob_Devices for the ommCpuBake measurement: the bake spread over N devices of THIS process (on a one-GPU box the ranks share the device)")
    ap.add_argument("--expand-threads", type=int, default=0, help="ommxBakerKnob_ExpandThreads (0 = library default)")
    ap.add_argument("--zero-ahead", type=int, default=0, help="ommxBakerKnob_ZeroAhead (0 = library default: on, 1 = off)")
    ap.add_argument("--concurrent", type=int, default=0, help="also measure K host threads baking concurrently on ONE baker through ommCpuBake (bakes/s for 1, 4, .. K threads; "
                                                              "the reference documents caller-level parallelism as a first-class strategy, docs/integration_guide.md:434)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this process becomes the launcher -- one rank per GPU under torch.distributed.run, same arguments,
        # rank 0's JSON line on this process's stdout.  (The driver's own torch.distributed.run command line lands in the branch below.)
        relaunch_under_torchrun(args.gpus)
    cfg = CONFIGS[args.config]
    tris = args.tris or cfg["tris"]
    host_steps = args.steps if args.host_api_steps < 0 else args.host_api_steps
    cpu_sample = cfg["cpu_sample"] if args.cpu_sample < 0 else args.cpu_sample

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d, or plainly as `python bench.py --gpus %d`" % (world, args.gpus, args.gpus, args.gpus)
    # self-test hooks (tests only): OMM_BENCH_ONE_GPU=1 puts every rank on GPU 0 and OMM_BENCH_BACKEND=gloo replaces RCCL, which
    # refuses two ranks on one device -- the driver's runs use neither
    torch.cuda.set_device(0 if os.environ.get("OMM_BENCH_ONE_GPU") == "1" else local)
    torch.zeros(1, device="cuda")   # the HIP context of THIS rank's device exists before the library allocates on "the current device"
    if world > 1:
        dist.init_process_group(os.environ.get("OMM_BENCH_BACKEND", "nccl"))

    tex, uv, ix, lv, kw = wl.workload(args.config, tris)
    tris = ix.size // 3

    prod = ot.Lib("product")
    prod.dll.ommxBakeDevice.argtypes = [C.c_void_p, C.POINTER(ot.BakeInputDesc), C.POINTER(C.c_void_p)]
    prod.dll.ommxGetDeviceBakeResultDesc.argtypes = [C.c_void_p, C.POINTER(C.POINTER(ot.BakeResultDesc))]
    prod.dll.ommxDestroyDeviceBakeResult.argtypes = [C.c_void_p]
    baker = prod.create_baker()
    if args.stream_chunks:
        prod.set_knob(baker, ot.KNOB_STREAM_CHUNKS, args.stream_chunks)
    if args.generic_pass:
        prod.set_knob(baker, ot.KNOB_GENERIC_PASS, args.generic_pass)
    if args.result_transfer:
        prod.set_knob(baker, ot.KNOB_RESULT_TRANSFER, args.result_transfer)
    if args.expand_threads:
        prod.set_knob(baker, ot.KNOB_EXPAND_THREADS, args.expand_threads)
    if args.zero_ahead:
        prod.set_knob(baker, ot.KNOB_ZERO_AHEAD, args.zero_ahead)
    th = prod.create_texture(baker, [tex], alpha_cutoff=0.5)
    host_desc = desc_for(th, uv, ix, lv, kw)
    # inputs resident in HBM before the timed region: torch owns the device buffers, the library gets raw pointers
    d_uv = torch.from_numpy(np.ascontiguousarray(uv)).cuda()
    d_ix = torch.from_numpy(ix.astype(np.int32)).cuda()
    d_lv = torch.from_numpy(np.ascontiguousarray(lv)).cuda() if lv is not None else None
    desc = ot.BakeInputDesc.from_buffer_copy(host_desc)
    desc.texCoords, desc.indexBuffer = d_uv.data_ptr(), d_ix.data_ptr()
    desc.subdivisionLevels = d_lv.data_ptr() if d_lv is not None else None

    import omm_amd.sharded as shard

    # N > 1: the library's one-call sharded bake (ommxShardedBakeRccl).  Its collectives are RCCL calls issued from C++; where the library cannot
    # have an RCCL communicator of its own (it could not be created, or the gloo self-test: two ranks on one GPU) the SAME call runs over the
    # process group's collectives (ommxCommFromCollectives).  OMM_BENCH_COLLECTIVES=torch keeps the caller-driven four-call path for the self-tests
    one_call = world > 1 and os.environ.get("OMM_BENCH_COLLECTIVES", "native") == "native"
    native = one_call and os.environ.get("OMM_BENCH_BACKEND", "nccl") == "nccl"
    comm = None
    borrowed = None
    if native:
        try:
            comm = shard.rccl_comm(prod.dll, torch, dist, rank, world)   # raises on every rank or on none
        except RuntimeError as e:
            if rank == 0:
                print("bench: native RCCL communicator unavailable (%s): the same call over torch.distributed's collectives instead" % e, file=sys.stderr)
            native = False
    if one_call and not native:
        borrowed = shard.CollectivesComm(prod.dll, torch, dist, rank, world)
        comm = borrowed.handle
    entry_n = ("ommxShardedBakeRccl (collectives issued by the library)" if native else
               ("ommxShardedBakeRccl over the process group's collectives (ommxCommFromCollectives)" if one_call else "ommxSharded* + torch.distributed"))

    def step():
        if one_call:
            return shard.sharded_bake_rccl(prod.dll, baker, C.byref(desc), comm)
        if world > 1:
            return shard.sharded_bake(prod.dll, baker, C.byref(desc), rank, world, torch, dist)
        out = C.c_void_p()
        r = prod.dll.ommxBakeDevice(baker, C.byref(desc), C.byref(out))
        assert r == ot.SUCCESS, "ommxBakeDevice failed: %d" % r
        return out

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if one_call:
        # one untimed probe bake: if the one-call entry fails on ANY rank (all ranks learn it), fall back together -- first to the same call over the
        # process group's collectives, then to the caller-driven four-call path -- instead of ending the run without a result line
        for attempt in range(2):
            ok = 1
            try:
                prod.dll.ommxDestroyDeviceBakeResult(step())
            except (RuntimeError, AssertionError) as e:
                ok = 0
                print("bench: rank %d: %s" % (rank, e), file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                break
            if native:
                if rank == 0:
                    print("bench: ommxShardedBakeRccl failed with the library's own communicator: the same call over torch.distributed's collectives instead", file=sys.stderr)
                native = False
                prod.dll.ommxRcclCommDestroy(comm)
                borrowed = shard.CollectivesComm(prod.dll, torch, dist, rank, world)
                comm = borrowed.handle
                entry_n = "ommxShardedBakeRccl over the process group's collectives (ommxCommFromCollectives)"
            else:
                if rank == 0:
                    print("bench: the one-call entry failed: caller-driven four-call path instead", file=sys.stderr)
                one_call = False
                entry_n = "ommxSharded* + torch.distributed"
                break
    if world > 1:
        # communicator set-up (seconds, once per process) is not part of a bake: force it before any step, whatever --warmup says
        w0 = torch.zeros(4, dtype=torch.int32, device="cuda")
        dist.all_reduce(w0)
        g0 = torch.empty(4 * world, dtype=torch.int32, device="cuda")
        dist.all_gather_into_tensor(g0, w0)
        torch.cuda.synchronize()
    last = None
    for _ in range(args.warmup):
        if last is not None:
            prod.dll.ommxDestroyDeviceBakeResult(last)
        last = step()
    sync()
    t0 = time.perf_counter()
    tms = []
    for _ in range(args.steps):
        if last is not None:
            prod.dll.ommxDestroyDeviceBakeResult(last)
        last = step()
        tms.append(get_timings(prod, baker))
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    micro_tris = float(tms[-1].microTriangles)  # whole job: every rank reports the full work-item list

    pd = C.POINTER(ot.BakeResultDesc)()
    prod.dll.ommxGetDeviceBakeResultDesc(last, C.byref(pd))
    rd = pd.contents
    result_info = {"arrayDataBytes": int(rd.arrayDataSize), "descs": int(rd.descArrayCount), "triangles": int(rd.indexCount)}
    prod.dll.ommxDestroyDeviceBakeResult(last)

    # ---- the SDK entry point proper: host arrays in, host arrays out (metric 2) ----
    host_ms, host_first_ms, host_tms = None, None, []
    if args.devices:
        prod.set_knob(baker, ot.KNOB_DEVICES, args.devices)
    if rank == 0 and host_steps > 0:
        t1 = time.perf_counter()
        r, out = prod.bake_raw(baker, host_desc)
        assert r == ot.SUCCESS
        prod.fn("ommCpuDestroyBakeResult")(out)
        host_first_ms = (time.perf_counter() - t1) * 1e3   # cold: the pinned staging buffer and the result pages are set up
        for _ in range(max(0, args.warmup - 1)):
            r, out = prod.bake_raw(baker, host_desc)
            assert r == ot.SUCCESS
            prod.fn("ommCpuDestroyBakeResult")(out)
        t1 = time.perf_counter()
        for _ in range(host_steps):
            r, out = prod.bake_raw(baker, host_desc)
            assert r == ot.SUCCESS
            host_tms.append(get_timings(prod, baker))
            prod.fn("ommCpuDestroyBakeResult")(out)
        host_ms = (time.perf_counter() - t1) / host_steps * 1e3
    if args.devices:
        prod.set_knob(baker, ot.KNOB_DEVICES, 0)

    comm_info = None
    if comm is not None:   # what the communicator itself says about its size: a SCALE line proves that the collectives really ran over N ranks
        cr, cw = C.c_uint32(0), C.c_uint32(0)
        prod.dll.ommxRcclCommInfo.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        if prod.dll.ommxRcclCommInfo(comm, C.byref(cr), C.byref(cw)) == ot.SUCCESS:
            comm_info = {"kind": "RCCL communicator made by the library (ncclCommCount / ncclCommUserRank)" if native else "caller collectives over torch.distributed (%s)" % dist.get_backend(),
                         "ranks": int(cw.value), "this_rank": int(cr.value)}
    if rank == 0:
        dev_ms = elapsed / args.steps * 1e3
        # headline: the SDK entry point itself (SURVEY.md section 8d defines both metrics on ommCpuBake); the device-resident entry is the named secondary.
        # N > 1 (and runs that skip the host bakes) have only the device-resident entry.
        headline_host = host_ms is not None and world == 1
        ms_per_step = host_ms if headline_host else dev_ms
        avg = lambda f, ts=tms: float(np.mean([getattr(t, f) for t in ts]))
        classify_ms = avg("classifyMs")
        persistent_ms = avg("persistentMs") or classify_ms
        generic_ms = avg("genericMs")
        # the dominant kernel: the persistent classify_tiles launch, or -- bakes of asset-sized triangles -- the deferred generic pass behind it
        dom_kernel = "classify_generic" if generic_ms > persistent_ms else "classify_tiles"
        tiles_ms = persistent_ms
        if dom_kernel == "classify_generic":
            persistent_ms = generic_ms   # (every roofline figure below is about the dominant kernel)
        t_last = tms[-1]
        bits = 2
        line = {
            "communicator": comm_info,
            "metric": "micro-triangles classified/sec (whole node)", "value": micro_tris / (ms_per_step * 1e-3),
            "unit": "micro-triangles/s", "n_gpus": world, "steps": host_steps if headline_host else args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["what"] % tris, "name": args.config,
                       "entry": "ommCpuBake (the SDK entry point: host arrays in, host arrays out, PCIe inclusive) with bakeFlags = ommCpuBakeFlags_EnableInternalThreads "
                                "(the baker's helper threads expand the compressed result; ommCpuBakeInputDescDefault has no flags: without it the result is streamed, ~2x the time)" if headline_host else
                                (entry_n if world > 1 else "ommxBakeDevice") + " (ommCpuBake contract, UV/index inputs and result arrays resident in HBM)",
                       "sharding": ("active work items partitioned over ranks; RCCL all-reduce of item metadata + all-gather of the OMM blocks as codec streams "
                                    "(%d of %d contribution bytes per rank on the wire)" % (int(tms[-1].exchangeBytes), int(tms[-1].contributionBytes))) if world > 1 else "none",
                       "result": result_info, "unique_items": int(t_last.uniqueItems), "active_items": int(t_last.activeItems),
                       "open_tiles": int(t_last.openTiles), "fine_micro_triangles": int(t_last.fineMicroTriangles),
                       "generic_pass_micro_triangles": int(t_last.genericMicroTriangles)},
            # `value` / `ms_per_step` / `bake_wall_time_ms`: the SDK call a drop-in user makes, ommCpuBake, host arrays in and out, PCIe inclusive (SURVEY.md
            # section 8d defines both metrics on it).  `device_resident`: the same bake through ommxBakeDevice (inputs and result arrays in HBM), same steps.
            "value_entry": "ommCpuBake" if headline_host else ("ommxBakeDevice" if world == 1 else ("ommxShardedBakeRccl" if one_call else "ommxSharded* + torch.distributed")),
            # A scaling curve over N must compare ONE entry: at N > 1 `value` is the device-resident sharded bake, so the N = 1 line carries the
            # device-resident single-GPU bake under this key as well (at N > 1 it repeats `value`)
            "value_same_entry_as_n_gt_1": {"entry": "ommxBakeDevice" if world == 1 else entry_n, "value": micro_tris / (dev_ms * 1e-3), "ms_per_step": dev_ms,
                                           "note": "the device-resident bake (UV / index inputs and result arrays in HBM): the entry whose sharded form is `value` for n_gpus > 1"},
            "bake_wall_time_ms": ms_per_step,
            "bake_wall_time_entry": "ommCpuBake (host arrays in/out, PCIe inclusive)" if headline_host else "device-resident entry (ommCpuBake was not timed in this run)",
            "device_resident": {"entry": "ommxBakeDevice" if world == 1 else entry_n, "ms_per_bake": dev_ms, "micro_triangles_per_s": micro_tris / (dev_ms * 1e-3), "bakes": args.steps,
                                "note": "the ommCpuBake contract with the UV / index inputs and the result arrays resident in HBM (no PCIe in the timed region)"},
            "rates": {"all_work_items": micro_tris / (ms_per_step * 1e-3),
                      "open_tiles_only": float(t_last.openTileMicroTriangles) / ((tiles_ms + generic_ms) * 1e-3) if tiles_ms + generic_ms > 0 else None,
                      "note": "`value` counts 4^level micro-triangles for every unique work item like the reference's loop does; most of them are settled by hierarchical queries "
                              "(summed-area table, curve-free regions) without per-micro-triangle work, so the rate over the micro-triangles of the tiles that reach classify_tiles "
                              "(device-resident bake, kernel time) is given too; the fine pass alone: fine_pass_only, measured on the CPU baseline's sample on both sides"},
            "phases_ms": {k: avg(k) for k in ("uploadMs", "setupMs", "triageMs", "classifyMs", "persistentMs", "genericMs", "digestMs", "tailMs", "gatherMs", "downloadMs", "totalMs")},
        }
        if host_ms is not None:
            havg = lambda f: avg(f, host_tms)
            line["host_api"] = {"entry": "ommCpuBake (host arrays in/out, PCIe inclusive)", "ms_per_bake": host_ms, "bakes": host_steps, "first_call_ms": host_first_ms,
                                "micro_triangles_per_s": micro_tris / (host_ms * 1e-3),
                                "result_transfer": {"mode": ["auto", "plain copy", "streamed placement (DMA engine, while the classification runs)",
                                                             "compressed (codec stream over PCIe, expanded by the baker's helper threads)"][int(host_tms[-1].resultTransfer) & 3],
                                                    "array_data_bytes": result_info["arrayDataBytes"], "bytes_over_pcie": int(host_tms[-1].compressedBytes) or int(host_tms[-1].streamedBytes) or result_info["arrayDataBytes"],
                                                    "codec_and_readback_ms": havg("compressMs"), "copy_and_expand_ms": havg("expandMs"), "expand_threads": int(host_tms[-1].expandThreads),
                                                    "zeroed_ahead_bytes": int(host_tms[-1].prefilledBytes), "expansion_skipped_bytes": int(host_tms[-1].expandSkippedBytes),
                                                    "copy_and_expand_ms_min_max": [float(min(t.expandMs for t in host_tms)), float(max(t.expandMs for t in host_tms))],
                                                    "bake_ms_min_max": [float(min(t.totalMs for t in host_tms)), float(max(t.totalMs for t in host_tms))],
                                                    "bake_ms_p50_p95": [float(np.percentile([t.totalMs for t in host_tms], 50)), float(np.percentile([t.totalMs for t in host_tms], 95))],
                                                    "copy_and_expand_ms_p50_p95": [float(np.percentile([t.expandMs for t in host_tms], 50)), float(np.percentile([t.expandMs for t in host_tms], 95))],
                                                    "expand_GBps": result_info["arrayDataBytes"] / (havg("expandMs") * 1e6) if havg("expandMs") > 0 else None,
                                                    "devices": int(host_tms[-1].devices) or 1},
                                "stream": {"ranges": int(host_tms[-1].streamChunks), "streamed_bytes": int(host_tms[-1].streamedBytes), "exposed_copy_ms": havg("streamTailMs"),
                                           "early_items": int(host_tms[-1].streamEarlyItems),
                                           "preview_ms": havg("streamPreviewMs"), "first_copy_issued_ms": havg("streamFirstCopyMs"), "last_copy_issued_ms": havg("streamLastCopyMs"),
                                           "range_ready_ms": [round(float(np.mean([t.streamRangeReadyMs[k] for t in host_tms])), 3) for k in range(min(32, int(host_tms[-1].streamChunks) + 1))],
                                           "note": "ranges > 0: finished OMM blocks cross PCIe (SDMA) straight to their final arrayData offsets while the ONE persistent classification "
                                                   "launch is still running, one copy per range of work items; exposed = from the end of the classification to the last byte on the "
                                                   "host; early_items = possible duplicates, classified with an earlier range than their own; ranges == 0: one copy after the bake "
                                                   "(small results, or a classification so long that the copy is not worth a quarter of its speed)"},
                                "phases_ms": {k: havg(k) for k in ("uploadMs", "setupMs", "triageMs", "classifyMs", "digestMs", "tailMs", "gatherMs", "downloadMs", "totalMs")}}

        # ---- roofline of the dominant kernel, classify_tiles (the persistent launch of the levels >= 6) ----
        # The kernel is bound by VALU issue slots, not by HBM and not by MFMA.  Instruction counts per launch come from the separate
        # rocprofv3 --pmc passes of profiles/collect.sh (they cannot be read from inside this process) and apply only to the workload AND the library
        # sources they were collected on: the summary carries a hash of omm_amd/csrc, and a summary of other sources is not used.
        launches = 1
        # HBM view on the units the launch really processes: packed states of the OPEN tiles (0.25 B per 4-state micro-triangle), their 48-byte records,
        # one pass over the texture and its summed-area table (5 B per texel)
        tw, thh = tex.shape[1], tex.shape[0]
        alg_bytes = bits / 8.0 * float(t_last.openTileMicroTriangles) + 48.0 * t_last.openTiles + tw * thh * 5.0
        formula = "(0.25 B x micro-triangles of the open tiles + 48 B x open tiles + 5 B x texels) / HIP-event duration of the launch"
        if dom_kernel == "classify_generic":   # queue entry read (8 B) + state ORed in (4 B word) per queued micro-triangle, one pass over the texture
            alg_bytes = 12.0 * float(t_last.genericMicroTriangles) + tw * thh * (4.0 if tex.dtype == np.float32 else 1.0)
            formula = "(12 B x queued micro-triangles + one pass over the texels) / HIP-event duration of the launch"
        hbm_achieved = alg_bytes / (persistent_ms * 1e-3) / 1e9 if persistent_ms > 0 else 0.0
        roof_hbm = {"bound": "hbm", "kernel": dom_kernel, "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS, "traffic": None,
                    "avg_launch_ms": persistent_ms, "algorithmic_bytes_per_launch": alg_bytes,
                    "formula": formula,
                    "all_work_items_view": {"note": "SURVEY.md section 8d per-unit figure (0.25 B per micro-triangle of EVERY unique work item + 24 B per item + 5 B per texel) over the same duration; "
                                                    "95 % of those micro-triangles are settled by hierarchical SAT queries and never reach this launch",
                                            "achieved": (bits / 8.0 * micro_tris + 24.0 * t_last.uniqueItems + tw * thh * 5.0) / (persistent_ms * 1e-3) / 1e9 if persistent_ms > 0 else None}}
        roof_hbm["all_work_items_view"]["frac"] = (roof_hbm["all_work_items_view"]["achieved"] or 0.0) / HBM_PEAK_GBS
        line["roofline"] = roof_hbm
        tpath = os.path.join(ROOT, "profiles", "pmc_latest_%s.json" % args.config)
        default_workload = tris == cfg["tris"]
        if world == 1 and default_workload and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("source_sha256_16") == source_hash() and tj.get("kernel", "classify_tiles") == dom_kernel:
                src = ("profiles/%s_pmc.md -- NOT counters of this run: separate rocprofv3 --pmc passes of `%s`, collected %s on %s; %s; per-launch counts are deterministic for the "
                       "workload and are used only when the hash of omm_amd/csrc matches (%s)") % (tj.get("tag"), tj.get("command"), tj.get("collected_utc", "in an earlier round"),
                                                                                                 tj.get("collected_on", "a gpurun lease"), tj.get("corrections"), tj.get("source_sha256_16"))
                roof_hbm["traffic"] = tj.get("traffic_bytes_per_launch")
                roof_hbm["traffic_source"] = src
                valu = tj.get("valu_wave_instructions_per_launch"); hz = tj.get("shader_clock_hz")
                if valu and hz and persistent_ms > 0:
                    ipc = valu / (persistent_ms * 1e-3 * hz * NUM_SIMDS)
                    line["roofline"] = {"bound": "valu_issue", "kernel": dom_kernel, "achieved": ipc, "peak": VALU_ISSUE_PEAK, "unit": "VALU wave-instructions / cycle / SIMD",
                                        "frac": ipc / VALU_ISSUE_PEAK, "traffic": tj.get("traffic_bytes_per_launch"), "avg_launch_ms": persistent_ms,
                                        "formula": "SQ_INSTS_VALU per launch (%.4g, PMC pass) / (HIP-event duration of the launch x %.4g Hz shader clock (GRBM_GUI_ACTIVE / 8 / duration of the counter pass) x 1024 SIMDs) / 0.5" % (valu, hz),
                                        "calibrated_issue_utilisation": tj.get("valu_issue_utilisation"), "scalar_issue_utilisation": tj.get("scalar_issue_utilisation"),
                                        "valu_lane_utilisation": tj.get("valu_lane_util"), "scalar_per_valu": tj.get("scalar_per_valu"),
                                        "note": "calibrated = every instruction class priced with its measured issue cost (profiles/valu_rates_mi355x.json) instead of 2 cycles",
                                        "source": src}
                    line["roofline_hbm"] = roof_hbm
            else:
                roof_hbm["traffic_source"] = "none: %s was collected on other sources (%s, tree is %s)" % (os.path.basename(tpath), tj.get("source_sha256_16"), source_hash())
        # the streaming part of the path for comparison: final gather of the surviving OMM blocks into arrayData order
        gather_ms = avg("gatherMs")
        if gather_ms > 0 and world == 1:
            gb = 2.0 * result_info["arrayDataBytes"] + 8.0 * result_info["descs"] + 8.0 * result_info["triangles"]
            line["roofline_streaming"] = {"bound": "hbm", "kernel": "tail_gather_omms (+ narrow_indices)", "achieved": gb / (gather_ms * 1e-3) / 1e9,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / (gather_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and args.concurrent > 0:
            # K caller threads share one baker and one texture (the SDK allows concurrent ommCpuBake calls on a baker: docs/integration_guide.md:434); every
            # thread bakes the same desc through ommCpuBake + ommCpuDestroyBakeResult; ctypes releases the GIL for the duration of a call
            import threading
            per_thread = max(4, int(min(400, 2000.0 / max(ms_per_step, 0.05))))
            cc = {}
            ks = sorted(set(k for k in (1, 4, 16, args.concurrent) if k <= args.concurrent))
            for K in ks:
                def work():
                    for _ in range(per_thread):
                        r, out = prod.bake_raw(baker, host_desc)
                        assert r == ot.SUCCESS
                        prod.fn("ommCpuDestroyBakeResult")(out)
                ths = [threading.Thread(target=work) for _ in range(K)]
                tc = time.perf_counter()
                for t_ in ths: t_.start()
                for t_ in ths: t_.join()
                dtc = time.perf_counter() - tc
                cc[str(K)] = {"bakes_per_s": K * per_thread / dtc, "ms_per_bake_per_thread": dtc / per_thread * 1e3, "bakes": K * per_thread}
            line["concurrent_bakes"] = {"entry": "ommCpuBake from K threads on one baker", "threads": cc,
                                        "scaling_vs_one_thread": {k: v["bakes_per_s"] / cc[str(ks[0])]["bakes_per_s"] for k, v in cc.items()}}
        if world == 1 and args.create_texture and args.config == "c2":
            line["create_texture_ms"] = {"entry": "ommCpuCreateTexture, UNORM8 + alphaCutoff (H2D + device summed-area table)", "4096": create_texture_ms(prod, baker, 4096, 1234),
                                         "8192": create_texture_ms(prod, baker, 8192, 1234)}
        if cpu_sample > 0 and world == 1:
            cb, cpu_res, (suv, six, slv), dt, ph = cpu_baseline(tex, uv, ix, lv, kw, cpu_sample)
            # correctness gate of the same run: the HIP library bakes the CPU sample, byte-for-byte comparison
            sdesc = desc_for(th, suv, six, slv, kw)
            gpu_res = prod.bake(baker, sdesc, want_stats=False)
            assert gpu_res.same_as(cpu_res), "GPU result differs from the CPU baseline on its sample: " + gpu_res.diff(cpu_res)
            # the fine pass alone, SAME numerator on both sides: the micro-triangles of the sample that enter ResampleFine (counted by the oracle) over the
            # oracle's own clock around ResampleFine, and over the GPU's classification time (HIP events) of the same sample, second bake (warm)
            prod.bake(baker, sdesc, want_stats=False)
            stm = get_timings(prod, baker)
            cb["sample"] = "first %d triangles of the same seeded stream (%.3g micro-triangles), SAT on, %.1f s at %d threads" % (cb["sample_triangles"], ph["micro_triangles"], dt, ph["threads"])
            line["cpu_baseline"] = cb
            line["fine_pass_only"] = {"numerator": "micro-triangles of the CPU sample that enter ResampleFine (coarse pass left them UnknownOpaque): %.4g" % ph["fine_micro_triangles"],
                                      "cpu_micro_triangles_per_s": ph["fine_micro_triangles"] / ph["fine_s"] if ph["fine_s"] > 0 else None,
                                      "cpu_seconds": ph["fine_s"], "cpu_threads": ph["threads"],
                                      "gpu_micro_triangles_per_s": ph["fine_micro_triangles"] / (stm.classifyMs * 1e-3) if stm.classifyMs > 0 else None,
                                      "gpu_classify_ms": float(stm.classifyMs),
                                      "note": "GPU time = all classification kernels of the sample's bake (triage of tiles, persistent launch), which also do the coarse pass"}
            line["speedup_vs_cpu_baseline"] = {"whole_bake_ommCpuBake": (micro_tris / (host_ms * 1e-3)) / cb["value"] if host_ms else None,
                                               "whole_bake_device_entry": (micro_tris / (dev_ms * 1e-3)) / cb["value"],
                                               "note": "whole-bake ratios include the hierarchical shortcuts and the baseline's serial set-up and tail; fine_pass_only and sat_off (c2) are the kernel-vs-kernel pairs"}
            line["parity_vs_cpu_baseline"] = "bit-exact on the CPU sample (arrayData %d B, %d descs, index buffer, histograms, index format)" % (gpu_res.array_data.size, len(gpu_res.descs))
            # second texture mode (no summed-area table: every micro-triangle takes the level-line pass), bounded samples on both sides
            if args.sat_off_sample > 0 and args.config == "c2":
                ks = min(args.sat_off_sample, tris)
                th2 = prod.create_texture(baker, [tex], alpha_cutoff=-1.0)
                guv, gix, glv = wl.subset(uv, ix, lv, 0, ks)
                d2 = desc_for(th2, guv, gix, glv, kw)
                prod.bake(baker, d2, want_stats=False)
                t2 = time.perf_counter()
                prod.bake(baker, d2, want_stats=False)
                gpu_dt = time.perf_counter() - t2
                gmt = micro_triangles_of(prod, baker, None)
                kc = min(max(args.sat_off_sample // 20, 200), ks)
                cb2, cpu_res2, (cuv, cix, clv), dt2, ph2 = cpu_baseline(tex, uv, ix, lv, kw, kc, sat=False, grow=False, sweep=False)
                gpu_res2 = prod.bake(baker, desc_for(th2, cuv, cix, clv, kw), want_stats=False)
                assert gpu_res2.same_as(cpu_res2), "SAT-off GPU result differs from the CPU baseline: " + gpu_res2.diff(cpu_res2)
                cb2["sample"] = "first %d triangles, SAT off, %.1f s at %d threads" % (kc, dt2, ph2["threads"])
                prod.destroy_texture(baker, th2)
                line["sat_off"] = {"entry": "ommCpuBake (host arrays, PCIe inclusive), texture without alphaCutoff", "gpu_micro_triangles_per_s": gmt / gpu_dt,
                                   "gpu_sample": "first %d triangles, %.1f ms" % (ks, gpu_dt * 1e3), "cpu_baseline": cb2, "speedup": (gmt / gpu_dt) / cb2["value"], "parity": "bit-exact on the CPU sample"}
        print(json.dumps(line))
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()   # rank 0 has done more (host-API bakes, the JSON line): all ranks tear their communicators down together
    prod.destroy_texture(baker, th)
    prod.destroy_baker(baker)
    if borrowed is not None:
        borrowed.destroy()
    elif comm is not None:
        prod.dll.ommxRcclCommDestroy(comm)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
