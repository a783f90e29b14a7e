#!/usr/bin/env python3
"""bench.py -- micro-triangles classified per second by ommCpuBake() on MI355X.

A "step" is one complete bake (the whole hot path: work-item setup, SAT coarse pass, level-line fine pass, special-index
promotion, XXH64 dedup, spatial sort, pack, index buffer) of BASELINE.json's metric configuration:
    1 M random-UV triangles, 4096^2 foliage-style UNORM8 alpha (texture alphaCutoff = 0.5 => SAT on),
    subdivision level 8, 4-state, Wrap / Linear, ForceOpaque promotion                       (configs[2] / [3])
Prints ONE JSON line (rank 0).  `value` = micro-triangles of all unique work items / wall time of the step.
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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


class BakeTimings(C.Structure):
    _fields_ = [("hostSetupMs", C.c_float), ("uploadMs", C.c_float), ("classifyMs", C.c_float), ("digestMs", C.c_float),
                ("tailMs", C.c_float), ("gatherMs", C.c_float), ("downloadMs", C.c_float), ("totalMs", C.c_float),
                ("microTriangles", C.c_uint64), ("uniqueItems", C.c_uint32), ("classifyLaunches", C.c_uint32),
                ("stateBytes", C.c_uint64), ("triageMs", C.c_float), ("activeItems", C.c_uint32), ("fineMicroTriangles", C.c_uint64), ("setupMs", C.c_float)]


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


def make_workload(args):
    """Seeded synthetic inputs (identical on every rank and for the CPU baseline)."""
    tex = ot.foliage_texture(args.seed, args.tex, args.tex, feature=args.feature)
    uv, ix = ot.random_triangles(args.seed + 1, args.tris, args.extent_texels / args.tex)
    return tex, uv, ix


def bake_desc(tex_handle, uv, ix, args, lo, hi):
    return ot.make_desc(tex_handle, uv[3 * lo:3 * hi], ix[:3 * (hi - lo)], args.level, addr=ot.WRAP, filt=ot.LINEAR,
                        promo=ot.PROMO_FORCE_OPAQUE, fmt=ot.FMT_4STATE, flags=ot.FLAG_THREADS)


def cpu_baseline(args, tex, uv, ix, sat=True, sample=None):
    """The oracle (bit-exact restatement of the reference CPU baker, OpenMP over work items like the reference) timed on a
    bounded sample of the same triangle stream: the reference needs 2 * 4^N bytes per work item (131 GB at full size)."""
    import multiprocessing
    cores = multiprocessing.cpu_count()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    orc = ot.Lib("oracle")
    b = orc.create_baker()
    t = orc.create_texture(b, [tex], alpha_cutoff=0.5 if sat else -1.0)
    k = min(sample if sample is not None else args.cpu_sample, args.tris)
    d = bake_desc(t, uv, ix, args, 0, k)
    t0 = time.time()
    res = orc.bake(b, d, want_stats=False)
    dt = time.time() - t0
    orc.destroy_texture(b, t)
    orc.destroy_baker(b)
    mt = k * 4 ** args.level
    model, sockets = "unknown", set()
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                sockets.add(ln.split(":", 1)[1].strip())
    except OSError:
        pass
    return {"value": mt / dt, "unit": "micro-triangles/s", "cores": cores, "kind": "port",
            "note": "same loop structure as the reference (static-schedule OpenMP over work items, serial tail): with many threads the sample is dominated by the serial tail, "
                    "so value / this mostly measures that tail plus the hierarchical SAT shortcut; sat_off below is the kernel-vs-kernel pair", "host": "%s, %d socket(s), %d hardware threads" % (model, max(1, len(sockets)), cores),
            "sample": "first %d triangles of the same seeded stream (%.3g micro-triangles), SAT %s, %.1f s" % (k, mt, "on" if sat else "off", dt)}, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--level", type=int, default=8)
    ap.add_argument("--tex", type=int, default=4096)
    ap.add_argument("--feature", type=int, default=64, help="foliage blob size in texels")
    ap.add_argument("--extent-texels", type=float, default=8.0, help="triangle size in texels")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--host-api-steps", type=int, default=2, help="extra untimed-for-value bakes through ommCpuBake (host arrays)")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="triangles baked by the CPU baseline (0 = skip)")
    ap.add_argument("--create-texture", type=int, default=1, help="time ommCpuCreateTexture at 4K and 8K (0 = skip)")
    ap.add_argument("--sat-off-sample", type=int, default=50000, help="triangles of the SAT-off (no coarse pass) GPU measurement; CPU uses 1/20 of it (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # self-test hooks (tests only): OMM_BENCH_ONE_GPU=1 puts every rank on GPU 0 and OMM_BENCH_BACKEND=gloo replaces RCCL, which
    # refuses two ranks on one device -- the driver's runs use neither
    torch.cuda.set_device(0 if os.environ.get("OMM_BENCH_ONE_GPU") == "1" else local)
    torch.zeros(1, device="cuda")   # the HIP context of THIS rank's device exists before the library allocates on "the current device"
    if world > 1:
        dist.init_process_group(os.environ.get("OMM_BENCH_BACKEND", "nccl"))

    tex, uv, ix = make_workload(args)
    # strong scaling: every rank sees the whole (fixed) triangle stream; the library partitions the ACTIVE work items over the
    # ranks (ommxSharded*), metadata are merged by an RCCL all-reduce and the surviving OMM blocks by an RCCL all-gather
    lo, hi = 0, args.tris

    prod = ot.Lib("product")
    prod.dll.ommxGetLastBakeTimings.argtypes = [C.c_void_p, C.POINTER(BakeTimings)]
    prod.dll.ommxBakeDevice.argtypes = [C.c_void_p, C.POINTER(ot.BakeInputDesc), C.POINTER(C.c_void_p)]
    prod.dll.ommxGetDeviceBakeResultDesc.argtypes = [C.c_void_p, C.POINTER(C.POINTER(ot.BakeResultDesc))]
    prod.dll.ommxDestroyDeviceBakeResult.argtypes = [C.c_void_p]
    baker = prod.create_baker()
    th = prod.create_texture(baker, [tex], alpha_cutoff=0.5)
    host_desc = bake_desc(th, uv, ix, args, lo, hi)
    # inputs resident in HBM before the timed region: torch owns the device buffers, the library gets raw pointers
    d_uv = torch.from_numpy(np.ascontiguousarray(uv[3 * lo:3 * hi])).cuda()
    d_ix = torch.from_numpy(ix[:3 * (hi - lo)].astype(np.int32)).cuda()
    desc = ot.BakeInputDesc.from_buffer_copy(host_desc)
    desc.texCoords, desc.indexBuffer = d_uv.data_ptr(), d_ix.data_ptr()

    import omm_amd.sharded as shard

    # N > 1: the library's one-call sharded bake, RCCL collectives issued from C++ (ommxShardedBakeRccl); OMM_BENCH_COLLECTIVES=torch keeps
    # the caller-driven path (collectives through torch.distributed) for the gloo self-tests
    native = world > 1 and os.environ.get("OMM_BENCH_COLLECTIVES", "native") == "native" and os.environ.get("OMM_BENCH_BACKEND", "nccl") == "nccl"
    comm = None
    if native:
        try:
            comm = shard.rccl_comm(prod.dll, torch, dist, rank, world)   # raises on every rank or on none
        except RuntimeError as e:
            if rank == 0:
                print("bench: native RCCL communicator unavailable (%s): collectives through torch.distributed instead" % e, file=sys.stderr)
            native = False

    def step():
        if native:
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
        tm = BakeTimings()
        prod.dll.ommxGetLastBakeTimings(baker, C.byref(tm))
        tms.append(tm)
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

    # the SDK entry point proper (host arrays in, host arrays out) -- reported next to `value`, never as `value`
    host_ms, host_tm = None, None
    if rank == 0 and args.host_api_steps > 0:
        t1 = time.perf_counter()
        r, out = prod.bake_raw(baker, host_desc)
        assert r == ot.SUCCESS
        prod.fn("ommCpuDestroyBakeResult")(out)
        host_first_ms = (time.perf_counter() - t1) * 1e3   # cold: fresh result pages are faulted in
        t1 = time.perf_counter()
        for _ in range(args.host_api_steps):
            r, out = prod.bake_raw(baker, host_desc)
            assert r == ot.SUCCESS
            host_tm = BakeTimings()
            prod.dll.ommxGetLastBakeTimings(baker, C.byref(host_tm))
            prod.fn("ommCpuDestroyBakeResult")(out)
        host_ms = (time.perf_counter() - t1) / args.host_api_steps * 1e3

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        avg = lambda f: float(np.mean([getattr(t, f) for t in tms]))
        classify_ms = avg("classifyMs")
        launches = max(1, tms[-1].classifyLaunches)
        # algorithmic bytes of the classification launch (DESIGN.md "Roofline"): 0.25 B written per 4-state micro-triangle,
        # 24 B of UV read per work item, one pass over the texture and its summed-area table
        alg_bytes = 0.25 * tms[-1].microTriangles + 24.0 * tms[-1].uniqueItems + args.tex * args.tex * (1 + 4)
        achieved = alg_bytes / (classify_ms * 1e-3) / 1e9 if classify_ms > 0 else 0.0
        line = {
            "metric": "micro-triangles classified/sec (whole node)", "value": micro_tris / (elapsed / args.steps),
            "unit": "micro-triangles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d random-UV triangles (%.1f texels), %dx%d foliage-style UNORM8 alpha + SAT, subdiv level %d, 4-state, Wrap/Linear"
                                   % (args.tris, args.extent_texels, args.tex, args.tex, args.level),
                       "entry": (("ommxShardedBakeRccl (collectives issued by the library)" if native else "ommxSharded* + torch.distributed") if world > 1 else "ommxBakeDevice") + " (ommCpuBake contract, UV/index inputs and result arrays resident in HBM)", "sharding": "active work items partitioned over ranks; RCCL all-reduce of item metadata + chunked all-gather of OMM blocks" if world > 1 else "none",
                       "result": result_info, "unique_items": int(tms[-1].uniqueItems), "active_items": int(tms[-1].activeItems),
                       "fine_micro_triangles": int(tms[-1].fineMicroTriangles)},
            "bake_wall_time_ms": ms_per_step,
            # `value` is measured on the device-resident entry point (inputs and result arrays in HBM, the bench contract); its peer below
            # is the SDK call proper, host arrays in and out, i.e. the same bake plus one PCIe copy of the result
            "value_entry": "ommxBakeDevice" if world == 1 else ("ommxShardedBakeRccl" if native else "ommxSharded* + torch.distributed"),
            "rates": {"all_work_items": micro_tris / (elapsed / args.steps),
                      "active_items_only": float(tms[-1].activeItems) * 4.0 ** args.level / (classify_ms * 1e-3) if classify_ms > 0 else None,
                      "fine_pass_only": float(tms[-1].fineMicroTriangles) / (classify_ms * 1e-3) if classify_ms > 0 else None,
                      "note": "`value` counts 4^level micro-triangles for every unique work item like the reference's loop does; most items are settled by one summed-area-table "
                              "query (hierarchical culling), so the rate over the items that reach classify_tiles and over the micro-triangles that reach the level-line pass are given too"},
            "host_api": None if host_ms is None else {"entry": "ommCpuBake (host arrays in/out, PCIe inclusive)", "ms_per_bake": host_ms, "first_call_ms": host_first_ms,
                                                       "uploadMs": host_tm.uploadMs, "downloadMs": host_tm.downloadMs,
                                                       "micro_triangles_per_s": micro_tris / (host_ms * 1e-3)},
            "phases_ms": {k: avg(k) for k in ("uploadMs", "setupMs", "triageMs", "classifyMs", "digestMs", "tailMs", "gatherMs", "downloadMs", "totalMs")},
            "roofline": {"bound": "hbm", "kernel": "classify_tiles", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": classify_ms / launches,
                         "algorithmic_bytes_per_launch": alg_bytes / launches,
                         "note": "classification is fp32-VALU/sqrt/div bound, not HBM bound (SURVEY.md section 8d)"},
        }
        # HBM bytes per launch from the PMC counters cannot be read from inside this process: they come from the separate
        # rocprofv3 --pmc passes of profiles/collect.sh (FETCH_SIZE x2 on gfx950, WRITE_SIZE x1), committed under profiles/, and
        # apply only to the workload AND the library sources they were collected on: the summary carries a hash of omm_amd/csrc,
        # and a summary older than the sources is not printed.
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
        default_workload = (args.tris, args.level, args.tex, args.feature, args.extent_texels, args.seed) == (1000000, 8, 4096, 64, 8.0, 1234)
        if world == 1 and default_workload and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("source_sha256_16") == source_hash():
                line["roofline"]["traffic"] = tj.get("traffic_bytes_per_launch")
                line["roofline"]["traffic_source"] = "profiles/%s_pmc.md (separate rocprofv3 --pmc passes of `%s`; %s; sources %s)" % (
                    tj.get("tag"), tj.get("command"), tj.get("corrections"), tj.get("source_sha256_16"))
                # issue-slot view of the same profile (profiles/summarize_pmc.py: SQ instruction classes x measured cycles per class)
                line["roofline"]["issue"] = {"valu_issue_utilisation": tj.get("valu_issue_utilisation"), "scalar_issue_utilisation": tj.get("scalar_issue_utilisation"),
                                             "valu_lane_utilisation": tj.get("valu_lane_util"), "valu_instr_per_cycle_per_simd": tj.get("valu_instr_per_cycle_per_simd")}
            else:
                line["roofline"]["traffic_source"] = "none: profiles/hbm_traffic_latest.json was collected on other sources (%s, tree is %s)" % (tj.get("source_sha256_16"), source_hash())
        # the streaming part of the path for comparison: final gather of the surviving OMM blocks into arrayData order
        # (read + write of arrayData, HIP events around gather + index narrowing)
        gather_ms = avg("gatherMs")
        if gather_ms > 0 and world == 1:
            gb = 2.0 * result_info["arrayDataBytes"] + 8.0 * result_info["descs"] + 8.0 * result_info["triangles"]
            line["roofline_streaming"] = {"bound": "hbm", "kernel": "tail_gather_omms (+ narrow_indices)", "achieved": gb / (gather_ms * 1e-3) / 1e9,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / (gather_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and args.create_texture:
            line["create_texture_ms"] = {"entry": "ommCpuCreateTexture, UNORM8 + alphaCutoff (H2D + device summed-area table)", "4096": create_texture_ms(prod, baker, 4096, args.seed),
                                         "8192": create_texture_ms(prod, baker, 8192, args.seed)}
        if args.cpu_sample > 0 and world == 1:
            cb, cpu_res = cpu_baseline(args, tex, uv, ix)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline"] = line["value"] / cb["value"]
            # correctness gate of the same run: the HIP library bakes the CPU sample, byte-for-byte comparison
            k = min(args.cpu_sample, args.tris)
            gpu_res = prod.bake(baker, bake_desc(th, uv, ix, args, 0, k), want_stats=False)
            assert gpu_res.same_as(cpu_res), "GPU result differs from the CPU baseline on its sample: " + gpu_res.diff(cpu_res)
            line["parity_vs_cpu_baseline"] = "bit-exact on the CPU sample (arrayData %d B, %d descs, index buffer, histograms, index format)" % (gpu_res.array_data.size, len(gpu_res.descs))
            # second texture mode (no summed-area table: every micro-triangle takes the level-line pass), bounded samples on both sides
            if args.sat_off_sample > 0:
                ks = min(args.sat_off_sample, args.tris)
                th2 = prod.create_texture(baker, [tex], alpha_cutoff=-1.0)
                d2 = bake_desc(th2, uv, ix, args, 0, ks)
                prod.bake(baker, d2, want_stats=False)
                t2 = time.perf_counter()
                prod.bake(baker, d2, want_stats=False)
                gpu_dt = time.perf_counter() - t2
                kc = min(max(args.sat_off_sample // 20, 200), ks)
                cb2, cpu_res2 = cpu_baseline(args, tex, uv, ix, sat=False, sample=kc)
                gpu_res2 = prod.bake(baker, bake_desc(th2, uv, ix, args, 0, kc), want_stats=False)
                assert gpu_res2.same_as(cpu_res2), "SAT-off GPU result differs from the CPU baseline: " + gpu_res2.diff(cpu_res2)
                prod.destroy_texture(baker, th2)
                line["sat_off"] = {"entry": "ommCpuBake (host arrays, PCIe inclusive), texture without alphaCutoff", "gpu_micro_triangles_per_s": ks * 4.0 ** args.level / gpu_dt,
                                   "gpu_sample": "first %d triangles, %.1f ms" % (ks, gpu_dt * 1e3), "cpu_baseline": cb2, "parity": "bit-exact on the CPU sample"}
        print(json.dumps(line))
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()   # rank 0 has done more (host-API bakes, the JSON line): all ranks tear their communicators down together
    prod.destroy_texture(baker, th)
    prod.destroy_baker(baker)
    if comm is not None:
        prod.dll.ommxRcclCommDestroy(comm)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
