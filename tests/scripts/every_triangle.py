"""EVERY triangle of a BASELINE configuration against the oracle (not part of the suite: minutes of CPU).

The reference algorithm -- and so the oracle -- keeps 2 * 4^N bytes per work item (131 GB for configs[2] as one bake), so the configuration's seeded
triangle stream is cut into slices of SLICE triangles; each slice is baked through ommCpuBake of the HIP library and by the oracle, and the two results are
compared in full (arrayData, descriptors, index buffer, histograms, index format).  That covers the classification of every triangle of the workload;
what a slice does not cover -- dedup, ordering and offsets across the WHOLE workload -- is covered by the full-size tests of the suite (invariants,
first / last triangles, sha256 across rank counts).

usage (GPU box): python tests/scripts/every_triangle.py CONFIG [TRIS [SLICE [FIRST_SLICE [SLICES]]]]     e.g.  every_triangle.py c2 1000000 50000
prints one line per slice and a summary; exit status 1 on a mismatch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import ommtest as ot
import workloads as wl
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
tris = int(sys.argv[2]) if len(sys.argv) > 2 else bench.CONFIGS[cfg]["tris"]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
count = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 30

tex, uv, ix, lv, kw = wl.workload(cfg, tris)
tris = ix.size // 3
product, oracle = ot.Lib("product"), ot.Lib("oracle")
hw, eff, host = bench.host_info()
oracle.dll.oracle_ommxSetThreads(eff)
pb = product.create_baker(); pt = product.create_texture(pb, [tex], alpha_cutoff=0.5)
ob = oracle.create_baker(); otx = oracle.create_texture(ob, [tex], alpha_cutoff=0.5)
bad, done, micro, t_gpu, t_cpu, nbytes = [], 0, 0.0, 0.0, 0.0, 0
slices = [(a, min(tris, a + per)) for a in range(0, tris, per)][first:first + count]
for a, b in slices:
    suv, six, slv = wl.subset(uv, ix, lv, a, b)
    t0 = time.time()
    got = product.bake(pb, bench.desc_for(pt, suv, six, slv, kw), want_stats=False)
    t1 = time.time()
    want = oracle.bake(ob, bench.desc_for(otx, suv, six, slv, kw), want_stats=False)
    t2 = time.time()
    same = got.same_as(want)
    mt = bench.micro_triangles_of(product, pb, None)
    done += b - a; micro += mt; t_gpu += t1 - t0; t_cpu += t2 - t1; nbytes += int(got.array_data.size)
    print("triangles %8d..%8d: %s  (arrayData %d B, %d descs; product %.3f s, oracle %.1f s)" % (a, b, "bit-exact" if same else "MISMATCH " + str(got.diff(want))[:300],
                                                                                             got.array_data.size, len(got.descs), t1 - t0, t2 - t1), flush=True)
    if not same:
        bad.append((a, b))
print("%s: %d of %d triangles in %d slices (%.3e micro-triangles, %.2f GB of arrayData) compared in full with the oracle at %d threads: %d mismatching slices %s; "
      "product %.1f s, oracle %.1f s" % (cfg, done, tris, len(slices), micro, nbytes / 1e9, eff, len(bad), bad, t_gpu, t_cpu))
sys.exit(1 if bad else 0)
