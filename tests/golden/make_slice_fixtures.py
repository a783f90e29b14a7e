#!/usr/bin/env python3
"""Per-slice oracle answers for EVERY triangle of the BASELINE configurations (review item: "turn the every-triangle differential into driver-run evidence").

    python tests/golden/make_slice_fixtures.py c2            (build container: runs the CPU oracle only; c2 ~7 min, cards ~1 min, c4 ~1 h on 8 cores)
    python tests/golden/make_slice_fixtures.py c4 8          (every 8th slice of the configuration)

The reference algorithm -- and so the oracle -- keeps 2 x 4^N bytes per work item, so a configuration's seeded triangle stream (tests/workloads.py) is cut into
slices the oracle's memory fits; every slice is one complete bake (reference semantics: src/bake_cpu_impl.cpp:1923-1985).  For each slice the script records
the sizes and XXH64(seed 0) of arrayData / descArray / indexBuffer, the index format and both histograms into tests/golden/slice_fixtures.json (merged per
configuration, so configurations can be generated one at a time).  tests/test_oracle_fixtures.py::test_hip_library_reproduces_every_slice bakes the same slices
through ommCpuBake on the GPU box and compares with these committed answers -- no oracle runs next to it there.  What a slice does not cover (dedup, order and
offsets across the WHOLE workload) the full-size tests of the suite do."""
import json, os, sys, time
import numpy as np
import xxhash

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ommtest as ot  # noqa: E402
import workloads as wl  # noqa: E402

PATH = os.path.join(HERE, "slice_fixtures.json")
# configuration -> (triangles, triangles per slice): the slices of tests/scripts/every_triangle.py
PLAN = {"c2": (1000000, 50000), "cards": (40000, 10000), "c4": (4000000, 25000), "c1": (100000, 50000)}


def slices_of(cfg, every=1):
    tris, per = PLAN[cfg]
    return [(a, min(tris, a + per)) for a in range(0, tris, per)][::every]


def digest(res):
    x = lambda a: xxhash.xxh64(np.ascontiguousarray(a).tobytes(), seed=0).hexdigest()
    return {"arrayDataSize": int(res.array_data.size), "descArrayCount": int(len(res.descs)), "indexCount": int(res.index.size), "indexFormat": int(res.index_format),
            "arrayData": x(res.array_data), "descArray": xxhash.xxh64(res.desc_bytes, seed=0).hexdigest(), "indexBuffer": x(res.index),
            "descArrayHistogram": [list(map(int, h)) for h in res.array_hist], "indexHistogram": [list(map(int, h)) for h in res.index_hist]}


class Slicer:
    """one baker + texture of a library for a configuration; bake(a, b) = the result of triangles [a, b) of the seeded stream"""
    def __init__(self, lib, cfg):
        import bench
        self.lib, self.bench = lib, bench
        self.tex, self.uv, self.ix, self.lv, self.kw = wl.workload(cfg, PLAN[cfg][0])
        self.baker = lib.create_baker()
        self.texture = lib.create_texture(self.baker, [self.tex], alpha_cutoff=0.5)

    def bake(self, a, b):
        suv, six, slv = wl.subset(self.uv, self.ix, self.lv, a, b)
        return self.lib.bake(self.baker, self.bench.desc_for(self.texture, suv, six, slv, self.kw), want_stats=False)

    def close(self):
        self.lib.destroy_texture(self.baker, self.texture)
        self.lib.destroy_baker(self.baker)


def load():
    return json.load(open(PATH)) if os.path.exists(PATH) else {}


if __name__ == "__main__":
    cfg = sys.argv[1]
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    orc = ot.Lib("oracle")
    orc.dll.oracle_ommxSetThreads(int(os.environ.get("OMM_ORACLE_THREADS", str(os.cpu_count() or 1))))
    s = Slicer(orc, cfg)
    out = load()
    have = {(e["first"], e["end"]): e for e in out.get(cfg, {}).get("slices", [])}
    t00 = time.time()
    for a, b in slices_of(cfg, every):
        if (a, b) in have:
            continue
        t0 = time.time()
        have[(a, b)] = {"first": a, "end": b, "expect": digest(s.bake(a, b))}
        print("%s %8d..%8d  arrayData %d B  %s  (%.1f s)" % (cfg, a, b, have[(a, b)]["expect"]["arrayDataSize"], have[(a, b)]["expect"]["arrayData"], time.time() - t0), flush=True)
        out[cfg] = {"triangles": PLAN[cfg][0], "per_slice": PLAN[cfg][1], "workload": "tests/workloads.py: workload(%r, %d), subset(first, end)" % (cfg, PLAN[cfg][0]),
                    "slices": [have[k] for k in sorted(have)]}
        json.dump(out, open(PATH + ".tmp", "w"), indent=0)
        os.replace(PATH + ".tmp", PATH)
    s.close()
    print("%s: %d slices in %.0f s" % (cfg, len(have), time.time() - t00))
