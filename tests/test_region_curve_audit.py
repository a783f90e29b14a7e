"""Audit of the curve-free-region test (omm_amd/csrc/region_curve.h): the HIP kernels settle a whole bird-curve sub-triangle -- a work item, a tile of 4096
micro-triangles, a group of 64 -- without classifying its micro-triangles when the level curve alpha == cutoff provably cannot reach it.  The audit build
of the oracle includes that header as it is, evaluates the test for EVERY sub-triangle of EVERY level of every work item it bakes (the hierarchy levels the
kernels use, and all the others) and compares a verdict "all descendants have the pure state s" with the states the reference algorithm produced for them.
counters (oracle/omm_oracle.c: audit_region_item): [0] sub-triangles tested, [1] settled, [2] micro-triangles under settled sub-triangles, [3] micro-triangles
whose reference state differs (must be 0), [4] / [5] work items whose shape admits the test / all, [6] / [7] = [1] / [2] for sub-triangles whose cells hold
texels on both sides of the cutoff (what the summed-area table cannot settle); work items without the corner bound (thin ones, or micro-triangles smaller
than the rounding of their vertices: RcShape::fat == 0) get the weaker verdict "pure state unless PointInTriangle puts a wrong-side cell corner inside":
[8] sub-triangles with that verdict, [9] descendants it covers, [10] descendants left to the full pass (a wrong-side corner inside), [11] covered ones whose
reference state differs (must be 0)."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
import ommtest as ot
import workloads as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def audit():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libomm_oracle_audit.so"])
    saved = ot.oracle_path
    ot.oracle_path = lambda: os.path.join(ROOT, "oracle", "libomm_oracle_audit.so")
    try:
        lib = ot.Lib("oracle")
    finally:
        ot.oracle_path = saved
    lib.dll.orc_audit_region_counter.restype = C.c_longlong
    lib.dll.orc_audit_region_counter.argtypes = [C.c_int]
    return lib


def counters(lib):
    return [lib.dll.orc_audit_region_counter(i) for i in range(12)]


def test_region_verdicts_on_the_bench_workloads(audit):
    """the first triangles of the BASELINE configurations' own streams, with and without the summed-area table"""
    audit.dll.orc_audit_region_reset()
    for kind, n in (("c2", 120), ("c1", 1200), ("c4", 700), ("cards", 40)):
        tex, uv, ix, lv, kw = wl.workload(kind, n)
        k = dict(kw); level = k.pop("level")
        b = audit.create_baker()
        for cutoff in ((0.5, -1.0) if kind == "c2" else (0.5,)):
            t = audit.create_texture(b, [tex], alpha_cutoff=cutoff)
            audit.bake(b, ot.make_desc(t, uv, ix, level, levels=lv, filt=ot.LINEAR, flags=ot.FLAG_THREADS, **k), want_stats=False)
            audit.destroy_texture(b, t)
        audit.destroy_baker(b)
    c = counters(audit)
    assert c[0] > 10_000_000 and c[4] > 1000, c
    assert c[6] > 100_000 and c[7] > 1_000_000, c      # settled where the texels around the sub-triangle are NOT all on one side
    assert c[8] > 1_000_000 and c[9] > 10_000_000, c   # the weaker verdict (no corner bound): configs[4]'s level-9 / 10 items are all of that kind
    assert c[3] == 0 and c[11] == 0, c


def test_region_verdicts_on_adversarial_inputs(audit):
    """textures that stress the error bounds, thin triangles, nearly vertical / horizontal edges, UV offsets, every address mode, 2-state and every promotion
    (tests/region_cases.py; the GPU suite bakes the same cases on the device and compares the arrays with the oracle)"""
    import region_cases
    audit.dll.orc_audit_region_reset()
    region_cases.run(audit)
    c = counters(audit)
    assert c[0] > 500_000 and c[1] > 400_000 and c[6] > 2_000, tuple(c)
    assert c[3] == 0 and c[11] == 0, c


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [101, 202, 303])
def test_region_verdicts_seed_sweep_on_every_lease(audit, seed):
    """The same audit with other seeds (textures' noise, triangle streams), as part of the GPU suite: every lease of a GPU box re-audits the constants of
    region_curve.h on inputs the CPU suite has not seen (pure CPU work, a few seconds per seed; marked gpu only for WHERE it runs)."""
    import region_cases
    audit.dll.orc_audit_region_reset()
    region_cases.run(audit, seed=seed, tri_seed_base=7000 + seed)
    c = counters(audit)
    assert c[0] > 300_000 and c[1] > 200_000, tuple(c)
    assert c[3] == 0 and c[11] == 0, c
