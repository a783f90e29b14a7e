"""Committed oracle-output fixtures for the BASELINE.json configurations (tests/golden/oracle_fixtures.json, made by
tests/golden/make_oracle_fixtures.py): XXH64 of every result array + counts + ommDebugGetStats2.

CPU: the oracle still reproduces the cheap cases (and, for C0, the answers of the reference's own translation units recorded in SURVEY.md 8c).
GPU: the HIP library reproduces ALL of them -- a comparison with committed answers, independent of an oracle run on the same machine."""
import json
import os
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ommtest as ot                      # noqa: E402
import make_oracle_fixtures as mk         # noqa: E402

FIXTURES = json.load(open(os.path.join(HERE, "golden", "oracle_fixtures.json")))


def check(lib, fx):
    res = mk.bake(lib, dict(fx["case"]))
    got, want = mk.digest(res), fx["expect"]
    for k in want:
        if k != "reference_answer":
            assert got[k] == want[k], (fx["case"]["name"], k, got[k], want[k])
    ref = want.get("reference_answer", {})
    if "arrayData_hex" in ref:
        assert res.array_data.tobytes().hex() == ref["arrayData_hex"] and list(res.index) == ref["index"]
    if "arrayData_xxh64_seed0" in ref:
        assert got["arrayData"] == ref["arrayData_xxh64_seed0"]


def test_fixture_file_covers_the_baseline_configs():
    names = {f["case"]["name"] for f in FIXTURES}
    assert {"C0_2state", "C0_4state", "C1_full_100k", "C2_replica_20k", "C4_replica_2k"} <= names
    c0 = {f["case"]["name"]: f["expect"] for f in FIXTURES}
    assert c0["C0_2state"]["reference_answer"]["arrayData_hex"] == "fc" + "ff" * 31
    assert c0["C0_4state"]["reference_answer"]["arrayData_xxh64_seed0"] == "1055f30807d9815f"


@pytest.mark.parametrize("fx", [f for f in FIXTURES if f["case"]["cpu"]], ids=lambda f: f["case"]["name"])
def test_oracle_reproduces_fixture(fx):
    check(ot.Lib("oracle"), fx)


@pytest.mark.gpu
@pytest.mark.parametrize("fx", FIXTURES, ids=lambda f: f["case"]["name"])
def test_hip_library_reproduces_fixture(fx):
    check(ot.Lib("product"), fx)


# ---- every triangle of the BASELINE configurations, slice by slice (tests/golden/slice_fixtures.json, made by tests/golden/make_slice_fixtures.py) ----
import make_slice_fixtures as ms          # noqa: E402

SLICES = ms.load()


def test_slice_fixture_file_covers_every_triangle_of_the_metric_configuration():
    """all 20 slices of configs[2] (1 M triangles: the configuration BASELINE's metric is quoted on), all 4 of the asset-shaped cards, and all 160 of configs[4]
    (4 M triangles, per-triangle levels 4 - 10 and dynamic levels; 45 minutes of the container's 8 cores)"""
    assert [(e["first"], e["end"]) for e in SLICES["c2"]["slices"]] == ms.slices_of("c2")
    assert [(e["first"], e["end"]) for e in SLICES["cards"]["slices"]] == ms.slices_of("cards")
    assert [(e["first"], e["end"]) for e in SLICES["c4"]["slices"]] == ms.slices_of("c4")
    assert sum(e["end"] - e["first"] for e in SLICES["c2"]["slices"]) == 1000000


def _check_slices(lib, cfg, entries):
    s = ms.Slicer(lib, cfg)
    bad = []
    for e in entries:
        got, want = ms.digest(s.bake(e["first"], e["end"])), e["expect"]
        bad += [(cfg, e["first"], k, got[k], want[k]) for k in want if got[k] != want[k]]
    s.close()
    assert not bad, bad[:6]


def test_oracle_reproduces_a_slice_of_the_cards():
    """the generator's answers are the oracle's (cheap case on the CPU: the last 10 000 of the asset-shaped quads)"""
    _check_slices(ot.Lib("oracle"), "cards", SLICES["cards"]["slices"][-1:])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", sorted(SLICES))
def test_hip_library_reproduces_every_slice(cfg):
    """ommCpuBake on every committed slice of the configuration against the oracle's committed digests: for configs[2] that is every one of its 1 000 000
    triangles (6.55e10 micro-triangles, 1.3 GB of arrayData), compared with answers computed in the build container, not with an oracle run next to it"""
    _check_slices(ot.Lib("product"), cfg, SLICES[cfg]["slices"])
