"""Pins the CPU oracle against every known-answer test the reference holds for the ommCpuBake path
(/root/reference/support/tests/test_omm_bake_cpu.cpp), in all six suite configurations."""
import pytest
import kat_runner as kr
from kat_cases import CASES, CONFIGS, LEAFLET_MIP, LEAFLET_LEVEL
import ommtest as ot

FAST_CONFIGS = ["Default", "AlphaCutoff"]


def _configs(c):
    # the full 6-config sweep for everything cheap; slow cases run in the two configs that differ in algorithm
    return FAST_CONFIGS if c["slow"] else list(CONFIGS)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_kat(oracle, case):
    b = oracle.create_baker()
    for cfg in _configs(case):
        res = kr.run_case(oracle, b, case, cfg)
        assert res.stats_tuple() == case["expect"], (cfg, "reference line %d" % case["ref"])
    oracle.destroy_baker(b)


@pytest.mark.parametrize("name,ref,mip_start,num_mip,cutoff,expect", LEAFLET_MIP, ids=[c[0] for c in LEAFLET_MIP])
def test_leaflet_mip(oracle, name, ref, mip_start, num_mip, cutoff, expect):
    b = oracle.create_baker()
    for cfg in CONFIGS:
        assert kr.run_leaflet_mip(oracle, b, mip_start, num_mip, cutoff, cfg).stats_tuple() == expect, cfg
    oracle.destroy_baker(b)


@pytest.mark.parametrize("name,ref,level,expect", LEAFLET_LEVEL, ids=[c[0] for c in LEAFLET_LEVEL])
def test_leaflet_level(oracle, name, ref, level, expect):
    b = oracle.create_baker()
    for cfg in CONFIGS:
        assert kr.run_leaflet_level(oracle, b, level, cfg).stats_tuple() == expect, cfg
    oracle.destroy_baker(b)


def test_leaflet_level12_workload_too_big(oracle):
    # test_omm_bake_cpu.cpp:2021-2031
    b = oracle.create_baker()
    assert kr.run_leaflet_level(oracle, b, 12, "Default", max_workload=512, expect=ot.WORKLOAD_TOO_BIG) is None
    oracle.destroy_baker(b)
