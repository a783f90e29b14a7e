"""The oracle against support/tests/test_omm_log.cpp (exact log strings, incl. the workload figure 137972015 of
ValidateWorkloadSize, bake_cpu_impl.cpp:662-713, for the reference's own 511-triangle fixture)."""
import log_cases


def test_oracle_log_cases(oracle):
    log_cases.run_log_cases(oracle)


def test_oracle_basic_cases(oracle):
    """support/tests/test_basic.cpp"""
    import basic_cases
    basic_cases.run_basic_cases(oracle)
