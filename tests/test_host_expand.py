"""Host side of ommCpuBake's compressed result (omm_amd/csrc/host_expand.cpp): the worker pool and the expansion of a codec stream into the caller's
arrayData, against a scalar model of the device codec (tests/native/expand_check.cpp).  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_codec_expansion_and_worker_pool(tmp_path):
    exe = str(tmp_path / "expand_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "native", "expand_check.cpp"),
                           os.path.join(ROOT, "omm_amd", "csrc", "host_expand.cpp")])
    out = subprocess.check_output([exe], text=True, timeout=300)
    assert out.startswith("ok 288 scatter 24 zero-ahead 18"), out
