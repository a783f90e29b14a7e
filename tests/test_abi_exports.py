"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and exports every symbol that
include/*.h declares; struct layouts match the SDK's (sizes the reference itself asserts)."""
import ctypes as C
import os
import re
import pytest
import ommtest as ot

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in ("omm_mi355x.h", "omm_mi355x_ext.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        names += re.findall(r"OMM_MI355X_API\s+[\w\s\*]+?\b(omm\w+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    path = ot.product_path()
    if not os.path.exists(path):
        pytest.fail("libomm-lib.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    dll = C.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 25, names
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, missing
    dll.ommGetLibraryDesc.restype = C.c_uint32  # 3 bytes in a register
    v = dll.ommGetLibraryDesc()
    assert (v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff) == (1, 9, 0)


def test_struct_layouts():
    assert C.sizeof(ot.BakeInputDesc) == 136          # serialize_impl.cpp:86
    assert C.sizeof(ot.BakeResultDesc) == 80
    assert C.sizeof(ot.MicromapDesc) == 8 and C.sizeof(ot.UsageCount) == 8
    assert C.sizeof(ot.TextureDesc) == 24 and C.sizeof(ot.TextureMipDesc) == 24
    assert ot.BakeInputDesc.maxWorkloadSize.offset == 128 and ot.BakeInputDesc.subdivisionLevels.offset == 120
    assert ot.BakeInputDesc.maxSubdivisionLevel.offset == 112 and ot.BakeInputDesc.maxArrayDataSize.offset == 116


def test_host_side_argument_checks_need_no_gpu():
    """entry points reject bad handles before touching the device (bake.cpp:36-135)"""
    dll = C.CDLL(ot.product_path())
    assert dll.ommCreateBaker(None, None) == ot.INVALID_ARGUMENT
    assert dll.ommDestroyBaker(None) == ot.INVALID_ARGUMENT
    assert dll.ommCpuBake(None, None, None) == ot.INVALID_ARGUMENT
    assert dll.ommCpuDestroyBakeResult(None) == ot.INVALID_ARGUMENT
    assert dll.ommCpuGetBakeResultDesc(None, None) == ot.INVALID_ARGUMENT
    lib = ot.Lib("product")
    b = lib.create_baker(baker_type=0)               # ommBakerType_GPU bakers can be created but do not bake (test_basic.cpp:46-51)
    d = ot.default_bake_desc()
    assert lib.bake_raw(b, d)[0] == ot.INVALID_ARGUMENT
    assert lib.destroy_baker(b) == ot.SUCCESS
    b = lib.create_baker()
    assert lib.bake_raw(b, d)[0] == ot.INVALID_ARGUMENT   # no texture set
    assert lib.destroy_baker(b) == ot.SUCCESS


def test_no_cpu_fallback_without_a_gpu():
    """the product has no CPU path.  On a machine without a HIP device texture creation (the first call that
    needs HBM) fails loudly -- FAILURE + a Fatal log line -- instead of computing anything on the host."""
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int(0)
    if hip.hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a HIP device is present: the fail-loudly path cannot be exercised here")
    import numpy as np
    lib = ot.Lib("product")
    msgs = []
    b = lib.create_baker(callback=lambda sev, msg, user: msgs.append((sev, msg.decode())))
    t = lib.create_texture(b, [np.zeros((16, 16), np.float32)], expect=ot.FAILURE)
    assert t is None
    assert msgs and msgs[-1][0] == 3 and "no CPU fallback" in msgs[-1][1], msgs       # ommMessageSeverity_Fatal
    assert lib.destroy_baker(b) == ot.SUCCESS


def _build_example(std_args, out):
    import subprocess
    lib_dir = os.path.join(ROOT, "omm_amd", "lib")
    cmd = std_args + ["-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "minimal_sample.c"), "-o", out,
                      "-L" + lib_dir, "-lomm-lib", "-lm", "-Wl,-rpath," + lib_dir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr


def test_header_is_valid_c99_and_cpp17_and_the_example_links(tmp_path):
    """include/omm_mi355x.h is what a C or C++ caller compiles against: examples/minimal_sample.c builds warning-free in both languages
    and links against the drop-in library"""
    _build_example(["gcc", "-std=c99", "-pedantic"], str(tmp_path / "sample_c"))
    _build_example(["g++", "-std=c++17", "-x", "c++"], str(tmp_path / "sample_cpp"))
