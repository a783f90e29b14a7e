"""omm_amd -- MI355X-native opacity-micromap baker behind the OMM SDK's ommCpuBake() C ABI.

The product is the shared library ``omm_amd/lib/libomm-lib.so`` (sources in ``omm_amd/csrc``); this package only
tells Python callers where it is.  There is no Python or CPU implementation of the bake in here.
"""
import os

LIBRARY_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libomm-lib.so")


def library_path():
    if not os.path.exists(LIBRARY_PATH):
        raise RuntimeError("libomm-lib.so has not been built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    return LIBRARY_PATH
