#!/usr/bin/env python3
"""Extracts the reference's own golden test DATA into small committed fixtures.

Run in the build container (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_fixtures.py

Outputs (data only -- inputs and expected outputs, no reference source text):
  blobs.json      the serialized input/output byte blobs embedded in
                  support/tests/test_omm_bake_cpu.cpp:2034-2304 (8x8 StandardCircle, level 4),
                  hex-encoded, keyed by the variable name used there.
  leaflet_b.bin   channel 2 of assets/tests/leaflet.png as raw bytes, row-major
                  (the reference tests read exactly this channel: test_omm_bake_cpu.cpp:662-669).
  leaflet.json    its dimensions.
  texcoord_kat.json  the GetTexCoord input/expected tables of support/tests/test_texture.cpp:40-266.
  sdk_exports.txt    the names of the 25 OMM_API functions libraries/omm-lib/include/omm.h declares (= the SDK library's export list).
"""
import json, os, re, sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def extract_blobs():
    src = open(os.path.join(REF, "support/tests/test_omm_bake_cpu.cpp")).read()
    blobs = {}
    for m in re.finditer(r"std::vector<unsigned char>\s+(\w+)\s*=\s*\{(.*?)\};", src, re.S):
        name, body = m.group(1), m.group(2)
        data = bytes(int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", body))
        blobs[name] = data.hex()
    return blobs


def extract_sdk_exports():
    src = open(os.path.join(REF, "libraries/omm-lib/include/omm.h")).read()
    return sorted(set(re.findall(r"OMM_API\s+\w+\s+(?:OMM_CALL\s+)?(omm\w+)\s*\(", src)))


def extract_texcoord_kats():
    """(mode, [x,y], [w,h], [ex,ey]) tuples asserted by support/tests/test_texture.cpp:40-266."""
    src = open(os.path.join(REF, "support/tests/test_texture.cpp")).read()
    pat = re.compile(r"TexCoordTest\(omm::TextureAddressMode::(\w+),\s*\{\s*(-?\d+),\s*(-?\d+)\s*,?\s*\},\s*\{\s*(-?\d+),\s*(-?\d+)\s*,?\s*\},\s*\{\s*(-?\d+),\s*(-?\d+)\s*,?\s*\}\)")
    return [[m.group(1)] + [int(m.group(i)) for i in range(2, 8)] for m in pat.finditer(src)]


def extract_leaflet():
    from PIL import Image
    im = Image.open(os.path.join(REF, "assets/tests/leaflet.png"))
    w, h = im.size
    bands = im.getbands()
    assert len(bands) >= 3, bands
    data = im.tobytes()
    ch = len(bands)
    blue = bytes(data[2::ch])
    assert len(blue) == w * h
    return w, h, ch, blue


if __name__ == "__main__":
    blobs = extract_blobs()
    json.dump(blobs, open(os.path.join(HERE, "blobs.json"), "w"), indent=0, sort_keys=True)
    print("blobs:", {k: len(v) // 2 for k, v in blobs.items()})
    tk = extract_texcoord_kats()
    json.dump(tk, open(os.path.join(HERE, "texcoord_kat.json"), "w"))
    print("texcoord kats:", len(tk))
    w, h, ch, blue = extract_leaflet()
    open(os.path.join(HERE, "leaflet_b.bin"), "wb").write(blue)
    json.dump({"width": w, "height": h, "channels": ch, "channel": 2}, open(os.path.join(HERE, "leaflet.json"), "w"))
    print("leaflet:", w, h, ch)
    ex = extract_sdk_exports()
    open(os.path.join(HERE, "sdk_exports.txt"), "w").write("\n".join(ex) + "\n")
    print("sdk exports:", len(ex))


def make_abi_layout():
    """tests/golden/abi_layout_sdk.txt: output of tests/native/abi_layout.c compiled against the SDK's own omm.h (146 sizeof / offsetof /
    enumerator facts of the CPU-baker interface).  tests/test_abi_exports.py compares include/omm_mi355x.h against it."""
    import subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "abi_ref")
        subprocess.check_call(["g++", "-std=c++17", "-x", "c++", "-Wno-deprecated-declarations", "-I/root/reference/libraries/omm-lib/include",
                               "-DOMM_HEADER=<omm.h>", os.path.join(root, "tests", "native", "abi_layout.c"), "-o", exe])
        open(os.path.join(root, "tests", "golden", "abi_layout_sdk.txt"), "w").write(subprocess.check_output([exe], text=True))
