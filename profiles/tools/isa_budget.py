#!/usr/bin/env python3
"""Instruction budget of the per-micro-triangle pieces of classify_tiles, by category, from the ISA of profiles/tools/isa_budget.hip (gfx950, the product's flags).
usage: python profiles/tools/isa_budget.py > profiles/r04_isa_budget.md      (needs hipcc; no GPU)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --cuda-device-only -S".split()
CATS = [("fp32 add / sub", r"v_(add|sub|subrev)_f32"), ("fp32 mul", r"v_mul_f32"), ("fp32 fma / mad", r"v_(fma|mad|fmac|mac)_f32"),
        ("fp32 min / max / med", r"v_(min|max|med)3?_f32|v_(min|max)_f32"), ("rcp / sqrt / div helpers", r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp|trig)_"),
        ("fp64", r"_f64"), ("conversions / floor / ceil / fract", r"v_cvt_|v_(floor|ceil|trunc|rndne|fract)_"),
        ("compares", r"v_cmp|v_cmpx"), ("selects / moves / lane moves", r"v_cndmask|v_mov_|v_readlane|v_readfirstlane|v_writelane|v_accvgpr|v_swap|v_perm|v_bfi|v_bfe|v_alignbit|v_mbcnt"),
        ("integer / bit logic", r"v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|not|bcnt|ffb|min|max|lshlrev|lshrrev|ashrrev|add3|or3|and_or|lshl_add|lshl_or|xad|addc|subb|mul_lo|mul_hi|mad_u|mad_i)[_a-z0-9]*(u32|i32|b32|u16|i16|b16|u64|i64|b64|u24|i24|co)"),
        ("LDS", r"^ds_"), ("global / scratch memory", r"^(global|flat|buffer|scratch)_"), ("scalar memory", r"^s_(load|buffer_load|store)"),
        ("branches", r"^s_(cbranch|branch|setpc|call)"), ("waits / barriers / nops", r"^s_(waitcnt|barrier|nop|sleep|endpgm)"), ("other scalar ALU", r"^s_")]
def classify(op):
    for name, rx in CATS:
        if re.search(rx, op): return name
    return "other vector" if op.startswith("v_") else "other"
def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "b.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(ROOT, "profiles", "tools", "isa_budget.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    kernels = collections.OrderedDict()
    for m in re.finditer(r"^(k\d_\w+):\s*;.*?\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        cnt = collections.Counter()
        for line in m.group(2).splitlines():
            line = line.split(";")[0].strip()
            if not line or line.endswith(":") or line.startswith("."): continue
            cnt[classify(line.split()[0])] += 1
        kernels[m.group(1)] = cnt
    names = list(kernels)
    frame = kernels.get("k0_frame_decode_vertices", collections.Counter())
    print("# ISA instruction budget of the per-micro-triangle pieces of `classify_tiles` (gfx950; static counts of isolated kernels, `profiles/tools/isa_budget.hip`)\n")
    print("Every kernel is one call of the device function per lane on top of the same frame (LDS set-up, split bird decode, the three interpolated vertices: column k0);")
    print("the other columns are given MINUS that frame.  Both sides of data-dependent branches are counted, so branchy pieces (edge tests, region test) are upper bounds.\n")
    print("| category | " + " | ".join(n.split("_", 1)[1].replace("_", " ") for n in names) + " |")
    print("|---|" + "---|" * len(names))
    allcats = [c for c, _ in CATS] + ["other vector", "other"]
    tot = {n: 0 for n in names}; vtot = {n: 0 for n in names}
    for c in allcats:
        row = []
        for n in names:
            v = kernels[n][c] - (frame[c] if n != names[0] and n[:2] in ("k1", "k2", "k3") else 0)
            row.append(v); tot[n] += v
            if c not in ("LDS", "global / scratch memory", "scalar memory", "branches", "waits / barriers / nops", "other scalar ALU", "other"): vtot[n] += v
        if any(row): print("| %s | " % c + " | ".join(str(v) for v in row) + " |")
    print("| **vector ALU** | " + " | ".join("**%d**" % vtot[n] for n in names) + " |")
    print("| **all** | " + " | ".join("**%d**" % tot[n] for n in names) + " |")
if __name__ == "__main__":
    main()
