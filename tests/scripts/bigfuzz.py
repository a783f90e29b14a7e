"""Long randomized differential run (not part of the suite): `python tests/scripts/bigfuzz.py LO HI` bakes the fuzz cases LO..HI-1 of
tests/test_gpu_parity.py::_fuzz_case with the HIP library and the oracle and compares the full results.
Round 1: seeds 1000..3999 -> 0 mismatches (15 min, dominated by the CPU oracle).  Round 3: the knobs vary with the seed (see below).
Round 5: ommCpuBake over 2 / 3 devices of the process (ranks sharing the GPU) varies with the seed too; `tests/scripts/r05_bigfuzz.sh` runs four ranges side by side."""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, ommtest as ot
import test_gpu_parity as T
product, oracle = ot.Lib("product"), ot.Lib("oracle")
bad = []
lo, hi = int(sys.argv[1]), int(sys.argv[2])
walks = len(sys.argv) > 3 and sys.argv[3] == "walks"   # (round 5) `LO HI walks`: the cases of test_fuzz_deferred_texel_walks, generic pass deferred / in the launch / automatic
for seed in range(lo, hi):
    mips, uv, ix, level, cutoff, sat, kw = T._walk_case(seed) if walks else T._fuzz_case(seed)
    # (round 3) the per-baker knobs vary with the seed too: streamed result of ommCpuBake forced with 1 / 3 / 5 ranges or left alone, generic texel-loop
    # path inside the persistent launch / deferred / automatic
    # (round 5) ... and the multi-device form of ommCpuBake (ommxBakerKnob_Devices) in a quarter of the cases
    knobs = [(ot.KNOB_STREAM_CHUNKS, (0, 1, 3, 5)[seed % 4]), (ot.KNOB_GENERIC_PASS, (seed // 4) % 3), (ot.KNOB_DEVICES, (0, 0, 2, 0, 0, 3, 0, 0)[(seed // 12) % 8])]
    if walks:
        knobs = [(ot.KNOB_GENERIC_PASS, (2, 2, 1, 0)[seed % 4]), (ot.KNOB_DEVICES, (0, 0, 0, 2)[(seed // 4) % 4])]
    try:
        T.both(product, oracle, mips, uv, ix, level, sat=sat, cutoff=cutoff, knobs=knobs, **kw)
    except Exception as e:   # noqa: BLE001  (a failed bake counts like a differing one)
        bad.append((seed, str(e)[:200]))
        print("MISMATCH seed", seed, type(e).__name__, str(e)[:300], flush=True)
    if (seed - lo) % 100 == 99:
        print("progress: seeds %d..%d compared, %d mismatches" % (lo, seed + 1, len(bad)), flush=True)
print("seeds %d..%d done, %d mismatches" % (lo, hi, len(bad)))
