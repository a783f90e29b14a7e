"""One-off stress of the single-pass scans and the two tail paths (not part of the suite): test_descriptor_order_and_offsets_both_tail_paths at tile-boundary and large counts.
usage (GPU box): python tests/scripts/stress_tail_paths.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import ommtest as ot
import test_gpu_parity as tg

product, oracle = ot.Lib("product"), ot.Lib("oracle")
flags_list = [ot.FLAG_THREADS, ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL, ot.FLAG_THREADS | ot.FLAG_NO_SPECIAL | ot.FLAG_NO_DEDUP | ot.FLAG_FORCE32, ot.FLAG_THREADS | ot.FLAG_NO_DEDUP]
n = 0
for count in (1023, 1024, 1025, 2047, 2049, 3073, 16384, 16385, 65537, 131072, 250001):
    for flags in flags_list:
        tg.test_descriptor_order_and_offsets_both_tail_paths(product, oracle, count, flags)
        n += 1
        print("ok", count, hex(flags), flush=True)
print("stress ok:", n, "cases")
