#!/bin/bash
# every triangle of configs[2], of the asset-shaped `cards` workload and of configs[4] against the oracle, slice by slice (tests/scripts/every_triangle.py)
# usage (GPU box): bash tests/scripts/r05_every_triangle.sh   -> gpurun_out/r05_every_triangle_<config>.log
mkdir -p gpurun_out
timeout ${T_C2:-600} python tests/scripts/every_triangle.py c2 1000000 50000 > gpurun_out/r05_every_triangle_c2.log 2>&1; tail -1 gpurun_out/r05_every_triangle_c2.log | cut -c1-400
timeout ${T_CARDS:-300} python tests/scripts/every_triangle.py cards 40000 10000 > gpurun_out/r05_every_triangle_cards.log 2>&1; tail -1 gpurun_out/r05_every_triangle_cards.log | cut -c1-400
timeout ${T_C4:-900} python tests/scripts/every_triangle.py c4 4000000 25000 > gpurun_out/r05_every_triangle_c4.log 2>&1; tail -1 gpurun_out/r05_every_triangle_c4.log | cut -c1-400
