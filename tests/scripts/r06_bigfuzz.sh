#!/bin/bash
# Long differential fuzz of the round-5 library on one lease: four seed ranges side by side (the CPU oracle is the slow half; 4 OpenMP threads each).
# usage (GPU box): bash tests/scripts/r06_bigfuzz.sh LO PER_PROCESS [SECONDS [walks]]   -> gpurun_out/r06_bigfuzz_<k>.log, summary on stdout
lo=${1:-30000}; per=${2:-1500}; secs=${3:-1100}; mode=$4   # mode "walks": the deferred-generic-pass cases
mkdir -p gpurun_out; export OMP_NUM_THREADS=4
for k in 0 1 2 3; do
  a=$((lo + k * per)); b=$((a + per))
  timeout $secs python tests/scripts/bigfuzz.py $a $b $mode > gpurun_out/r06_bigfuzz${mode}_$k.log 2>&1 &
done
wait
for k in 0 1 2 3; do echo "range $k: $(grep -c MISMATCH gpurun_out/r06_bigfuzz${mode}_$k.log) mismatch lines; $(tail -1 gpurun_out/r06_bigfuzz${mode}_$k.log | cut -c1-200)"; done
