# the measurements behind the round-5 numbers, on one lease: GPU suite, the four configurations' sets (kernel table, counter passes, then the bench line that
# carries them), concurrent small bakes, the transfer modes, ommCpuBake over 2 / 4 / 8 ranks that share the one GPU, the default bench line.
# usage (GPU box): bash profiles/scripts/r05_final_run.sh <tag>
tag=${1:-r05_v10}
[ -z "$SKIP_PYTEST" ] && { timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_pytest.log 2>&1; tail -2 gpurun_out/${tag}_pytest.log; }
for c in c2 cards c4 c1; do bash profiles/collect.sh ${tag}_$c $c > gpurun_out/collect_$c.log 2>&1; done
timeout 300 python bench.py --config c1 --concurrent 16 --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 20 --warmup 5 > gpurun_out/${tag}_c1_concurrent.json 2> gpurun_out/${tag}_c1_concurrent.err
for m in 1 2 3; do timeout 300 python bench.py --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 10 --result-transfer $m > gpurun_out/${tag}_c2_transfer$m.json 2> gpurun_out/${tag}_c2_transfer$m.err; done
for n in 2 4 8; do timeout 300 python bench.py --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 6 --devices $n > gpurun_out/${tag}_c2_devices$n.json 2> gpurun_out/${tag}_c2_devices$n.err; done
timeout 600 python bench.py > gpurun_out/${tag}_default_bench.json 2> gpurun_out/${tag}_default_bench.err
ls gpurun_out | grep ${tag} | head -30
