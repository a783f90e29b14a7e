# the measurements behind the round-4 numbers, on one lease: GPU suite, the four configurations' bench + kernel table + counter passes, concurrent small bakes,
# the streamed call's kernel table, the default bench line.  usage (GPU box): bash profiles/scripts/r04_final_run.sh <tag>
tag=${1:-r04_v12}
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for c in c2 cards c4 c1; do bash profiles/collect.sh ${tag}_$c $c > gpurun_out/collect_$c.log 2>&1; done
timeout 300 python bench.py --config c1 --concurrent 16 --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 20 --warmup 5 > gpurun_out/${tag}_c1_concurrent.json 2> gpurun_out/${tag}_c1_concurrent.err
bash profiles/scripts/r04_stream_trace.sh ${tag}_stream c2 > gpurun_out/${tag}_stream.log 2>&1
ls gpurun_out | grep ${tag} | head -20
