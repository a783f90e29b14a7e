timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for c in c2 cards c4 c1; do bash profiles/collect.sh r04_v11_$c $c > gpurun_out/collect_$c.log 2>&1; done
for k in 1 4 16; do timeout 300 python bench.py --config c1 --concurrent $k --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 20 --warmup 5 > gpurun_out/r04_v11_c1_concurrent_$k.json 2> gpurun_out/r04_v11_c1_concurrent_$k.err; done
bash profiles/scripts/r04_stream_trace.sh r04_v11_stream c2 > gpurun_out/r04_v11_stream.log 2>&1
for w in 1 2 4 8; do GPU_MAX_HW_QUEUES=1 timeout 600 python profiles/scripts/r03_ranks_in_turn.py $w c2 3 > gpurun_out/r04_ranks_in_turn_c2_w$w.log 2>&1; done
ls gpurun_out | grep r04_v11 | head -20
