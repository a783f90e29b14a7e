O=gpurun_out/r02; mkdir -p $O
B="--steps 3 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0"
for v in stats nofine nop1; do
  OMM_AMD_LIBRARY=$PWD/omm_amd/lib/variants/libomm-$v.so python bench.py $B > $O/var_$v.json 2> $O/var_$v.err
  python -c "import json;d=json.load(open('$O/var_$v.json'));print('$v', d['ms_per_step'], d['phases_ms'])"
done
grep OMMX_STATS $O/var_stats.err | tail -1
timeout 900 python -m pytest tests -m gpu -x -q -k "kat or golden or fuzz or stats or minimal or basic or log" 2>&1 | tail -3
