# A/B timing of library variants (omm_amd/lib/variants/*.so, built with different EXTRA flags) + the default build; usage: bash profiles/scripts/r02_ab.sh [pytest -k expr]
O=gpurun_out/r02; mkdir -p $O
B="--steps 5 --warmup 2 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0"
python bench.py $B > $O/ab_default.json 2> $O/ab_default.err
python -c "import json;d=json.load(open('$O/ab_default.json'));print('default', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms'].items()})"
for f in omm_amd/lib/variants/*.so; do
  v=$(basename $f .so)
  OMM_AMD_LIBRARY=$PWD/$f python bench.py $B > $O/ab_$v.json 2> $O/ab_$v.err
  python -c "import json;d=json.load(open('$O/ab_$v.json'));print('$v', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms'].items()})" || tail -3 $O/ab_$v.err
  grep OMMX_STATS $O/ab_$v.err | tail -1
done
if [ -n "$1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -5; fi
