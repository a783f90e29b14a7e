#!/bin/bash
# A/B of library builds on one box: bench.py per build (OMM_AMD_LIBRARY), short form.  usage: r04_ab.sh <out-prefix> <config> <steps> lib1 lib2 ...
out=$1; cfg=$2; steps=$3; shift 3
mkdir -p gpurun_out
for lib in "$@"; do
  name=$(basename $lib .so)
  OMM_AMD_LIBRARY=$PWD/$lib timeout 600 python bench.py --config $cfg --steps $steps --warmup 2 --cpu-sample 0 --create-texture 0 --sat-off-sample 0 > gpurun_out/${out}_${cfg}_${name}.json 2> gpurun_out/${out}_${cfg}_${name}.err
  python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/${out}_${cfg}_${name}.json").read().strip().splitlines()[-1])
    p=j["phases_ms"]; h=j.get("host_api",{})
    print("${cfg} ${name}: step %.2f ms classify %.2f persistent %.2f generic %.2f | ommCpuBake %.2f | fine utri %.3g open tiles %d active %d" % (j["ms_per_step"], p["classifyMs"], p["persistentMs"], p["genericMs"], h.get("ms_per_bake",0) or 0, j["config"]["fine_micro_triangles"], j["config"]["open_tiles"], j["config"]["active_items"]))
except Exception as e:
    print("${cfg} ${name}: FAILED", e); print(open("gpurun_out/${out}_${cfg}_${name}.err").read()[-2000:])
PY
done
