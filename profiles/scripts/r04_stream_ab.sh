#!/bin/bash
# streamed ommCpuBake at the metric configuration: library builds (OMM_AMD_LIBRARY) x ranges (--stream-chunks).  usage: r04_stream_ab.sh <prefix> "<chunks list>" lib...
out=$1; chunks=$2; shift 2
for lib in "$@"; do for k in $chunks; do
  name=$(basename $lib .so)
  OMM_AMD_LIBRARY=$PWD/$lib timeout 600 python bench.py --config c2 --steps 8 --warmup 2 --cpu-sample 0 --create-texture 0 --sat-off-sample 0 --stream-chunks $k > gpurun_out/${out}_${name}_k$k.json 2> gpurun_out/${out}_${name}_k$k.err
  python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/${out}_${name}_k$k.json").read().strip().splitlines()[-1]); st=j["host_api"]["stream"]; p=j["host_api"]["phases_ms"]
    rr=st["range_ready_ms"]
    print("${name} k=$k: ommCpuBake %.2f ms | classify %.2f first copy %.2f last range %.2f exposed %.2f preview %.2f | device %.2f" % (j["ms_per_step"], p["classifyMs"], st["first_copy_issued_ms"], rr[-2] if len(rr)>1 else -1, st["exposed_copy_ms"], st["preview_ms"], j["device_resident"]["ms_per_bake"]))
except Exception as e:
    print("${name} k=$k FAILED", e)
PY
done; done
