# A/B of two library builds on the ommCpuBake transfer (alternating, 3 rounds): usage r05_transfer_ab.sh libA.so libB.so
for i in 1 2 3; do for lib in "$@"; do
OMM_AMD_LIBRARY=$PWD/$lib timeout 300 python bench.py --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 20 > gpurun_out/r05_ab_tmp.json 2>/dev/null; python - <<PY
import json
j=json.loads(open("gpurun_out/r05_ab_tmp.json").read().strip().splitlines()[-1])
h=j["host_api"]; rt=h["result_transfer"]; print("$lib run $i:", round(h["ms_per_bake"],2), [round(x,2) for x in rt["copy_and_expand_ms_min_max"]], [round(x,2) for x in rt["bake_ms_min_max"]])
PY
done; done
