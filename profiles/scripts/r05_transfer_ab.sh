# A/B of two library builds on the ommCpuBake transfer (alternating, N rounds; a fresh process per run: the expansion rate is a property of the process): usage r05_transfer_ab.sh rounds libA.so libB.so
n=$1; shift
for i in $(seq 1 $n); do for lib in "$@"; do
OMM_AMD_LIBRARY=$PWD/$lib timeout 300 python bench.py --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 12 > gpurun_out/r05_ab_tmp.json 2>/dev/null; python - <<PY
import json
j=json.loads(open("gpurun_out/r05_ab_tmp.json").read().strip().splitlines()[-1])
h=j["host_api"]; rt=h["result_transfer"]; print("$lib run $i:", round(h["ms_per_bake"],2), [round(x,2) for x in rt["copy_and_expand_ms_min_max"]], [round(x,2) for x in rt["bake_ms_min_max"]])
PY
done; done
