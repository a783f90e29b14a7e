#!/bin/bash
# quick check of the asset-shaped path: the parity tests that exercise the deferred generic pass, then two short cards benches (device phases)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
[ -n "$1" ] && [ "$1" != main ] && export OMM_AMD_LIBRARY=$R/profiles/bin/ab/$1.so
[ "$2" = notest ] || timeout 900 python -m pytest tests -m gpu -x -q -k "texel or nearest or walk or promotion or degenerate or generic or Leaflet or leaflet" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --config cards --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --host-api-steps 0 --steps 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['phases_ms']; print('$1 generic %.2f persistent %.2f classify %.2f device %.2f' % (p['genericMs'], p['persistentMs'], p['classifyMs'], j['device_resident']['ms_per_bake']))"; done
