#!/bin/bash
# per-bake distribution of ommCpuBake at the metric configuration over N bakes: usage: r06_hostapi_jitter.sh <N> <variant|main> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; n=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset OMM_AMD_LIBRARY; else export OMM_AMD_LIBRARY=$R/profiles/bin/ab/$v.so; fi
  timeout 600 python bench.py --config c2 --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 3 --host-api-steps $n 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['host_api']; r=h['result_transfer']
print('$v: ms_per_bake %.2f  bake p50/p95 %s min/max %s  expand p50/p95 %s min/max %s' % (h['ms_per_bake'], r.get('bake_ms_p50_p95'), r['bake_ms_min_max'], r.get('copy_and_expand_ms_p50_p95'), r['copy_and_expand_ms_min_max']))"
done
