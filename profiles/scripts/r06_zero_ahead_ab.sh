#!/bin/bash
# ommCpuBake at the metric configuration with and without the zeroing ahead of the result block, N bakes each, alternating processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; n=${1:-100}
for z in 0 1 0 1 0 1; do
  timeout 600 python bench.py --config c2 --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 3 --host-api-steps $n --zero-ahead $z 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['host_api']; r=h['result_transfer']
print('zero-ahead %s: ms_per_bake %.2f  bake p50/p95 %s min/max %s  expand p50/p95 %s  zeroed %.2f GB skipped %.2f GB' % ('off' if $z else 'on', h['ms_per_bake'], [round(x,2) for x in r['bake_ms_p50_p95']], [round(x,2) for x in r['bake_ms_min_max']], [round(x,2) for x in r['copy_and_expand_ms_p50_p95']], r['zeroed_ahead_bytes']/1e9, r['expansion_skipped_bytes']/1e9))"
done
