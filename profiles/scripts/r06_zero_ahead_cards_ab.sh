R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for z in 0 1 0 1 0 1; do
  timeout 600 python bench.py --config cards --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --steps 3 --host-api-steps 40 --zero-ahead $z 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['host_api']; r=h['result_transfer']
print('cards zero-ahead %s: ms_per_bake %.2f  p50/p95 %s  expand p50/p95 %s  skipped %.2f GB of %.2f' % ('off' if $z else 'on', h['ms_per_bake'], [round(x,2) for x in r['bake_ms_p50_p95']], [round(x,2) for x in r['copy_and_expand_ms_p50_p95']], r['expansion_skipped_bytes']/1e9, r['array_data_bytes']/1e9))"
done
