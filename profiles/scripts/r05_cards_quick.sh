timeout 600 python -m pytest tests -m gpu -x -q -k "texel or nearest or walk or promotion or degenerate" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --config cards --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --host-api-steps 0 --steps 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['phases_ms']; print('generic %.2f persistent %.2f device %.2f' % (p['genericMs'], p['persistentMs'], j['device_resident']['ms_per_bake']))"; done
