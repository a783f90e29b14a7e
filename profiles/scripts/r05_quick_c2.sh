#!/bin/bash
# digest / device timing of the metric configuration + the digest tests: usage: bash profiles/scripts/r05_quick_c2.sh
timeout 600 python -m pytest tests -m gpu -x -q -k "digest or HexagonsReuse or highest_levels or texel or fuzz_deferred" 2>&1 | tail -2
for c in c2 cards c4; do timeout 600 python bench.py --config $c --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --host-api-steps 0 --steps 6 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['phases_ms']; print('$c: digest %.3f generic %.2f persistent %.2f device %.2f' % (p['digestMs'], p['genericMs'], p['persistentMs'], j['device_resident']['ms_per_bake']))"; done
