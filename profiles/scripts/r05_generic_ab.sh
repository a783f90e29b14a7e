#!/bin/bash
# usage: bash profiles/scripts/r05_generic_ab.sh <name> ...   -- the cards bake with profiles/bin/ab/<name>.so (or "main" = the tree's library), twice each
for name in "$@"; do
  if [ "$name" = main ]; then unset OMM_AMD_LIBRARY; else export OMM_AMD_LIBRARY=$PWD/profiles/bin/ab/$name.so; fi
  for i in 1 2; do timeout 300 python bench.py --config cards --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --host-api-steps 0 --steps 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['phases_ms']; print('$name: generic %.2f persistent %.2f device %.2f' % (p['genericMs'], p['persistentMs'], j['device_resident']['ms_per_bake']))"; done
done
