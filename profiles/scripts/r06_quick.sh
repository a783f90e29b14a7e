#!/bin/bash
# device phases of a bench configuration for library variants: usage: r06_quick.sh <config> <variant|main> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; cfg=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset OMM_AMD_LIBRARY; else export OMM_AMD_LIBRARY=$R/profiles/bin/ab/$v.so; fi
  for i in 1 2; do timeout 300 python bench.py --config $cfg --cpu-sample 0 --sat-off-sample 0 --create-texture 0 --host-api-steps 0 --steps 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['phases_ms']; print('$cfg $v: persistent %.3f generic %.2f classify %.3f device %.3f' % (p['persistentMs'], p['genericMs'], p['classifyMs'], j['device_resident']['ms_per_bake']))"; done
done
