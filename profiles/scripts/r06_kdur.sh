#!/bin/bash
# kernel durations of a bench configuration for library variants: usage r06_kdur.sh <config> <variant|main> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; cfg=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset OMM_AMD_LIBRARY; else export OMM_AMD_LIBRARY=$R/profiles/bin/ab/$v.so; fi
  rm -rf /tmp/ks_$v; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$v -o t --output-format csv -- python $R/bench.py --config $cfg --steps 4 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0 > /tmp/ks_$v.log 2>&1
  f=$(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1)
  echo "== $cfg $v"; python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:7]:
    print("%-70s calls %s avg %.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
