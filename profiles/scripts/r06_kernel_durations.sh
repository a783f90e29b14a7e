#!/bin/bash
# kernel durations of the cards bench for a library variant
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
[ "$1" = main ] || export OMM_AMD_LIBRARY=$R/profiles/bin/ab/$1.so
rm -rf /tmp/ks_$1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$1 -o t --output-format csv -- python $R/bench.py --config cards --steps 3 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0 > /tmp/ks_$1.log 2>&1
f=$(find /tmp/ks_$1 -name "*kernel_stats.csv" | head -1)
echo "== $1"; python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:6]:
    print("%-60s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
