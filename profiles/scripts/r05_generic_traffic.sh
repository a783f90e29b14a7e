#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of classify_generic on the cards workload for library variants: usage: bash profiles/scripts/r05_generic_traffic.sh <name|main> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = main ]; then unset OMM_AMD_LIBRARY; else export OMM_AMD_LIBRARY=$R/profiles/bin/ab/$name.so; fi
  O=/tmp/traffic_$name; rm -rf $O; mkdir -p $O
  B="python $R/bench.py --config cards --steps 1 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
  for c in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o pmc -- $B > $O/log_$c 2>&1 || tail -c 400 $O/log_$c; done
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "classify_generic" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print("$name", {c: "%.4g" % (v / max(1, cnt[(k, c)])) for c, v in d.items()})
if not acc: print("$name: no counters")
PY
done
