R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/sp; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/sp -o t --output-format csv -- python $R/tests/scripts/full_size_sharded_check.py 8 > /tmp/sp.log 2>&1
tail -3 /tmp/sp.log
f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:28]:
    print("%-70s calls %5s total %9.1f us avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3))
PY
