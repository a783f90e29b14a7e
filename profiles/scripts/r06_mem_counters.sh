#!/bin/bash
# Memory-system counters of one bench configuration, one counter group per pass (kernel-trace only), averaged per launch for the kernels named.
# (the TCC_* / TCP_* derived counters abort rocprofv3 on this image -- signal 6 after the 300-s limit --: they are not in the list)
# usage: bash profiles/scripts/r06_mem_counters.sh <tag> <config> [library variant under profiles/bin/ab | main]
tag=${1:-r06_mem}; cfg=${2:-cards}; var=${3:-main}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
[ "$var" = main ] || export OMM_AMD_LIBRARY=$R/profiles/bin/ab/$var.so
O=$R/gpurun_out/$tag; mkdir -p $O
[ -s $R/gpurun_out/avail_counters.txt ] || rocprofv3 --list-avail > $R/gpurun_out/avail_counters.txt 2>&1
B="python $R/bench.py --config $cfg --steps 2 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -o pmc -- $B > $O/p$i.log 2>&1 || { echo "pass $i ($grp) failed:"; tail -c 300 $O/p$i.log; }
done
python3 - <<PY > $O/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        k = k[:k.find("<")] + ("<..DEFER>" if k.rstrip(" >").endswith("true") and "classify_tiles" in k else "") if "<" in k else k
        if not any(s in k for s in ("classify_generic", "classify_tiles", "generic_merge", "triage_groups", "tail_gather")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        if (f, r["Dispatch_Id"]) not in seen:
            seen.add((f, r["Dispatch_Id"])); dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for k, d in sorted(acc.items()):
    print("== %s   (avg %.3f ms in the counter passes)" % (k, sum(dur[k]) / max(1, len(dur[k]))))
    for c, v in sorted(d.items()): print("   %-44s %.5g per launch" % (c, v / max(1, cnt[(k, c)])))
PY
cat $O/summary.txt
rm -rf $O/p[0-9]*/
