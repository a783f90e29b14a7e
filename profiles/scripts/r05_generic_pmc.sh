#!/bin/bash
# counters of classify_generic on the cards workload (two passes): usage: bash profiles/scripts/r05_generic_pmc.sh <tag>
tag=${1:-gd}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/$tag; mkdir -p $O
B="python $R/bench.py --config cards --steps 2 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o pmc -- $B > $O/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/p2 -o pmc -- $B > $O/p2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if "classify_generic" not in k and "classify_tiles" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: "%.4g" % (v / max(1, cnt[(k, c)])) for c, v in d.items()})
PY
rm -rf $O/p1 $O/p2
