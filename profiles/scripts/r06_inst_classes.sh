#!/bin/bash
# instruction classes of the kernels of one bench configuration (two counter passes), per launch.  usage: r06_inst_classes.sh <tag> <config> [variant|main] [kernel substring]
tag=${1:-r06_cls}; cfg=${2:-cards}; var=${3:-main}; kern=${4:-classify_generic}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
[ "$var" = main ] || export OMM_AMD_LIBRARY=$R/profiles/bin/ab/$var.so
O=$R/gpurun_out/$tag; mkdir -p $O
B="python $R/bench.py --config $cfg --steps 2 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1)); timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -o pmc -- $B > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -c 300 $O/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$kern" not in r["Kernel_Name"] or "serial" in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for c in sorted(acc): print("%-28s %.4g" % (c, acc[c] / max(1, cnt[c])))
v = acc["SQ_INSTS_VALU"] / max(1, cnt["SQ_INSTS_VALU"])
other = v - sum(acc[k] / max(1, cnt[k]) for k in acc if k.startswith("SQ_INSTS_VALU_"))
print("%-28s %.4g" % ("VALU not classified", other))
PY
rm -rf $O/p[0-9]*/
