# instruction-fetch / wait counters of classify_tiles (which SQ counters exist is listed first)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r02/ifetch; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|SQC)_[A-Z0-9_]+" | sort -u > $O/avail.txt
grep -E "IFETCH|ICACHE|INST_LEVEL|WAIT_INST|WAIT_ANY|WAIT_EXP|DCACHE|SQC_|EXP_REQ|LEVEL_WAVES|VALU_DEP|BUSY_CU|CYCLES" $O/avail.txt | tr '\n' ' '
echo
B="python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o pmc -- $B > $O/$n.log 2>&1 || tail -3 $O/$n.log; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU
run p2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM
run p3 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_EXP_GDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM
run p4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES
cd $R
python - <<'PY'
import csv,glob,collections,os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r02/ifetch")
for p in sorted(glob.glob(O+"/p*/**/*counter_collection.csv",recursive=True)):
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(p)):
        if "classify_tiles" in r["Kernel_Name"] and "4096" in r["Kernel_Name"]:
            a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print(k, v[0], "%.5g"%(v[1]/max(1,v[0])))
PY
