R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r02/classes0; mkdir -p $O
B="python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $O/p1 -o pmc -- $B > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/p2 -o pmc -- $B > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p3 -o pmc -- $B > $O/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VSKIPPED GRBM_GUI_ACTIVE --output-format csv -d $O/p4 -o pmc -- $B > $O/p4.log 2>&1
cd $R
python - <<'PY'
import csv,glob,collections,os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r02/classes0")
for p in sorted(glob.glob(O+"/p*/**/*counter_collection.csv",recursive=True)):
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(p)):
        if "classify_tiles" in r["Kernel_Name"]:
            a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print(k, v[0], "%.4g"%(v[1]/max(1,v[0])))
PY
