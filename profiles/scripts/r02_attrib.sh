# instruction counts of classify_tiles per component: debug variants (wrong results on purpose) under one PMC pass each
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r02/attrib; mkdir -p $O
B="python $R/bench.py --steps 2 --warmup 0 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
for f in default $R/omm_amd/lib/variants/*.so; do
  v=$(basename $f .so); [ $f = default ] || export OMM_AMD_LIBRARY=$f
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $O/$v -o pmc -- $B > $O/$v.log 2>&1
done
cd $R
python - <<'PY'
import csv,glob,collections,os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r02/attrib")
for d in sorted(glob.glob(O+"/*/")):
    v=os.path.basename(d.rstrip("/"))
    agg=collections.defaultdict(lambda:[0,0.0]); dur=[]
    for p in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(p)):
            if "classify_tiles" in r["Kernel_Name"] and "4096" in r["Kernel_Name"]:
                a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for p in glob.glob(d+"/**/*kernel_trace.csv",recursive=True):
        for r in csv.DictReader(open(p)):
            if "classify_tiles" in r["Kernel_Name"] and "4096" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print(v, "ms=%.2f"%(sum(dur)/max(1,len(dur))), " ".join("%s=%.4g"%(k.replace("SQ_INSTS_",""),x[1]/max(1,x[0])) for k,x in sorted(agg.items())))
PY
