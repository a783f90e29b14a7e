# kernel-by-kernel timeline of ONE bake (everything around classify_tiles): set-up, triage, tail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r02/small; mkdir -p $O
B="python $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- $B > $O/t.log 2>&1
cd $R
python - <<'PY'
import csv,glob,os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r02/small")
rows=[]
for p in glob.glob(O+"/t/**/*kernel_trace.csv",recursive=True): rows+=list(csv.DictReader(open(p)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "classify_tiles" in r["Kernel_Name"] and "4096" in r["Kernel_Name"]]
a,b=idx[-2],idx[-1]
seg=rows[a+1:min(len(rows),b+1)]
t0=int(seg[0]["Start_Timestamp"])
n=0
for r in seg:
    n+=1
    print("+%8.1f us %7.1f us %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:70]))
print("launches between two classify_tiles:", n)
PY
