R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r02/dedup; mkdir -p $O
B="python $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/hash -o t -- $B > $O/hash.log 2>&1

cd $R
python - <<'PY'
import csv,glob,os,collections
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r02/dedup")
for v in ("hash",):
    rows=[]
    for p in glob.glob(O+"/"+v+"/**/*kernel_trace.csv",recursive=True):
        rows+=list(csv.DictReader(open(p)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    # last bake: find last classify_tiles 4096, print kernels after it until end
    idx=[i for i,r in enumerate(rows) if "classify_tiles" in r["Kernel_Name"] and "4096" in r["Kernel_Name"]][-1]
    print("==",v)
    t0=int(rows[idx]["End_Timestamp"])
    for r in rows[idx+1:idx+40]:
        print("  +%7.1f us  %6.1f us  %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:90]))
PY
