#!/bin/bash
# ordered list of the launches of ONE configs[1] bake (device-resident entry): usage: bash profiles/scripts/r05_c1_order.sh [config]
cfg=${1:-c1}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/order_$cfg; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -o t -- python $R/bench.py --config $cfg --cpu-sample 0 --sat-off-sample 0 --host-api-steps 0 --create-texture 0 --steps 2 --warmup 1 > $O/log 2>&1
cd $R
python - <<P
import csv,glob
rows=[]
for f in glob.glob('$O/t/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:70]))
for f in glob.glob('$O/t/**/*memory_copy_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')))
rows.sort()
# last bake = after the last classify_tiles-free gap: print the last 90 rows
t0=rows[-90][0] if len(rows)>90 else rows[0][0]
with open('$O/order.txt','w') as o:
    prev=None
    for s,e,n in rows[-90:]:
        o.write('%9.1f %7.1f gap %6.1f  %s\n'%((s-t0)/1e3,(e-s)/1e3,((s-prev)/1e3 if prev else 0),n)); prev=e
P
rm -rf $O/t
