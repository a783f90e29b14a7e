#!/bin/bash
# ordered list of what the GPU does during ONE ommCpuBake of a configuration (kernels + copies, with the idle gaps between them): usage: bash profiles/scripts/r05_hostapi_order.sh [config]
cfg=${1:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/hostorder_$cfg; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -o t -- python $R/bench.py --config $cfg --cpu-sample 0 --sat-off-sample 0 --host-api-steps 3 --create-texture 0 --steps 1 --warmup 1 > $O/log 2>&1
cd $R
python - <<P
import csv,glob
rows=[]
for f in glob.glob('$O/t/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60]))
for f in glob.glob('$O/t/**/*memory_copy_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')+' '+r.get('Size','')))
rows.sort()
rows=rows[-130:]
t0=rows[0][0]
with open('$O/order.txt','w') as o:
    prev=None
    for s,e,n in rows:
        o.write('%9.1f %8.1f gap %7.1f  %s\n'%((s-t0)/1e3,(e-s)/1e3,((s-prev)/1e3 if prev else 0),n)); prev=max(prev or 0,e)
P
rm -rf $O/t
