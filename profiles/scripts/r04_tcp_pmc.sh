cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "TCP_[A-Z_]*\(sum\)\?" | sort -u | head -80 > $R/gpurun_out/tcp_counters.txt
BENCH="python $R/bench.py --config cards --steps 2 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_ACCESSES_sum"; do
  n=$(echo $grp | cut -c1-20 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/tcp_$n -o pmc -- $BENCH > $R/gpurun_out/tcp_$n.log 2>&1
  python3 - <<PY
import csv,glob,collections
for f in glob.glob("$R/gpurun_out/tcp_$n/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'classify_generic' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
    for k in acc: print(k, acc[k]/max(cnt[k],1)*1.0, "per dispatch-row avg; rows", cnt[k])
PY
  rm -rf $R/gpurun_out/tcp_$n
done
