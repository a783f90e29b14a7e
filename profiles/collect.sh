#!/bin/bash
# Runs on the GPU box (via gpurun): the bench of one configuration, kernel-trace stats of the same command, then PMC passes (kernel-trace only,
# one counter group per pass: HBM bytes, instruction classes, issue / wait cycles), all under gpurun_out/<tag>/.
# usage: bash profiles/collect.sh <tag> [config]     e.g.  gpurun -- 'bash profiles/collect.sh r03_v1 c2'
# then copy gpurun_out/<tag>/{bench.json -> profiles/<tag>_bench.json, kernel_stats.md, <tag>_pmc.md, <tag>_hbm_traffic.json -> profiles/pmc_latest_<config>.json}
tag=${1:-rXX}; cfg=${2:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/$tag; mkdir -p $O
BENCH="python $R/bench.py --config $cfg"
[ -x $R/profiles/bin/valu_rates ] && [ ! -s $O/valu_rates.json ] && $R/profiles/bin/valu_rates > $O/valu_rates.json 2> $O/valu_rates.err
# (the trace run skips the bench's side bakes -- CPU-baseline parity samples, SAT-off sample, the ommCpuBake steps -- so that every classify_tiles launch in
#  the stats table is the full workload through the device-resident entry and its average is comparable with roofline.avg_launch_ms)
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $BENCH --cpu-sample 0 --sat-off-sample 0 --host-api-steps 0 --create-texture 0 > $O/trace.log 2>&1
PM="--steps 2 --warmup 1 --cpu-sample 0 --host-api-steps 0 --sat-off-sample 0 --create-texture 0"
pass() { n=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o pmc -- $BENCH $PM > $O/pmc_$n.log 2>&1; }
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass C1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
pass C2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass C3 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass C4 SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VSKIPPED GRBM_GUI_ACTIVE
cd $R
python profiles/summarize_rocprof.py $(find $O/trace -name "*.db" | head -1) > $O/kernel_stats.md 2> $O/kernel_stats.err
python profiles/summarize_pmc.py $tag $O "python bench.py --config $cfg $PM" > $O/pmc_summary.json 2> $O/pmc_summary.err
# the bench line comes LAST and reads the counters that were just collected on this very box (round 5: every committed <tag>_<config>_bench.json carries the
# valu_issue roofline and its traffic figure; bench.py accepts the summary only when its source hash equals the tree's)
[ -s $O/${tag}_hbm_traffic.json ] && cp $O/${tag}_hbm_traffic.json $R/profiles/pmc_latest_$cfg.json
cd /tmp
timeout 1200 $BENCH > $O/bench.json 2> $O/bench.err
cd $R
tail -c 900 $O/bench.json; echo; cat $O/pmc_summary.json; head -14 $O/kernel_stats.md
# only the summaries travel back (gpurun merges at most 64 MiB; the raw trace database and counter CSVs of one configuration are larger than that)
[ -z "$OMMX_KEEP_RAW" ] && rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_C1 $O/pmc_C2 $O/pmc_C3 $O/pmc_C4
