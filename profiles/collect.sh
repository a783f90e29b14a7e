#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench command, then PMC passes (kernel-trace only, one counter
# group per pass), all under gpurun_out/.  usage: bash profiles/collect.sh <tag>
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/$tag; mkdir -p $O
BENCH="python $R/bench.py"
timeout 900 $BENCH > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $BENCH > $O/trace.log 2>&1
PM="--steps 2 --warmup 1 --cpu-sample 0 --host-api-steps 0"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- $BENCH $PM > $O/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_SQ -o pmc -- $BENCH $PM > $O/pmc_SQ.log 2>&1
cd $R
python profiles/summarize_rocprof.py $(find $O/trace -name "*.db" | head -1) > $O/kernel_stats.md 2> $O/kernel_stats.err
python profiles/summarize_pmc.py $tag $O "python bench.py $PM" > $O/pmc_summary.json 2> $O/pmc_summary.err
cp profiles/${tag}_pmc.md profiles/${tag}_hbm_traffic.json $O/ 2>/dev/null   # gpurun merges only gpurun_out/: copy these four into profiles/ afterwards
tail -c 600 $O/bench.json; echo; cat $O/pmc_summary.json; head -12 $O/kernel_stats.md
