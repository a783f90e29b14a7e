#!/bin/bash
# kernel trace of STREAMED ommCpuBake calls (host arrays in / out): which kernels run in front of and next to the persistent classification launch.
# usage (GPU box): bash profiles/scripts/r04_stream_trace.sh <tag> [config]
tag=${1:-r04_stream}; cfg=${2:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/$tag; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --config $cfg --steps 1 --warmup 1 --host-api-steps 5 --cpu-sample 0 --sat-off-sample 0 --create-texture 0 > $O/trace.log 2>&1
cd $R
python profiles/summarize_rocprof.py $(find $O/trace -name "*.db" | head -1) > $O/kernel_stats.md 2> $O/kernel_stats.err
head -40 $O/kernel_stats.md
rm -rf $O/trace
