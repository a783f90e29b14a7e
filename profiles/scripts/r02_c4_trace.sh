R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r02/c4t; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python $R/tests/scripts/c4_full.py > $O/t.log 2>&1
cd $R; python profiles/summarize_rocprof.py $(find $O/t -name "*.db" | head -1) | head -40 | cut -c1-120
