# after `gpurun -- bash profiles/scripts/r05_final_run.sh <tag>`: the summaries that are judged go from gpurun_out/ (scratch) into profiles/ (tracked)
# usage (here, not on the GPU box): bash profiles/scripts/r05_copy_results.sh <tag>
tag=${1:-r05_v4}
for c in c1 c2 c4 cards; do d=gpurun_out/${tag}_$c
  cp $d/bench.json profiles/${tag}_${c}_bench.json; cp $d/kernel_stats.md profiles/${tag}_${c}_kernel_stats.md
  cp $d/${tag}_${c}_pmc.md profiles/${tag}_${c}_pmc.md; cp $d/${tag}_${c}_hbm_traffic.json profiles/pmc_latest_$c.json
done
cp gpurun_out/${tag}_c1_concurrent.json profiles/${tag}_c1_concurrent_bench.json
for m in 1 2 3; do cp gpurun_out/${tag}_c2_transfer$m.json profiles/${tag}_c2_transfer${m}_bench.json; done
for n in 2 4 8; do cp gpurun_out/${tag}_c2_devices$n.json profiles/${tag}_c2_devices${n}_bench.json; done
cp gpurun_out/${tag}_default_bench.json profiles/${tag}_default_bench.json
