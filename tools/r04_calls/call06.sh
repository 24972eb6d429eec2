#!/bin/bash
# per-launch durations of the clustering loop at three batch sizes: what is throughput, what is the chain of one problem
out=gpurun_out/r04_c06; mkdir -p $out
for b in 940 3750 15000; do for mode in fused rounds; do
  MPRG_KLOOP=$mode MPRG_BACKEND=runtime python tools/forest_profile.py $b 3 > $out/forest_${b}_$mode.txt 2>&1
  echo "== $b $mode"; grep -E "^step 2|device time|per launch|cluster_loop|kmeans_fit" $out/forest_${b}_$mode.txt | cut -c1-400
done; done
