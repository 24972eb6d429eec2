#!/bin/bash
# kernel-level durations (rocprofv3) of the fused loop's three kernels against the per-round kernels, 15 000 alignments per step
out=gpurun_out/r04_c07; mkdir -p $out
export TMPDIR=/tmp
for mode in fused rounds; do
  MPRG_KLOOP=$mode MPRG_BACKEND=runtime rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$mode -- python3 tools/forest_profile.py 15000 2 > $out/run_$mode.txt 2>&1
  f=$(find $out/prof_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; head -14 $f | cut -c1-170
  cp $f $out/kernel_stats_$mode.csv
done
