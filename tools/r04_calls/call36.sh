#!/bin/bash
# the deep alignment under rocprofv3: kernel stats + FETCH_SIZE / WRITE_SIZE passes (ddeep, 2 000 x 4 000)
out=gpurun_out/r04_deep_prof; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/deep_profile.py 2000 4000 7 --passes 1 > $out/run_stats.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats.csv; head -12 $f | cut -c1-140
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$pmc -- python3 tools/deep_profile.py 2000 4000 7 --passes 1 > $out/run_$pmc.txt 2>&1
  f=$(find $out/pmc_$pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $out/pmc_$pmc.csv.gz
done
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
