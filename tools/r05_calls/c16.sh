#!/bin/bash
# round 5, call 16: config D (flat 10 000 x 20 000) after the word-wise ingest and the 64-column tiles of big problems: rocprofv3 stats + PMC
out=gpurun_out/r05_c16; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/config_d_profile.py --passes 4 > $out/config_d_timing.txt 2>&1; tail -25 $out/config_d_timing.txt | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_stats.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats.csv; head -14 $f | cut -c1-160
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$pmc -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_$pmc.txt 2>&1
  f=$(find $out/pmc_$pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $out/pmc_$pmc.csv.gz
done
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
timeout 900 python -m pytest tests/test_gpu_config_b.py -x -q -k "config_d" > $out/pytest_config_d.txt 2>&1; tail -3 $out/pytest_config_d.txt
