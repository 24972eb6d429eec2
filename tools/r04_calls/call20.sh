#!/bin/bash
# config D (10 000 x 20 000, -N 7): wall / device time, per entry point; rocprofv3 kernel stats; FETCH_SIZE / WRITE_SIZE passes
out=gpurun_out/r04_config_d; mkdir -p $out
export TMPDIR=/tmp
MPRG_CONFIG_D_OUT=$out/config_d_timing.json python tools/config_d_profile.py --passes 3 2>&1 | tee $out/config_d_timing.txt | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_stats.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats.csv; head -16 $f | cut -c1-150
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$pmc -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_$pmc.txt 2>&1
  f=$(find $out/pmc_$pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $out/pmc_$pmc.csv.gz
done
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
