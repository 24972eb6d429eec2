#!/bin/bash
# round 5, call 35: the GPU suite, then config D's measurement set (timing, rocprofv3 stats, FETCH / WRITE passes) for profiles/r05/config_d
out=gpurun_out/r05_c35; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
MPRG_CONFIG_D_OUT=$out/config_d_timing.json timeout 600 python tools/config_d_profile.py --passes 4 > $out/config_d_timing.txt 2>&1; tail -22 $out/config_d_timing.txt | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_stats.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats.csv
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$pmc -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_$pmc.txt 2>&1
  f=$(find $out/pmc_$pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $out/pmc_$pmc.csv.gz
done
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
python tools/summarize_config_d.py $out > $out/kernels.md; cat $out/kernels.md
