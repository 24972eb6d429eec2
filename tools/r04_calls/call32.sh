#!/bin/bash
# round-4 measurement set on the final sources: bench + rocprofv3 + PMC passes (measure_round.sh), config D (flat generator:
# timing, kernel stats, FETCH / WRITE passes), deep (hierarchical) alignments per entry point, GPU tests
bash tools/measure_round.sh r04 > gpurun_out/measure_r04.log 2>&1; tail -5 gpurun_out/measure_r04.log | cut -c1-300
out=gpurun_out/r04_config_d; mkdir -p $out
export TMPDIR=/tmp
MPRG_CONFIG_D_OUT=$out/config_d_timing.json python tools/config_d_profile.py --passes 3 2>&1 | tee $out/config_d_timing.txt | cut -c1-200 | head -8
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_stats.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats.csv
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$pmc -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_$pmc.txt 2>&1
  f=$(find $out/pmc_$pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $out/pmc_$pmc.csv.gz
done
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
out=gpurun_out/r04_deep; mkdir -p $out
MPRG_DEEP_OUT=$out/deep_2000x4000.json timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 2 --check 200 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $out/deep_2000x4000.txt
MPRG_KM_BIG_BYTES=0 MPRG_DEEP_OUT=$out/deep_2000x4000_before.json timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 1 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $out/deep_2000x4000_before.txt
MPRG_DEEP_OUT=$out/deep_5000x10000.json timeout 900 python tools/deep_profile.py 5000 10000 7 --passes 1 --check 60 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $out/deep_5000x10000.txt
MPRG_DEEP_OUT=$out/deep_10000x20000.json timeout 1500 python tools/deep_profile.py 10000 20000 7 --passes 1 --check 60 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $out/deep_10000x20000.txt
head -4 $out/deep_*.txt | cut -c1-160
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r04_pytest_gpu.txt
