#!/bin/bash
# round 5, call 57: the GPU suite and the deep alignments' measurement set on the round's last sources (as call 42)
# bytes, rocprofv3 stats, FETCH / WRITE passes) for profiles/r05/deep
out=gpurun_out/r05_c57; mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
for shape in "2000 4000" "5000 10000" "10000 20000"; do
  set -- $shape
  MPRG_DEEP_OUT=$out/deep_$1x$2.json timeout 1200 python tools/deep_profile.py $1 $2 7 --passes 3 > $out/deep_$1x$2.txt 2>&1
  grep -E '^\{' $out/deep_$1x$2.txt | tail -1 | cut -c1-200
done
for shape in "2000 4000" "10000 20000"; do
  set -- $shape
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$1 -- python3 tools/deep_profile.py $1 $2 7 --passes 1 > $out/run_stats_$1.txt 2>&1
  f=$(find $out/prof_$1 -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats_$1x$2.csv; head -6 $f | cut -c1-150
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_${pmc}_$1 -- python3 tools/deep_profile.py $1 $2 7 --passes 1 > $out/run_${pmc}_$1.txt 2>&1
    f=$(find $out/pmc_${pmc}_$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $out/pmc_${pmc}_$1x$2.csv.gz
  done
  rm -rf $out/prof_$1 $out/pmc_FETCH_SIZE_$1 $out/pmc_WRITE_SIZE_$1
done
