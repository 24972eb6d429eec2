#!/bin/bash
# round 5, call 13: new deep tests + parts test; which totals overflow at 15 000 first-pass; rocprofv3 of the deep alignments (stats + FETCH/WRITE)
out=gpurun_out/r05_c13; mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_ddeep.py tests/test_gpu_parity.py -x -q -k "hierarchical or many_workgroups" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
MPRG_PLAN_TRACE=1 timeout 600 python bench.py --batch 15000 --first-pass --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg > $out/bench_15000_first.json 2> $out/bench_15000_first.err
grep "\[plan\]" $out/bench_15000_first.err | cut -c1-400 | head -12
for shape in "2000 4000" "10000 20000"; do
  set -- $shape
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$1 -- python3 tools/deep_profile.py $1 $2 7 --passes 1 > $out/run_stats_$1.txt 2>&1
  f=$(find $out/prof_$1 -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats_$1x$2.csv; head -8 $f | cut -c1-150
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_${pmc}_$1 -- python3 tools/deep_profile.py $1 $2 7 --passes 1 > $out/run_${pmc}_$1.txt 2>&1
    f=$(find $out/pmc_${pmc}_$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $out/pmc_${pmc}_$1x$2.csv.gz
  done
  rm -rf $out/prof_$1 $out/pmc_FETCH_SIZE_$1 $out/pmc_WRITE_SIZE_$1
done
ls -la $out | head -30
