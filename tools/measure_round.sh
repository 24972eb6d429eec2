#!/bin/bash
# GPU-box helper: the measurement set of a round -> gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>/).
#   1. bench.py as the driver runs it (10 host worker processes; CPU baseline, verification, exclusive pass, end-to-end leg)
#   2. rocprofv3 --kernel-trace --stats of bench.py in its in-process mode (--workers 0: nothing forks under the profiler,
#      one stream: kernel durations are exclusive)
#   3. + 4. HBM traffic: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "nproc $(nproc)  affinity $(python -c 'import os; print(len(os.sched_getaffinity(0)))')  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" > $out/host.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" >> $out/host.txt
cat $out/host.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?"
inproc="--workers 0 --streams 1 --batch 8192 --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 2"   # one process generates its alignments serially: smaller batch
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py $inproc > $out/bench_under_rocprof.json 2> $out/prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $inproc > $out/pmc_fetch_bench.json 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $inproc > $out/pmc_write_bench.json 2> $out/pmc_write.err
stats=$(find $out/prof -name "*kernel_stats.csv" | head -1)
fetch=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1)
write=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
python tools/summarize_pmc.py $stats $fetch $write $out/bench_under_rocprof.json $out/pmc_summary.json > $out/pmc_summary.txt 2>&1
cp $stats $out/rocprofv3_kernel_stats.csv
cut -c1-600 $out/bench_default.json
head -30 $out/rocprofv3_kernel_stats.csv | cut -c1-160
# 5. the fused clustering loop (what a small shard runs: < 6 000 alignments per engine): kernel stats of 3 750 alignments per pass
inproc_small="--workers 0 --streams 1 --batch 3750 --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_small -- python3 bench.py $inproc_small > $out/bench_under_rocprof_3750.json 2> $out/prof_small.err
cp $(find $out/prof_small -name "*kernel_stats.csv" | head -1) $out/rocprofv3_kernel_stats_3750_fused.csv
head -8 $out/rocprofv3_kernel_stats_3750_fused.csv | cut -c1-160
rm -rf $out/prof $out/prof_small
gzip -c $fetch > $out/pmc_fetch_counter_collection.csv.gz; gzip -c $write > $out/pmc_write_counter_collection.csv.gz
rm -rf $out/pmc_fetch $out/pmc_write
