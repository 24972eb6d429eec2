#!/bin/bash
# GPU-box helper: the measurement set of a round -> gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>/).
#   1. bench.py as the driver runs it (8 host worker processes; includes the CPU baseline)
#   2. rocprofv3 --kernel-trace --stats of bench.py in its in-process mode (--workers 0: nothing forks under the profiler)
#   3. + 4. HBM traffic: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
tag=${1:-r01d}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench_default.json 2> $out/bench_default.err
inproc="--workers 0 --streams 4 --batch 8192 --no-cpu-baseline --steps 2"   # one process generates its alignments serially: smaller batch
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py $inproc > $out/bench_under_rocprof.json 2> $out/prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $inproc > $out/pmc_run_bench.json 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $inproc > $out/pmc_write_bench.json 2> $out/pmc_write.err
find $out -name "*.csv" | head -20
cut -c1-300 $out/bench_default.json
