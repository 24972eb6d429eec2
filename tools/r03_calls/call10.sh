#!/bin/bash
mkdir -p gpurun_out/r03_c10
for w in 1 2 4; do
MPRG_BENCH_TRACE=1 python bench.py --no-cpu-baseline --no-end-to-end --workers $w --steps 6 --warmup 2 > gpurun_out/r03_c10/bench_w$w.json 2> gpurun_out/r03_c10/bench_w$w.err
done
grep trace gpurun_out/r03_c10/bench_w1.err
