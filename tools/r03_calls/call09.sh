#!/bin/bash
mkdir -p gpurun_out/r03_c09
MPRG_BENCH_TRACE=1 python bench.py --no-cpu-baseline --no-end-to-end --workers 1 --steps 6 --warmup 2 > gpurun_out/r03_c09/bench_w1.json 2> gpurun_out/r03_c09/bench_w1.err
grep trace gpurun_out/r03_c09/bench_w1.err
MPRG_BENCH_TRACE=1 python bench.py --no-cpu-baseline --no-end-to-end --workers 0 --steps 6 --warmup 2 > gpurun_out/r03_c09/bench_w0.json 2> gpurun_out/r03_c09/bench_w0.err
grep trace gpurun_out/r03_c09/bench_w0.err
