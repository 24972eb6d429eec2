#!/bin/bash
# round-3 baseline of the round-2 tree: default bench, one worker with the whole batch, one worker with 3000
set -x
mkdir -p gpurun_out/r03_base
python bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r03_base/bench_w10.json 2> gpurun_out/r03_base/bench_w10.err
python bench.py --no-cpu-baseline --no-end-to-end --workers 1 --steps 3 --warmup 1 > gpurun_out/r03_base/bench_w1.json 2> gpurun_out/r03_base/bench_w1.err
python bench.py --no-cpu-baseline --no-end-to-end --workers 1 --batch 3000 --steps 5 --warmup 1 > gpurun_out/r03_base/bench_w1_b3000.json 2> gpurun_out/r03_base/bench_w1_b3000.err
nproc; free -g
tail -c 600 gpurun_out/r03_base/*.json
