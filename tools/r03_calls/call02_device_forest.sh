#!/bin/bash
# device-resident forest: GPU parity tests, then the bench with one worker (whole batch) and with the default workers
set -x
mkdir -p gpurun_out/r03_c02
python -m pytest tests -m gpu -x -q > gpurun_out/r03_c02/pytest_gpu.txt 2>&1; tail -5 gpurun_out/r03_c02/pytest_gpu.txt
python bench.py --no-cpu-baseline --no-end-to-end --workers 1 --steps 3 --warmup 1 > gpurun_out/r03_c02/bench_w1.json 2> gpurun_out/r03_c02/bench_w1.err
python bench.py --no-cpu-baseline --no-end-to-end --workers 2 --steps 3 --warmup 1 > gpurun_out/r03_c02/bench_w2.json 2> gpurun_out/r03_c02/bench_w2.err
python bench.py --no-cpu-baseline --no-end-to-end --steps 5 --warmup 1 > gpurun_out/r03_c02/bench_w10.json 2> gpurun_out/r03_c02/bench_w10.err
tail -c 300 gpurun_out/r03_c02/*.err
