#!/bin/bash
set -x
mkdir -p gpurun_out/r03_c05
python tools/forest_profile.py 30000 3 > gpurun_out/r03_c05/prof_30000.txt 2>&1
head -4 gpurun_out/r03_c05/prof_30000.txt
for w in 1 2 3; do
python bench.py --no-cpu-baseline --no-end-to-end --workers $w --steps 6 --warmup 2 > gpurun_out/r03_c05/bench_w$w.json 2> gpurun_out/r03_c05/bench_w$w.err
done
python -m pytest tests -m gpu -x -q > gpurun_out/r03_c05/pytest_gpu.txt 2>&1; tail -5 gpurun_out/r03_c05/pytest_gpu.txt
