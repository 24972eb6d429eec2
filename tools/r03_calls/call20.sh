#!/bin/bash
mkdir -p gpurun_out/r03_c20
( time python bench.py ) > gpurun_out/r03_c20/bench_default.json 2> gpurun_out/r03_c20/bench_default.err
tail -5 gpurun_out/r03_c20/bench_default.err
python -m pytest tests -m gpu -x -q > gpurun_out/r03_c20/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r03_c20/pytest_gpu.txt
