#!/bin/bash
bash tools/measure_round.sh r03_m3 > gpurun_out/r03_m3.log 2>&1
tail -12 gpurun_out/r03_m3.log | cut -c1-400
python -m pytest tests -m gpu -q > gpurun_out/r03_m3/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r03_m3/pytest_gpu.txt
