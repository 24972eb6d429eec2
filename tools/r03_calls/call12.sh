#!/bin/bash
mkdir -p gpurun_out/r03_c12
python -m pytest tests -m gpu -x -q > gpurun_out/r03_c12/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r03_c12/pytest_gpu.txt
python tools/forest_profile.py 30000 2 > gpurun_out/r03_c12/prof_wave.txt 2>&1
MPRG_KMEANS_WAVE=0 python tools/forest_profile.py 30000 2 > gpurun_out/r03_c12/prof_wg.txt 2>&1
grep -A12 "device time" gpurun_out/r03_c12/prof_wave.txt
grep -A8 "device time" gpurun_out/r03_c12/prof_wg.txt
