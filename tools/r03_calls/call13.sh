#!/bin/bash
mkdir -p gpurun_out/r03_c13
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_c13/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r03_c13/pytest_gpu.txt
for mode in 2 0; do
MPRG_KM_MODE=$mode python tools/forest_profile.py 30000 2 > gpurun_out/r03_c13/prof_mode$mode.txt 2>&1
grep -A9 "device time" gpurun_out/r03_c13/prof_mode$mode.txt
done
