#!/bin/bash
set -x
mkdir -p gpurun_out/r03_c03
python tools/forest_profile.py 3000 3 > gpurun_out/r03_c03/prof_3000.txt 2>&1
python tools/forest_profile.py 30000 3 > gpurun_out/r03_c03/prof_30000.txt 2>&1
cat gpurun_out/r03_c03/prof_3000.txt gpurun_out/r03_c03/prof_30000.txt
