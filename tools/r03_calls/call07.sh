#!/bin/bash
mkdir -p gpurun_out/r03_c07
python tools/forest_profile.py 30000 2 > gpurun_out/r03_c07/prof_30000.txt 2>&1
grep -A12 pipelined gpurun_out/r03_c07/prof_30000.txt
