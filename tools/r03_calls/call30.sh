#!/bin/bash
mkdir -p gpurun_out/r03_c30
python tools/phase_timing.py 8192 > gpurun_out/r03_c30/phase_default.txt 2>&1
MPRG_KM_MODE=0 python tools/phase_timing.py 8192 > gpurun_out/r03_c30/phase_general.txt 2>&1
cat gpurun_out/r03_c30/phase_default.txt | head -14; echo; head -14 gpurun_out/r03_c30/phase_general.txt
