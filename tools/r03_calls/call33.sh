#!/bin/bash
# centres of a fit feature-major across its restarts (km_ws): parity + per-entry-point times
mkdir -p gpurun_out/r03_c33
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddeep.py -m gpu -x -q 2>&1 | tail -4
python tools/forest_profile.py 30000 2 > gpurun_out/r03_c33/prof.txt 2>&1
grep "step 1\|kmeans\|device time" gpurun_out/r03_c33/prof.txt
