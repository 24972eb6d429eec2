#!/bin/bash
# deep alignments with the LDS-tiled distance phase of the wide fits
out=gpurun_out/r04_c31; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ddeep.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
MPRG_DEEP_OUT=$out/deep_2000x4000.json timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 2 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-200 | tee $out/deep_2000x4000.txt
MPRG_DEEP_OUT=$out/deep_5000x10000.json timeout 900 python tools/deep_profile.py 5000 10000 7 --passes 1 --check 60 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-200 | tee $out/deep_5000x10000.txt
